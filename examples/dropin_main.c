/*
 * dropin_main.c -- the reference's main() loop (qpsk.c:289-359) written against include/qpsk_dropin.h: the calls a
 * maintainer of MonsieurETM/QPSK keeps (create_control_loop, rrc_make, rx_frame on int16 PCM blocks), served by
 * libqpsk_hip.  The stimulus is the reference's own transmitter at CENTER + 50 Hz (qpsk.c:320), run on the GPU
 * through the batched API (qpsk_tx_symbols) since the drop-in surface is the receive path.
 *
 *   gcc -std=c11 -O2 -Iinclude examples/dropin_main.c -Lqpsk_amd -lqpsk_hip -Wl,-rpath,$PWD/qpsk_amd -o dropin_main
 *   ./dropin_main [blocks]
 *
 * Prints the loop's frequency estimate (fbb_offset_freq, qpsk.c:217) as the reference does; with the shipped
 * parameters it settles near the 50 Hz the transmitter is off by.  Exit code 0 iff it does.
 */
#define _DEFAULT_SOURCE /* M_PI, clock_gettime under -std=c11 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "qpsk_dropin.h"

#define FS 9600.0           /* qpsk.h:16-23, the shipped values */
#define RS 2400.0
#define CENTER 1500.0
#define FRAME_SIZE 512
#define CYCLES ((int)(FS / RS))
#define TAU (2.0 * M_PI)

#define CHECK(call)                                                                        \
    do {                                                                                   \
        if ((call) != QPSK_OK) {                                                           \
            fprintf(stderr, "%s:%d: %s\n  -> %s\n", __FILE__, __LINE__, #call, qpsk_last_error()); \
            exit(2);                                                                       \
        }                                                                                  \
    } while (0)

int main(int argc, char **argv)
{
    const int blocks = argc > 1 ? atoi(argv[1]) : 400;
    const int nsym = FRAME_SIZE / CYCLES;
    if (qpsk_device_count() <= 0) {
        fprintf(stderr, "no HIP device: libqpsk_hip has no CPU path\n");
        return 2;
    }

    /* ---- the receive side, as in the reference's main() */
    qpsk_params p;
    qpsk_params_default(&p);                               /* FS, RS, FRAME_SIZE of qpsk.h */
    CHECK(qpsk_dropin_configure(&p, CENTER));              /* replaces the compile-time #defines */
    create_control_loop((float)(TAU / 100.0), -1.0f, 1.0f); /* qpsk.c:302 */
    rrc_make((float)FS, (float)RS, .35f);                  /* qpsk.c:308 */

    /* ---- the transmitter (qpsk.c:316-333) through the batched API: one transmitter, carried state */
    qpsk_ctx *tx;
    CHECK(qpsk_ctx_create(&tx, -1, &p, NULL));
    CHECK(qpsk_tx_reset(tx, 1, CENTER + 50.0));
    void *d_sym, *d_pcm;
    CHECK(qpsk_dev_alloc(tx, &d_sym, nsym));
    CHECK(qpsk_dev_alloc(tx, &d_pcm, FRAME_SIZE * sizeof(int16_t)));
    uint8_t dibits[FRAME_SIZE];
    int16_t frame[FRAME_SIZE];

    srand(1);
    float hz = 0.0f;
    double rx_seconds = 0.0;            /* time spent inside rx_frame() alone: what replaces the reference's CPU work */
    for (int k = 0; k < blocks; k++) {
        for (int i = 0; i < nsym; i++) dibits[i] = (uint8_t)(rand() & 3);   /* two random bits per symbol, qpsk.c:325-327 */
        CHECK(qpsk_dev_upload(tx, d_sym, dibits, nsym));
        CHECK(qpsk_tx_symbols(tx, d_sym, nsym, d_pcm, NULL));
        CHECK(qpsk_dev_download(tx, frame, d_pcm, FRAME_SIZE * sizeof(int16_t)));

        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        rx_frame(frame);                                   /* qpsk.c:351 */
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if (k >= 10) rx_seconds += (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
        hz = qpsk_dropin_offset_freq();
        if (k % 50 == 49) printf("block %4d: timing index %d, offset %.2f Hz\n", k + 1, qpsk_dropin_timing_index(), hz);
    }
    qpsk_dev_free(tx, d_sym);
    qpsk_dev_free(tx, d_pcm);
    qpsk_ctx_destroy(tx);
    qpsk_dropin_shutdown();
    printf("final offset estimate %.2f Hz (transmitter is 50 Hz above the receiver's centre)\n", hz);
    if (blocks > 10)
        printf("rx_frame(): %.1f us per %d-sample block = %.3f Msamples/s through the drop-in (one block per call, host buffers)\n",
               1e6 * rx_seconds / (blocks - 10), FRAME_SIZE, (double)FRAME_SIZE * (blocks - 10) / rx_seconds / 1e6);
    return fabsf(hz - 50.0f) < 5.0f ? 0 : 1;
}
