/*
 * loopback.c -- a C host using nothing but libqpsk_hip: the reference's transmitter (qpsk_packet_mod ->
 * tx_frame, qpsk.c:225-285) and its receive path (rrc_fir -> decimate -> Costas -> slicer, qpsk.c:125-212)
 * for a batch of independent frames on EVERY GPU of the node, one context per device, no host thread per
 * device needed (all calls are asynchronous on the device's stream), no collective.
 *
 *   gcc -std=c11 -O2 -Iinclude examples/loopback.c -Lqpsk_amd -lqpsk_hip -Wl,-rpath,$PWD/qpsk_amd -o loopback
 *   ./loopback [frames_per_gpu] [symbols_per_frame]
 *
 * Each device: random dibits -> qpsk_tx_symbols (shaped complex baseband) -> qpsk_rx_batch with the timing
 * offset of the two 63-sample filter delays (126 mod 8 = 6).  Received symbol i is transmitted symbol i - 15 up
 * to the loop's quarter-turn ambiguity; the program resolves the rotation per frame and counts decision errors
 * after the loop has settled.  Exit code 0 iff there are none.
 *
 * Decisions are taken from costas_frame[] (qpsk.c:197), by quadrant: the reference's phase detector
 * (costas_loop.c:44-47) is at rest when |I| = |Q|, so the loop parks the constellation on the diagonals, and
 * qpsk_demod() (qpsk.c:74-79) then turns it by another 45 degrees, i.e. onto its own decision boundaries
 * ("hit and miss", README.md:10).  The library reproduces those slicer bytes bit for bit (d_sym); a receiver
 * that wants the data reads the quadrant of costas_frame[] instead, as below.
 */
#define _POSIX_C_SOURCE 199309L   /* clock_gettime under -std=c11 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "qpsk_hip.h"

#define CHECK(call)                                                                        \
    do {                                                                                   \
        if ((call) != QPSK_OK) {                                                           \
            fprintf(stderr, "%s:%d: %s\n  -> %s\n", __FILE__, __LINE__, #call, qpsk_last_error()); \
            exit(2);                                                                       \
        }                                                                                  \
    } while (0)

enum { DELAY = 15, SETTLE = 256, MAX_DEV = 16 };

typedef struct {
    qpsk_ctx *ctx;
    uint8_t *h_tx;
    float *h_z;
    void *d_tx, *d_bb, *d_rx, *d_z, *d_freq, *d_phase;
} device_job;

/* quadrant of a de-rotated symbol as the dibit value whose constellation point (qpsk.c:58-63), turned by 45
 * degrees, lies there: 1 -> I, j -> II, -1 -> III, -j -> IV */
static uint8_t quadrant_dibit(float re, float im)
{
    return re > 0.0f ? (im > 0.0f ? 0 : 2) : (im > 0.0f ? 1 : 3);
}

static uint32_t lcg(uint32_t *s) { return *s = *s * 1664525u + 1013904223u; }

int main(int argc, char **argv)
{
    const int frames = argc > 1 ? atoi(argv[1]) : 256;
    const int nsym = argc > 2 ? atoi(argv[2]) : 2048;
    int ndev = qpsk_device_count();
    if (ndev <= 0) {
        fprintf(stderr, "no HIP device: libqpsk_hip has no CPU path\n");
        return 2;
    }
    if (ndev > MAX_DEV) ndev = MAX_DEV;

    qpsk_params p;
    qpsk_params_default(&p);
    p.fs = 19200.0;
    p.rs = 2400.0;                       /* CYCLES = 8 */
    p.frame_size = nsym * 8;
    p.timing_mode = QPSK_TIMING_FIXED;
    p.fixed_index = 6;                   /* (63 + 63) mod 8 */

    device_job job[MAX_DEV];
    const size_t nsyms = (size_t)frames * nsym, nsamp = nsyms * 8;
    for (int d = 0; d < ndev; d++) {
        device_job *j = &job[d];
        CHECK(qpsk_ctx_create(&j->ctx, d, &p, NULL));
        j->h_tx = malloc(nsyms);
        j->h_z = malloc(nsyms * 2 * sizeof(float));
        uint32_t seed = 12345u + 977u * (uint32_t)d;
        for (size_t i = 0; i < nsyms; i++) j->h_tx[i] = (uint8_t)(lcg(&seed) >> 30);
        CHECK(qpsk_dev_alloc(j->ctx, &j->d_tx, nsyms));
        CHECK(qpsk_dev_alloc(j->ctx, &j->d_rx, nsyms));
        CHECK(qpsk_dev_alloc(j->ctx, &j->d_z, nsyms * 2 * sizeof(float)));
        CHECK(qpsk_dev_alloc(j->ctx, &j->d_bb, nsamp * 2 * sizeof(float)));
        CHECK(qpsk_dev_alloc(j->ctx, &j->d_freq, frames * sizeof(float)));
        CHECK(qpsk_dev_alloc(j->ctx, &j->d_phase, frames * sizeof(float)));
        CHECK(qpsk_dev_upload(j->ctx, j->d_tx, j->h_tx, nsyms));
        CHECK(qpsk_tx_reset(j->ctx, frames, 1550.0));
    }

    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int d = 0; d < ndev; d++) {     /* every device gets its work before anyone is waited for */
        device_job *j = &job[d];
        CHECK(qpsk_tx_symbols(j->ctx, j->d_tx, nsym, NULL, j->d_bb));
        CHECK(qpsk_rx_batch(j->ctx, j->d_bb, frames, j->d_rx, j->d_freq, j->d_phase, j->d_z, NULL, NULL));
    }
    for (int d = 0; d < ndev; d++) CHECK(qpsk_ctx_sync(job[d].ctx));
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const double sec = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);

    /* quarter turn of the constellation on the dibit value: 1 -> j -> -1 -> -j is 0 -> 1 -> 3 -> 2 (qpsk.c:58-63) */
    static const uint8_t turn[4] = {1, 3, 0, 2};
    long errors = 0, checked = 0;
    for (int d = 0; d < ndev; d++) {
        device_job *j = &job[d];
        CHECK(qpsk_dev_download(j->ctx, j->h_z, j->d_z, nsyms * 2 * sizeof(float)));
        for (int f = 0; f < frames; f++) {
            const uint8_t *tx = j->h_tx + (size_t)f * nsym;
            const float *z = j->h_z + (size_t)f * nsym * 2;
            long best = -1;
            for (int r = 0; r < 4; r++) {
                long miss = 0;
                for (int i = SETTLE; i < nsym; i++) {
                    uint8_t v = tx[i - DELAY];
                    for (int k = 0; k < r; k++) v = turn[v];
                    miss += v != quadrant_dibit(z[2 * i], z[2 * i + 1]);
                }
                if (best < 0 || miss < best) best = miss;
            }
            if (nsym > SETTLE) {
                errors += best;
                checked += nsym - SETTLE;
            }
        }
    }
    printf("%d device(s) x %d frames x %d symbols: tx + rx in %.3f ms (first call, includes allocation inside the library)\n",
           ndev, frames, nsym, sec * 1e3);
    printf("decisions checked %ld, errors %ld\n", checked, errors);

    for (int d = 0; d < ndev; d++) {
        device_job *j = &job[d];
        qpsk_dev_free(j->ctx, j->d_tx); qpsk_dev_free(j->ctx, j->d_rx); qpsk_dev_free(j->ctx, j->d_bb);
        qpsk_dev_free(j->ctx, j->d_freq); qpsk_dev_free(j->ctx, j->d_phase); qpsk_dev_free(j->ctx, j->d_z);
        qpsk_ctx_destroy(j->ctx);
        free(j->h_tx); free(j->h_z);
    }
    return errors == 0 && checked > 0 ? 0 : 1;
}
