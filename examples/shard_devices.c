/*
 * shard_devices.c -- a C host for N devices (include/qpsk_hip.h, MULTI): one batch of independent frames, shard r of N
 * on device r (contiguous frame ranges, no collective), one host thread and two streams per device inside the library, the symbols /
 * freq / phase of every step gathered over PCIe into ONE host array per output while the next step's kernel runs.
 *
 * It stands where the reference has `while (fread(frame, ...)) rx_frame(frame);` (qpsk.c:344-354) over its process-global modem state
 * (qpsk.c:36-53, costas_loop.c:13-23).
 *
 *   gcc -std=c11 -O2 -Iinclude examples/shard_devices.c -Lqpsk_amd -lqpsk_hip -Wl,-rpath,$PWD/qpsk_amd -o shard_devices
 *   ./shard_devices [total_frames] [symbols_per_frame] [steps] [shards]
 * `shards` (default: the number of GPUs) may exceed the GPU count: shards then share devices round robin (a rehearsal on a one-GPU box).
 *
 * The frames: random dibits through the library's transmit chain on each shard's device (qpsk_tx_symbols, the reference's
 * qpsk_packet_mod -> tx_frame, qpsk.c:225-285), straight into the shard's input buffer.  Checks: every shard's frames lock (offset
 * estimate 0 Hz: the baseband carries none), the gathered arrays are identical step after step and identical between the overlapped and the serial schedule.
 * Prints the time per step of both schedules and the PCIe bound of the gather.
 */
#define _POSIX_C_SOURCE 199309L
#include <math.h>
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

static double now(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static uint32_t lcg(uint32_t *s) { return *s = *s * 1664525u + 1013904223u; }

int main(int argc, char **argv)
{
    const long long total = argc > 1 ? atoll(argv[1]) : 2048;
    const int nsym = argc > 2 ? atoi(argv[2]) : 2048;
    const int steps = argc > 3 ? atoi(argv[3]) : 20;
    const int ngpu = qpsk_device_count();
    if (ngpu <= 0) {
        fprintf(stderr, "no HIP device: libqpsk_hip has no CPU path\n");
        return 2;
    }
    int nshard = argc > 4 ? atoi(argv[4]) : ngpu;
    if (nshard < 1 || nshard > 64) nshard = ngpu;
    int devices[64];
    for (int r = 0; r < nshard; r++) devices[r] = r % ngpu;

    qpsk_params p;
    qpsk_params_default(&p);
    p.fs = 19200.0;
    p.rs = 2400.0;                       /* CYCLES = 8 */
    p.frame_size = nsym * 8;
    p.timing_mode = QPSK_TIMING_FIXED;
    p.fixed_index = 6;                   /* the two 63-sample filter delays: 126 mod 8 */

    qpsk_multi *mj;
    CHECK(qpsk_multi_create(&mj, devices, nshard, &p));
    CHECK(qpsk_multi_load(mj, total, NULL));

    /* every shard makes its own frames on its own device: the transmit chain's shaped complex baseband (in front of the up-mix,
     * qpsk.c:243): no carrier offset, the loops lock at 0 Hz */
    for (int r = 0; r < nshard; r++) {
        long long first, count;
        qpsk_ctx *ctx;
        float *d_in;
        CHECK(qpsk_multi_shard(mj, r, NULL, &first, &count, &ctx, &d_in));
        if (count == 0) continue;
        const size_t n = (size_t)count * (size_t)nsym;
        uint8_t *h_tx = malloc(n);
        uint32_t seed = 4242u + 7919u * (uint32_t)first;
        for (size_t i = 0; i < n; i++) h_tx[i] = (uint8_t)(lcg(&seed) >> 30);
        void *d_tx;
        CHECK(qpsk_dev_alloc(ctx, &d_tx, n));
        CHECK(qpsk_dev_upload(ctx, d_tx, h_tx, n));
        CHECK(qpsk_tx_reset(ctx, (int)count, 1500.0));
        CHECK(qpsk_tx_symbols(ctx, d_tx, nsym, NULL, d_in));
        CHECK(qpsk_ctx_sync(ctx));
        CHECK(qpsk_dev_free(ctx, d_tx));
        free(h_tx);
    }

    const size_t nall = (size_t)total * (size_t)nsym;
    uint8_t *sym = malloc(nall), *sym_ref = malloc(nall);
    float *freq = malloc(sizeof(float) * (size_t)total), *phase = malloc(sizeof(float) * (size_t)total);

    /* serial schedule: a step's copy-back is waited for before the next step starts */
    CHECK(qpsk_multi_rx_begin(mj, 0));
    CHECK(qpsk_multi_rx_end(mj, 0, sym_ref, freq, phase));      /* first call: allocation inside the library */
    double t0 = now();
    for (int k = 0; k < steps; k++) {
        CHECK(qpsk_multi_rx_begin(mj, 0));
        CHECK(qpsk_multi_rx_end(mj, 0, sym, freq, phase));
    }
    const double serial = (now() - t0) / steps;
    int bad = memcmp(sym, sym_ref, nall) != 0;

    /* overlapped schedule: step k + 1 is enqueued before step k's results are waited for (two result slots) */
    t0 = now();
    CHECK(qpsk_multi_rx_begin(mj, 0));
    for (int k = 1; k < steps; k++) {
        CHECK(qpsk_multi_rx_begin(mj, k & 1));
        CHECK(qpsk_multi_rx_end(mj, (k - 1) & 1, sym, freq, phase));
        bad |= memcmp(sym, sym_ref, nall) != 0;
    }
    CHECK(qpsk_multi_rx_end(mj, (steps - 1) & 1, sym, freq, phase));
    const double overlapped = (now() - t0) / steps;
    bad |= memcmp(sym, sym_ref, nall) != 0;

    /* the same overlapped schedule with the copy-back by DMA straight into page-locked arrays and the symbols four to a byte: what a host
     * that gathers every step should run (the copy-back, not the kernel, is the bound otherwise) */
    const size_t prow = ((size_t)nsym + 3) / 4;
    uint8_t *psym[2];
    float *pfreq[2], *pphase[2];
    CHECK(qpsk_multi_set_packed(mj, 1));
    for (int k = 0; k < 2; k++) {
        CHECK(qpsk_host_alloc((void **)&psym[k], (size_t)total * prow));
        CHECK(qpsk_host_alloc((void **)&pfreq[k], sizeof(float) * (size_t)total));
        CHECK(qpsk_host_alloc((void **)&pphase[k], sizeof(float) * (size_t)total));
        CHECK(qpsk_multi_set_direct_output(mj, k, psym[k], pfreq[k], pphase[k]));
    }
    CHECK(qpsk_multi_rx_begin(mj, 0));
    CHECK(qpsk_multi_rx_end(mj, 0, NULL, NULL, NULL));          /* first touch of the page-locked arrays */
    t0 = now();
    CHECK(qpsk_multi_rx_begin(mj, 0));
    for (int k = 1; k < steps; k++) {
        CHECK(qpsk_multi_rx_begin(mj, k & 1));
        CHECK(qpsk_multi_rx_end(mj, (k - 1) & 1, NULL, NULL, NULL));
    }
    CHECK(qpsk_multi_rx_end(mj, (steps - 1) & 1, NULL, NULL, NULL));
    const double packed = (now() - t0) / steps;
    CHECK(qpsk_unpack_symbols_host(psym[(steps - 1) & 1], total, nsym, sym));
    bad |= memcmp(sym, sym_ref, nall) != 0;
    bad |= memcmp(pfreq[(steps - 1) & 1], freq, sizeof(float) * (size_t)total) != 0;
    printf("overlapped, direct DMA, symbols four to a byte: %.3f ms per step (%.1f MiB gathered per step)\n", packed * 1e3,
           ((double)total * (double)prow + 8.0 * (double)total) / 1048576.0);
    for (int k = 0; k < 2; k++) {
        CHECK(qpsk_multi_set_direct_output(mj, k, NULL, NULL, NULL));
        qpsk_host_free(psym[k]); qpsk_host_free(pfreq[k]); qpsk_host_free(pphase[k]);
    }

    long unlocked = 0;
    for (long long f = 0; f < total; f++) {
        const double hz = (double)freq[f] * p.rs / 6.283185307179586;
        unlocked += !(fabs(hz) < 2.0);
    }
    const double bytes = (double)nall + 8.0 * (double)total;
    printf("%lld frames x %d symbols on %d shard(s) of %d GPU(s): %.3f ms per step serial, %.3f ms overlapped "
           "(%.1f MiB gathered per step: %.3f ms at 63 GB/s per device in parallel)\n",
           total, nsym, nshard, ngpu, serial * 1e3, overlapped * 1e3, bytes / 1048576.0, bytes / nshard / 63e9 * 1e3);
    printf("frames not locked (|offset estimate| >= 2 Hz): %ld; gathered symbols %s across steps and schedules\n", unlocked, bad ? "DIFFER" : "identical");
    qpsk_multi_destroy(mj);
    free(sym); free(sym_ref); free(freq); free(phase);
    return (bad || unlocked) ? 1 : 0;
}
