/*
 * txchain.hip -- the transmit side of the reference (qpsk.c:225-285) for a batch of transmitters with carried
 * state: Gray map + zero-stuffing here, the RRC shaping is rrc_fir_kernel (the same
 * function the reference calls, qpsk.c:243), then the carrier up-mix and the int16 conversion.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"

namespace qpsk {

/* qpsk_packet_mod() + the zero-stuffing loop of tx_frame() (qpsk.c:232-238, 273-282): symbol index
 * k = (bits[s] << 1) | bits[s+1] -> constellation[k] (qpsk.c:58-63, 270) at sample i*CYCLES, zeros between */
__global__ void __launch_bounds__(256)
tx_map_kernel(const uint8_t *__restrict__ sym, float2 *__restrict__ sig, size_t total_samples, int cycles)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total_samples) return;
    float2 v = make_float2(0.0f, 0.0f);
    if (i % cycles == 0) {
        const int k = sym[i / cycles] & 3;
        v.x = k == 0 ? 1.0f : (k == 3 ? -1.0f : 0.0f);
        v.y = k == 1 ? 1.0f : (k == 2 ? -1.0f : 0.0f);
    }
    sig[i] = v;
}

/* qpsk.c:248-261: phase *= rect; signal *= phase; sample = (int16_t)(crealf(signal) * 16384.0f); then the
 * carrier phase is renormalised.  Serial per transmitter like mixer_kernel: 16 transmitters per single-wave
 * workgroup, LDS tiles for coalesced rows.  state: [n][4] = phase.re, phase.im, rect.re, rect.im */
constexpr int TXM_STREAMS = 16;
constexpr int TXM_TILE = 128;

__global__ void __launch_bounds__(64)
tx_upmix_kernel(const float2 *__restrict__ sig, int16_t *__restrict__ pcm, float *state, int nstreams, int length)
{
    /* two passes per tile, as in mixer_kernel: lanes 0..15 run only the carrier recurrence and leave the tile's 64
     * phases in LDS; then all 64 lanes take one transmitter's row at a time (lane = sample) */
    __shared__ __attribute__((aligned(16))) float2 ph[TXM_STREAMS][TXM_TILE + 2];
    const int lane = threadIdx.x, f0 = blockIdx.x * TXM_STREAMS;
    const bool scanning = lane < TXM_STREAMS && f0 + lane < nstreams;
    const int f = min(f0 + (lane & (TXM_STREAMS - 1)), nstreams - 1);
    float2 p = make_float2(state[4 * f], state[4 * f + 1]);
    const float rr = state[4 * f + 2], ri = state[4 * f + 3];
    float nri = -ri;
    asm volatile("" : "+v"(nri));   /* opaque, see mixer_kernel */
    auto step = [&]() {              /* fbb_tx_phase *= fbb_tx_rect, qpsk.c:249 */
        const float2 a = make_float2(p.x * rr, p.y * rr);
        const float2 b = make_float2(p.y * nri, p.x * ri);
        p = make_float2(a.x + b.x, a.y + b.y);
    };
    const int ntiles = (length + TXM_TILE - 1) / TXM_TILE;
    /* the paired accesses need whole rows of transmitters, an even length and aligned bases (wave-uniform) */
    const bool fast = f0 + TXM_STREAMS <= nstreams && (length & 1) == 0 &&
                      (reinterpret_cast<uintptr_t>(sig) & 15) == 0 && (reinterpret_cast<uintptr_t>(pcm) & 3) == 0;
    float4 pre[TXM_STREAMS];
    auto fetch = [&](int t) {                               /* full tiles on the fast path only */
#pragma unroll
        for (int r = 0; r < TXM_STREAMS; r++) {
            const float2 *rowp = sig + (size_t)(f0 + r) * length + (size_t)t * TXM_TILE;   /* uniform base */
            pre[r] = reinterpret_cast<const float4 *>(rowp)[lane];
        }
    };
    if (fast && length >= TXM_TILE) fetch(0);
    for (int t = 0; t < ntiles; t++) {
        const int cnt = min(TXM_TILE, length - t * TXM_TILE);
        __syncthreads();
        if (scanning) {
            float4 *prow = reinterpret_cast<float4 *>(&ph[lane][0]);
            if (cnt == TXM_TILE) {
#pragma unroll 8
                for (int i = 0; i < TXM_TILE / 2; i++) {
                    step();
                    const float2 p0 = p;
                    step();
                    prow[i] = make_float4(p0.x, p0.y, p.x, p.y);
                }
            } else {
                for (int i = 0; i < cnt; i++) {
                    step();
                    ph[lane][i] = p;
                }
            }
        }
        __syncthreads();
        if (fast && cnt == TXM_TILE) {                      /* wave-uniform: no per-row branches, uniform row bases */
            float4 c[TXM_STREAMS];
#pragma unroll
            for (int r = 0; r < TXM_STREAMS; r++)
                c[r] = *reinterpret_cast<const float4 *>(&ph[r][2 * lane]);
#pragma unroll
            for (int r = 0; r < TXM_STREAMS; r++) {
                int16_t *orow = pcm + (size_t)(f0 + r) * length + (size_t)t * TXM_TILE;
                const float4 s = pre[r];
                /* real part of signal[i] * phase; the imaginary part is discarded; C conversion: toward zero */
                const int16_t lo = (int16_t)((s.x * c[r].x - s.y * c[r].y) * 16384.0f);
                const int16_t hi = (int16_t)((s.z * c[r].z - s.w * c[r].w) * 16384.0f);
                reinterpret_cast<uint32_t *>(orow)[lane] = (uint32_t)(uint16_t)lo | ((uint32_t)(uint16_t)hi << 16);
            }
            if ((t + 2) * TXM_TILE <= length) fetch(t + 1);
        } else {
            for (int i = lane; i < cnt; i += 64) {
#pragma unroll 1
                for (int r = 0; r < TXM_STREAMS && f0 + r < nstreams; r++) {
                    const size_t at = (size_t)(f0 + r) * length + (size_t)t * TXM_TILE + i;
                    const float2 c = ph[r][i], s = sig[at];
                    pcm[at] = (int16_t)((s.x * c.x - s.y * c.y) * 16384.0f);
                }
            }
        }
    }
    const float pr = p.x, pi = p.y;
    if (scanning) {
        const float mag = (float)sqrt((double)pr * (double)pr + (double)pi * (double)pi);   /* qpsk.c:253 */
        state[4 * f] = pr / mag;
        state[4 * f + 1] = pi / mag;
    }
}

int launch_tx_map(const uint8_t *sym, float *sig, size_t total_samples, int cycles, hipStream_t s)
{
    hipLaunchKernelGGL(tx_map_kernel, dim3((unsigned)((total_samples + 255) / 256)), dim3(256), 0, s, sym,
                       reinterpret_cast<float2 *>(sig), total_samples, cycles);
    return (int)hipGetLastError();
}

int launch_tx_upmix(const float *sig, int16_t *pcm, float *state, int nstreams, int length, hipStream_t s)
{
    hipLaunchKernelGGL(tx_upmix_kernel, dim3((nstreams + TXM_STREAMS - 1) / TXM_STREAMS), dim3(64), 0, s,
                       reinterpret_cast<const float2 *>(sig), pcm, state, nstreams, length);
    return (int)hipGetLastError();
}

} // namespace qpsk
