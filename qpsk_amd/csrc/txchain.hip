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
constexpr int TXM_TILE = 64;

__global__ void __launch_bounds__(64)
tx_upmix_kernel(const float2 *__restrict__ sig, int16_t *__restrict__ pcm, float *state, int nstreams, int length)
{
    __shared__ float2 tin[TXM_STREAMS][TXM_TILE + 1];
    __shared__ int16_t tout[TXM_STREAMS][TXM_TILE + 2];
    const int lane = threadIdx.x, f0 = blockIdx.x * TXM_STREAMS;
    const bool scanning = lane < TXM_STREAMS && f0 + lane < nstreams;
    const int f = min(f0 + (lane & (TXM_STREAMS - 1)), nstreams - 1);
    float pr = state[4 * f], pi = state[4 * f + 1];
    const float rr = state[4 * f + 2], ri = state[4 * f + 3];
    const int ntiles = (length + TXM_TILE - 1) / TXM_TILE;
    float2 pre[TXM_STREAMS];
    auto fetch = [&](int t) {
        const int s_ = min(t * TXM_TILE + lane, length - 1);
#pragma unroll
        for (int r = 0; r < TXM_STREAMS; r++)
            pre[r] = sig[(size_t)min(f0 + r, nstreams - 1) * length + s_];
    };
    fetch(0);
    for (int t = 0; t < ntiles; t++) {
#pragma unroll
        for (int r = 0; r < TXM_STREAMS; r++)
            tin[r][lane] = pre[r];
        if (t + 1 < ntiles) fetch(t + 1);
        __syncthreads();
        const int cnt = min(TXM_TILE, length - t * TXM_TILE);
        if (scanning) {
#pragma unroll 8
            for (int i = 0; i < cnt; i++) {
                const float nr = pr * rr - pi * ri;
                const float ni = pr * ri + pi * rr;
                pr = nr;
                pi = ni;
                const float2 s = tin[lane][i];
                const float re = s.x * pr - s.y * pi;       /* real part of signal[i] * phase; the imaginary part is discarded */
                tout[lane][i] = (int16_t)(re * 16384.0f);   /* C conversion: truncation toward zero (in range for this modem) */
            }
        }
        __syncthreads();
        if (lane < cnt) {
#pragma unroll
            for (int r = 0; r < TXM_STREAMS; r++)
                if (f0 + r < nstreams)
                    pcm[(size_t)(f0 + r) * length + t * TXM_TILE + lane] = tout[r][lane];
        }
    }
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
