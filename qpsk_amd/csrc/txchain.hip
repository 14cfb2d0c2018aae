/*
 * txchain.hip -- the transmit side of the reference (qpsk.c:225-285) for a batch of transmitters with carried
 * state: Gray map, zero-stuffing and RRC shaping in one kernel, then the carrier up-mix and
 * the int16 conversion.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"
#include "qpsk_device.h"

namespace qpsk {

/*
 * tx_shape_kernel: qpsk_packet_mod() + the zero-stuffing loop + rrc_fir(tx_filter, ...) (qpsk.c:273-282,
 * 232-238, 243).  The reference pushes CYCLES-1 zeros per symbol through the 127-tap filter, so 7 of 8 terms of
 * every output sample are +-0.  Leaving them out gives the same float: the accumulator starts at +0, adding +-0
 * to +0 gives +0 and adding +-0 to anything else changes nothing (rrc_fir.c:22-26 cannot produce -0).  What is
 * left is the reference's own term, in its own order, for the ~127/CYCLES taps that meet a symbol: an eighth of
 * the work of the full filter.  Checked bit for bit against the reference's transmitter (tests/golden/tx_*.npz).
 *
 * State per transmitter: its last TXS_HIST symbols (code 4 = none yet), which is what the reference's tx_filter
 * delay line holds.  grid = (tiles of TXS_TILE symbols, transmitters).
 */
constexpr int TXS_THREADS = 256;
constexpr int TXS_TILE = 256;        /* symbols per workgroup */
constexpr int TXS_HIST = 128;        /* >= 126 / CYCLES + 1 for every CYCLES >= 1 */

__global__ void __launch_bounds__(TXS_THREADS)
tx_shape_kernel(const uint8_t *__restrict__ sym, const uint8_t *__restrict__ hist, const float *__restrict__ taps_g,
                float2 *__restrict__ sig, int nsym, int cycles)
{
    __shared__ float taps[128];
    __shared__ float2 ss[TXS_HIST + TXS_TILE];            /* the symbols as constellation points; (0, 0) = none */
    const int tid = threadIdx.x, f = blockIdx.y;
    const int s_first = blockIdx.x * TXS_TILE;            /* first symbol of the tile, call-relative */
    if (tid < 128) taps[tid] = tid < NTAPS ? taps_g[tid] : 0.0f;
    for (int u = tid; u < TXS_HIST + TXS_TILE; u += TXS_THREADS) {
        const int s = s_first - TXS_HIST + u;
        int k = 4;
        if (s < 0) k = hist[(size_t)f * TXS_HIST + (TXS_HIST + s)];
        else if (s < nsym) k = sym[(size_t)f * nsym + s] & 3;
        /* constellation[] = 1, j, -j, -1 (qpsk.c:58-63) */
        ss[u] = make_float2(k == 0 ? 1.0f : (k == 3 ? -1.0f : 0.0f), k == 1 ? 1.0f : (k == 2 ? -1.0f : 0.0f));
    }
    __syncthreads();
    const int n_first = s_first * cycles;
    const int n_end = min(s_first + TXS_TILE, nsym) * cycles;
    for (int n = n_first + tid; n < n_end; n += TXS_THREADS) {
        /* tap i of rrc_fir.c:24-25 meets the input sample n - 126 + i; symbols sit at multiples of CYCLES */
        const int a = n - (NTAPS - 1);
        int s = a >= 0 ? (a + cycles - 1) / cycles : -((-a) / cycles);      /* ceil(a / CYCLES) */
        int i = s * cycles - a;
        float2 acc = make_float2(0.0f, 0.0f);
        for (; i < NTAPS; i += cycles, s++)
            fir_mac(acc, ss[s - s_first + TXS_HIST], taps[i]);   /* the reference's own term, rrc_fir.c:24-25 */
        sig[(size_t)f * nsym * cycles + n] = fir_gain(acc);
    }
}

/* the last TXS_HIST symbols after this call: older ones slide down, the call's symbols follow */
__global__ void __launch_bounds__(TXS_HIST)
tx_history_kernel(const uint8_t *__restrict__ sym, uint8_t *hist, int nsym)
{
    const int h = threadIdx.x, f = blockIdx.x;
    const int s = nsym - TXS_HIST + h;                     /* call-relative index of the symbol that lands at h */
    const uint8_t k = s < 0 ? hist[(size_t)f * TXS_HIST + (TXS_HIST + s)] : (uint8_t)(sym[(size_t)f * nsym + s] & 3);
    __syncthreads();
    hist[(size_t)f * TXS_HIST + h] = k;
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

int tx_history_symbols(void) { return TXS_HIST; }

int launch_tx_shape(const uint8_t *sym, uint8_t *hist, const float *taps, float *sig, int nstreams, int nsym,
                    int cycles, hipStream_t s)
{
    hipLaunchKernelGGL(tx_shape_kernel, dim3((nsym + TXS_TILE - 1) / TXS_TILE, nstreams), dim3(TXS_THREADS), 0, s, sym,
                       hist, taps, reinterpret_cast<float2 *>(sig), nsym, cycles);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(tx_history_kernel, dim3(nstreams), dim3(TXS_HIST), 0, s, sym, hist, nsym);
    return (int)hipGetLastError();
}

int launch_tx_upmix(const float *sig, int16_t *pcm, float *state, int nstreams, int length, hipStream_t s)
{
    hipLaunchKernelGGL(tx_upmix_kernel, dim3((nstreams + TXM_STREAMS - 1) / TXM_STREAMS), dim3(64), 0, s,
                       reinterpret_cast<const float2 *>(sig), pcm, state, nstreams, length);
    return (int)hipGetLastError();
}

} // namespace qpsk
