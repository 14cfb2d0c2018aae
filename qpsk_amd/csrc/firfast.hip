/*
 * firfast.hip -- qpsk_rrc_fir_batch_fast: the RRC FIR (reference rrc_fir.c:17-30) by overlap-save with 512-point complex
 * FFTs (the block length of algorithms/fft.h:44).  NOT a parity path: an FFT changes the summation
 * order of rrc_fir.c:22-26, so the output agrees with rrc_fir() to a few 1e-7 of the frame's peak and not bit for bit; the
 * library never routes it into qpsk_rx_batch, whose symbols must stay exact.  What it is for: callers that want the
 * full-rate filtered block (508 unfused flops per sample in rrc_fir_kernel, VALU-bound at ~0.65 ms per 4096 x 16384 block)
 * at ~130 fused flops per sample.
 *
 * One wave = one block of 512 points = 126 samples of overlap + 386 new ones:
 *     X = FFT512(segment);   Y = X * H;   y = IFFT512(Y) / 512;   out[386 b + i] = y[126 + i], i < 386
 * with H = GAIN * FFT512(h), h[k] = taps[126 - k] (rrc_fir.c: memory[126] is the newest sample), computed on the host in
 * double (host_math.c) -- GAIN once more on top of taps that already sum to GAIN, as the reference has it.
 *
 * The transform on 64 lanes x 8 points, 512 = 8 x 8 x 8, three radix-8 passes in registers with two exchanges through LDS
 * (4.6 KB per wave).  With n = t' + 8 r' + 64 r and k = k1 + 8 k1' + 64 k2':
 *     forward   lane t = t' + 8 r', registers r:   DFT8 over r   -> x W512^(t k1)    -> exchange -> lane (k1, t'), registers r'
 *               DFT8 over r'  -> x W64^(t' k1')  -> exchange -> lane (k1, k1'), registers t':  DFT8 over t' -> X[k], registers k2'
 *     inverse   the same passes backwards with conjugate twiddles; it ends on lane t, registers r: the layout the forward
 *               transform started from, so loads and stores are 512-byte rows, 64 lanes x 8 bytes, coalesced.
 * fp32, fused multiply-adds allowed (this file is compiled with the library's -ffp-contract=off like every other; the
 * FMAs here are written out with fmaf).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"

namespace qpsk {

namespace firfast {
constexpr int NTAPS = 127;                                          /* rrc_fir.h:13 */
constexpr int NFFT = 512, HIST = NTAPS - 1, HOP = NFFT - HIST;      /* 386 new samples per block */
constexpr int WAVES = 4;                                            /* blocks per workgroup */
constexpr int ROW = 72;                                             /* LDS row stride in float2.  Element (k1, t) of the first exchange at
                                                                       72 k1 + t, element (k1, k1', t') of the second at 72 k1 + 9 k1' + t':
                                                                       the 32 lanes of a half wave hit 32 different bank pairs on both
                                                                       sides of both exchanges ((8 k1 + t', 8 k1 + 9 k1') mod 32 are bijections) */
constexpr int LDS_PER_WAVE = 8 * ROW;                               /* float2 slots */

struct c32 { float x, y; };
__device__ __forceinline__ c32 cadd(c32 a, c32 b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ c32 csub(c32 a, c32 b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ c32 cmul(c32 a, c32 b) { return {fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x)}; }
/* multiply by -j (forward) or +j (inverse) */
template <bool INV> __device__ __forceinline__ c32 rot90(c32 a) { return INV ? c32{-a.y, a.x} : c32{a.y, -a.x}; }

/* 8-point DFT in registers: v[k] = sum_r v[r] W8^(+-r k), natural order in and out */
template <bool INV> __device__ __forceinline__ void dft8(c32 (&v)[8])
{
    constexpr float H = 0.70710678118654752440f;
    c32 a0 = cadd(v[0], v[4]), a1 = csub(v[0], v[4]), a2 = cadd(v[2], v[6]), a3 = rot90<INV>(csub(v[2], v[6]));
    c32 b0 = cadd(v[1], v[5]), b1 = csub(v[1], v[5]), b2 = cadd(v[3], v[7]), b3 = rot90<INV>(csub(v[3], v[7]));
    c32 e0 = cadd(a0, a2), e2 = csub(a0, a2), e1 = cadd(a1, a3), e3 = csub(a1, a3);       /* DFT4 of the even samples */
    c32 o0 = cadd(b0, b2), o2 = csub(b0, b2), o1 = cadd(b1, b3), o3 = csub(b1, b3);       /* DFT4 of the odd samples */
    /* odd part times W8^k: k = 1: (1 -+ j)/sqrt2, k = 2: -+j, k = 3: (-1 -+ j)/sqrt2 */
    c32 t1 = INV ? c32{(o1.x - o1.y) * H, (o1.x + o1.y) * H} : c32{(o1.x + o1.y) * H, (o1.y - o1.x) * H};
    c32 t2 = rot90<INV>(o2);
    c32 t3 = INV ? c32{(-o3.x - o3.y) * H, (o3.x - o3.y) * H} : c32{(o3.y - o3.x) * H, (-o3.x - o3.y) * H};
    v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
    v[1] = cadd(e1, t1); v[5] = csub(e1, t1);
    v[2] = cadd(e2, t2); v[6] = csub(e2, t2);
    v[3] = cadd(e3, t3); v[7] = csub(e3, t3);
}

/* between the two sides of an exchange: the lanes of a wave run in lock step and its LDS operations complete in order, so no
 * hardware barrier is needed -- only the compiler must not move a read above the writes of the other lanes */
__device__ __forceinline__ void exchange_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* tw[m] = exp(-2 pi j m / 512); the inverse transform conjugates */
template <bool INV> __device__ __forceinline__ c32 twiddle(const float2 *tw, int m)
{
    const float2 w = tw[m & (NFFT - 1)];
    return {w.x, INV ? -w.y : w.y};
}
} // namespace firfast

using namespace firfast;

__global__ void __launch_bounds__(64 * WAVES)
rrc_fir_fast_kernel(const float2 *__restrict__ x, const float2 *__restrict__ memory, float2 *__restrict__ y,
                    const float2 *__restrict__ Hf, const float2 *__restrict__ twg, int nframes, int length, int nblocks)
{
    __shared__ float2 tw[NFFT];
    __shared__ float2 ex[WAVES][LDS_PER_WAVE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < NFFT; i += blockDim.x) tw[i] = twg[i];
    __syncthreads();
    const long long job = (long long)blockIdx.x * WAVES + wave;          /* (frame, block) */
    if (job >= (long long)nframes * nblocks) return;
    const int frame = (int)(job / nblocks), b = (int)(job % nblocks);
    const float2 *xf = x + (size_t)frame * length;
    const float2 *mf = memory ? memory + (size_t)frame * NTAPS : nullptr;
    float2 *e = ex[wave];
    const int n0 = b * HOP - HIST;                                       /* frame index of the segment's first point */

    /* ---- load: lane t holds points t + 64 r.  Before the frame: the delay line (memory[1 + i] = x[i - 126]); past it: 0 */
    c32 v[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int n = n0 + lane + 64 * r;
        float2 s = make_float2(0.0f, 0.0f);
        if (n >= 0) { if (n < length) s = xf[n]; }
        else if (mf) s = mf[NTAPS + n];                                  /* n = -1 -> memory[126], the newest */
        v[r] = {s.x, s.y};
    }
    /* ---- forward pass A: DFT8 over r, twiddle W512^(t k1), exchange to lane (k1, t'), registers r' */
    dft8<false>(v);
#pragma unroll
    for (int k1 = 1; k1 < 8; k1++) v[k1] = cmul(v[k1], twiddle<false>(tw, lane * k1));
#pragma unroll
    for (int k1 = 0; k1 < 8; k1++) e[k1 * ROW + lane] = make_float2(v[k1].x, v[k1].y);
    exchange_fence();
    {
        const int k1 = lane >> 3, tp = lane & 7;
#pragma unroll
        for (int rp = 0; rp < 8; rp++) { const float2 s = e[k1 * ROW + tp + 8 * rp]; v[rp] = {s.x, s.y}; }
        /* pass B: DFT8 over r', twiddle W64^(t' k1'), exchange to lane (k1, k1'), registers t' */
        dft8<false>(v);
#pragma unroll
        for (int kp = 1; kp < 8; kp++) v[kp] = cmul(v[kp], twiddle<false>(tw, 8 * tp * kp));
#pragma unroll
        for (int kp = 0; kp < 8; kp++) e[k1 * ROW + 9 * kp + tp] = make_float2(v[kp].x, v[kp].y);
        exchange_fence();
        const int kq = lane & 7;                                         /* this lane's k1' */
#pragma unroll
        for (int t2 = 0; t2 < 8; t2++) { const float2 s = e[k1 * ROW + 9 * kq + t2]; v[t2] = {s.x, s.y}; }
        /* pass C: DFT8 over t' -> X[k1 + 8 k1' + 64 k2'] in register k2' */
        dft8<false>(v);
        /* ---- the filter: Y = X H (H carries GAIN and the 1/512 of the inverse transform) */
#pragma unroll
        for (int k2 = 0; k2 < 8; k2++) {
            const float2 h = Hf[k1 + 8 * kq + 64 * k2];
            v[k2] = cmul(v[k2], {h.x, h.y});
        }
        /* ---- inverse: pass C' over k2' -> index t', twiddle W64^-(t' k1'), back to lane (k1, t'), registers k1' */
        dft8<true>(v);
#pragma unroll
        for (int t2 = 1; t2 < 8; t2++) v[t2] = cmul(v[t2], twiddle<true>(tw, 8 * t2 * kq));
#pragma unroll
        for (int t2 = 0; t2 < 8; t2++) e[k1 * ROW + 9 * kq + t2] = make_float2(v[t2].x, v[t2].y);
        exchange_fence();
#pragma unroll
        for (int kp = 0; kp < 8; kp++) { const float2 s = e[k1 * ROW + 9 * kp + tp]; v[kp] = {s.x, s.y}; }
        /* pass B' over k1' -> index r', twiddle W512^-((t' + 8 r') k1), back to lane t = t' + 8 r', registers k1 */
        dft8<true>(v);
#pragma unroll
        for (int rp = 0; rp < 8; rp++) v[rp] = cmul(v[rp], twiddle<true>(tw, (tp + 8 * rp) * k1));
#pragma unroll
        for (int rp = 0; rp < 8; rp++) e[k1 * ROW + tp + 8 * rp] = make_float2(v[rp].x, v[rp].y);
    }
    exchange_fence();
#pragma unroll
    for (int k1 = 0; k1 < 8; k1++) { const float2 s = e[k1 * ROW + lane]; v[k1] = {s.x, s.y}; }
    /* pass A' over k1 -> y[t + 64 r] in register r */
    dft8<true>(v);
    /* ---- store the 386 valid points: segment index 126 .. 511 -> frame index 386 b .. */
    float2 *yf = y + (size_t)frame * length;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int i = lane + 64 * r - HIST, n = b * HOP + i;
        if (i >= 0 && n < length) yf[n] = make_float2(v[r].x, v[r].y);
    }
}

int rrc_fir_fast_hop(void) { return HOP; }
int rrc_fir_fast_nfft(void) { return NFFT; }

int launch_rrc_fir_fast(const float *x, const float *memory, float *y, const float *H, const float *tw, int nframes, int length,
                        hipStream_t s)
{
    const int nblocks = (length + HOP - 1) / HOP;
    const long long jobs = (long long)nframes * nblocks;
    const long long grid = (jobs + WAVES - 1) / WAVES;
    if (grid > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(rrc_fir_fast_kernel, dim3((unsigned)grid), dim3(64 * WAVES), 0, s, reinterpret_cast<const float2 *>(x),
                       reinterpret_cast<const float2 *>(memory), reinterpret_cast<float2 *>(y), reinterpret_cast<const float2 *>(H),
                       reinterpret_cast<const float2 *>(tw), nframes, length, nblocks);
    return (int)hipGetLastError();
}

} // namespace qpsk
