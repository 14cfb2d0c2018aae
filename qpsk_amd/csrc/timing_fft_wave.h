/*
 * timing_fft_wave.h -- the FFT timing estimate of ONE frame by ONE wave (definition and method: timing_fft.hip's header),
 * shared by timing_fft_kernel and by the estimate that runs inside rx_fused_pipe_kernel's prologue (rx_fused.hip).
 */
#ifndef QPSK_TIMING_FFT_WAVE_H
#define QPSK_TIMING_FFT_WAVE_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "qpsk_device.h"

namespace qpsk {
namespace tfft {
constexpr int N0 = 128;        /* first output used: the delay line is primed after 126 samples; multiple of CYCLES */
constexpr int NFFT = 512;      /* NFFT of fft.h:44 */
constexpr int LOG2N = 9;
constexpr int R = 8, PADS = 2; /* fir_full8_asm.h: 8 consecutive outputs per lane, window position p at slot p + 2 (p / 8) */
constexpr int WPOS = NFFT + HIST;          /* 638 window positions: position p = sample p + N0 - HIST */
constexpr int WSLOTS = 808;                /* float2 slots per wave: the stream's last (unused) pair read ends at slot 798 */
constexpr int MAX_FPW = 4;                 /* frames a wave takes one after the other */
__device__ __host__ constexpr int slot_of(int p) { return p + PADS * (p / R); }
static_assert(slot_of(R * 63) + slot_of(NTAPS + R - 1 + 1) + 2 <= WSLOTS, "window: every slot the stream reads exists");
static_assert(NFFT == 64 * R && slot_of(WPOS - 1) < WSLOTS && (NFFT + NFFT / 8) * sizeof(double) <= WSLOTS * sizeof(float2), "one pass of the stream per frame; p[] fits the window");

struct cd { double x, y; };

/* one output of the butterfly of fft.c:55-63: e + w o (sg = +1: the node's bin k < m/2) or e - w o (sg = -1) */
__device__ __forceinline__ cd half_butterfly(cd e, cd o, double wr, double wi, double sg)
{
    const double zr = wr * o.x - wi * o.y;      /* fft.c:57 */
    const double zi = wr * o.y + wi * o.x;      /* fft.c:58 */
    cd r;
    r.x = e.x + sg * zr;                        /* fft.c:60-63; a - b and a + (-b) are the same operation */
    r.y = e.y + sg * zi;
    return r;
}

/* LDS is shared by the lanes of ONE wave here and a wave's LDS instructions execute in order: this only keeps the
 * compiler from moving them across */
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

/* samples N0 - HIST .. N0 + NFFT - 1 = 2 .. 639 of a frame as pairs: pair i (samples 2i, 2i+1), i = 1 .. 319, lane l takes
 * 1 + l + 64 j.  aligned: the frame starts on a 16-byte boundary */
__device__ __forceinline__ void load_frame(const float2 *src, int lane, int aligned, float4 (&pre)[5])
{
    if (aligned) {
        const float4 *s4 = reinterpret_cast<const float4 *>(src);
#pragma unroll
        for (int j = 0; j < 5; j++) pre[j] = s4[min(1 + lane + 64 * j, (N0 + NFFT) / 2 - 1)];
    } else {
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const int i = min(1 + lane + 64 * j, (N0 + NFFT) / 2 - 1);
            const float2 a = src[2 * i], b = src[2 * i + 1];
            pre[j] = make_float4(a.x, a.y, b.x, b.y);
        }
    }
}

struct Pre5 { float4 v0, v1, v2, v3, v4; };      /* the five pairs a lane holds of one frame (no array: stays in registers everywhere) */

__device__ __forceinline__ Pre5 load_frame5(const float2 *src, int lane)      /* 16-byte aligned frames */
{
    const float4 *s4 = reinterpret_cast<const float4 *>(src);
    Pre5 p;
    p.v0 = s4[1 + lane];
    p.v1 = s4[65 + lane];
    p.v2 = s4[129 + lane];
    p.v3 = s4[193 + lane];
    p.v4 = s4[min(257 + lane, (N0 + NFFT) / 2 - 1)];
    return p;
}

__device__ __forceinline__ void stage_frame5(float2 *win, int lane, const Pre5 &p)
{
    *reinterpret_cast<float4 *>(win + slot_of(2 * lane)) = p.v0;
    *reinterpret_cast<float4 *>(win + slot_of(2 * (lane + 64))) = p.v1;
    *reinterpret_cast<float4 *>(win + slot_of(2 * (lane + 128))) = p.v2;
    *reinterpret_cast<float4 *>(win + slot_of(2 * (lane + 192))) = p.v3;
    if (lane < 63) *reinterpret_cast<float4 *>(win + slot_of(2 * (lane + 256))) = p.v4;
}

/* window from registers: pair i sits at positions 2 (i - 1), 2 (i - 1) + 1 -- one aligned 16-byte word of the image */
__device__ __forceinline__ void stage_frame(float2 *win, int lane, const float4 (&pre)[5])
{
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const int p = 2 * (lane + 64 * j);
        if (j < 4 || lane < 63) *reinterpret_cast<float4 *>(win + slot_of(p)) = pre[j];
    }
}

/* From the lane's 8 accumulators of the stream (outputs 8 lane .. 8 lane + 7 before the second GAIN) to the symbol-rate bin
 * k0 = NFFT / CYCLES of fftn(|y|^2) BEFORE its division by NFFT, valid in lane 0.  pv[]: the lane's |y|^2 (for the variant that
 * also runs the whole transform); yout_f: the frame's 512 filtered samples, or NULL.  win: the wave's window (the stream has
 * finished with it); the next staging may follow at once. */
__device__ __forceinline__ cd power_bin(const v2f (&acc)[R], float2 *win, int lane, const double2 *tw, int k0, float2 *yout_f,
                                        double (&pv)[R])
{
#pragma unroll
    for (int r = 0; r < R; r++) {
        const float2 y = fir_gain(make_float2(acc[r].x, acc[r].y));          /* rrc_fir.c:28 */
        if (yout_f) yout_f[R * lane + r] = y;                                  /* the estimator's view of rrc_fir(): tests compare it with rrc_fir_kernel */
        const double pr = (double)y.x * (double)y.x, pi = (double)y.y * (double)y.y;
        pv[r] = pr + pi;
    }
    /* transposition through the window's LDS: element o at double slot o + o/8 */
    double *tp = reinterpret_cast<double *>(win);
#pragma unroll
    for (int r = 0; r < R; r++) tp[(R + 1) * lane + r] = pv[r];
    wave_sync();
    cd v[R];
#pragma unroll
    for (int j = 0; j < R; j++) {
        const int o = lane + 64 * j;
        v[j].x = tp[o + (o >> 3)];
        v[j].y = 0.0;
    }
    wave_sync();      /* the next frame's staging overwrites tp[] */
    /* level s: nodes of size m = 2^s, the bin each must deliver is kb = k0 mod m: its twiddle index kb mod m/2 in a size-m
     * transform = entry (kb mod m/2) * (NFFT / m) of the size-NFFT table (bit-identical: scaling an angle by a power of two is exact) */
    /* (the nine twiddles are re-read per frame through the scalar cache: hoisted out of a frame loop they would hold 36
     * SGPRs across a stream that owns 64 of the wave's ~100) */
    const double2 *twp = tw;
    int k0v = k0;
    asm volatile("" : "+s"(twp), "+s"(k0v));
    auto level = [&](int s, double &wr, double &wi, double &sg) {
        const int m = 1 << s, kb = k0v & (m - 1), kk = kb & (m / 2 - 1);
        const double2 w = twp[kk << (LOG2N - s)];
        wr = w.x;
        wi = -1.0 * w.y;                   /* forward: w = cos - j sin (fft.c:55-56) */
        sg = kb >= m / 2 ? -1.0 : 1.0;
    };
    double wr, wi, sg;
    level(1, wr, wi, sg);
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = half_butterfly(v[j], v[j + 4], wr, wi, sg);      /* id, id + 256 */
    level(2, wr, wi, sg);
#pragma unroll
    for (int j = 0; j < 2; j++) v[j] = half_butterfly(v[j], v[j + 2], wr, wi, sg);      /* id, id + 128 */
    level(3, wr, wi, sg);
    cd u = half_butterfly(v[0], v[1], wr, wi, sg);                                       /* id, id + 64: id = lane */
#pragma unroll
    for (int s = 4; s <= LOG2N; s++) {
        const int delta = 32 >> (s - 4);                                                 /* id, id + 512 / 2^s */
        cd o;
        o.x = __shfl_down(u.x, delta);
        o.y = __shfl_down(u.y, delta);
        level(s, wr, wi, sg);
        u = half_butterfly(u, o, wr, wi, sg);
    }
    return u;
}

/* the estimator's own rule (one lane): X_k = u / NFFT (fft.c:117-119), index = first i with the largest Re(X_k e^{+j 2 pi i / CYCLES}) */
__device__ __forceinline__ int pick_index(cd u, const double2 *cs, int cycles, double2 *xk_out)
{
    const double dn = (double)NFFT;
    const double xr = u.x / dn, xi = u.y / dn;
    if (xk_out) *xk_out = make_double2(xr, xi);
    int best = 0;
    double hmax = xr * cs[0].x - xi * cs[0].y;
    for (int i = 1; i < cycles; i++) {
        const double c = xr * cs[i].x - xi * cs[i].y;
        if (c > hmax) { hmax = c; best = i; }
    }
    return best;
}

} // namespace tfft
} // namespace qpsk
#endif
