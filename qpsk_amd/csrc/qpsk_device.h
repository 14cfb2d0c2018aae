/*
 * qpsk_device.h -- per-sample / per-symbol arithmetic shared by every kernel.
 *
 * Everything here mirrors, operation for operation and in fp32 (fp64 where
 * the reference's C types promote), what gcc makes of the reference's
 * "complex float" expressions when built as its Makefile says (-std=c11, so
 * no FMA contraction, Makefile:7):
 *     real x complex  -> component-wise
 *     complex x complex -> (ac - bd, ad + bc)
 * The translation unit that includes this MUST be compiled -ffp-contract=off.
 */
#ifndef QPSK_DEVICE_H
#define QPSK_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sincos_f32.h"

namespace qpsk {

constexpr int NTAPS = 127;       /* rrc_fir.h:13 */
constexpr int HIST = NTAPS - 1;  /* samples of history one output needs */
constexpr double GAIN = 1.85;    /* rrc_fir.h:14, a double */
constexpr double TAU = 2.0 * 3.14159265358979323846; /* qpsk.h:29 */
constexpr float TAU_F = 0x1.921fb6p+2f; /* the float just above TAU: "phase > TAU" <=> phase >= TAU_F */
constexpr float ROT45 = 0x1.6a09e6p-1f; /* cosf((float)(M_PI/4)) == sinf(same), qpsk.h:30, qpsk.c:75 */

/* rrc_fir.c:28: "sample[j] = y * GAIN": complex float times double, narrowed */
__device__ __forceinline__ float2 fir_gain(float2 y)
{
    return make_float2((float)((double)y.x * GAIN), (float)((double)y.y * GAIN));
}

/* rrc_fir.c:25: y += memory[i] * coeffs[i] */
__device__ __forceinline__ void fir_mac(float2 &y, float2 m, float c)
{
    y.x = y.x + m.x * c;
    y.y = y.y + m.y * c;
}

struct Loop {
    float phase, freq;        /* costas_loop.c:13-14 */
};
struct LoopGains {
    float alpha, beta, min_freq, max_freq; /* costas_loop.c:16-23 */
};

/* costas_loop.c:61-67: float phase against the DOUBLE 2*pi */
__device__ __forceinline__ float phase_wrap(float p)
{
    while (p >= TAU_F)
        p = (float)((double)p - TAU);
    while (p <= -TAU_F)
        p = (float)((double)p + TAU);
    return p;
}

/* qpsk.c:74-79 with the natural symbol index (bits[1]<<1)|bits[0], qpsk.c:270 */
__device__ __forceinline__ int slicer(float2 z)
{
    const float rr = z.x * ROT45 - z.y * ROT45;
    const float ri = z.x * ROT45 + z.y * ROT45;
    return ((ri < 0.0f) ? 2 : 0) | ((rr < 0.0f) ? 1 : 0);
}

/* one iteration of qpsk.c:196-212: returns the de-rotated symbol, advances the loop.
 * EXACT_ZERO = true uses the sin/cos form that also reproduces sin(-0) = -0; the faster form is
 * identical for every other argument and a -0 phase can only be LOADED, never produced by the loop
 * (x + y is -0 only if both are), so callers run their first step with EXACT_ZERO = true. */
template <bool EXACT_ZERO = false>
__device__ __forceinline__ float2 costas_step(Loop &st, const LoopGains &g, float2 d)
{
    const SinCos w = EXACT_ZERO ? sincos_f32(st.phase) : sincos_f32_costas(st.phase);
    /* d * (cos - j sin)  (qpsk.c:197, qpsk.h:36) */
    float2 z;
    z.x = d.x * w.c + d.y * w.s;
    z.y = d.y * w.c - d.x * w.s;
    /* costas_loop.c:44-47, sgn(0) = -1 */
    const float e = (z.x > 0.0f ? z.y : -z.y) - (z.y > 0.0f ? z.x : -z.x);
    /* costas_loop.c:56-59 */
    st.freq = st.freq + g.beta * e;
    st.phase = phase_wrap(st.phase + st.freq + g.alpha * e);
    /* costas_loop.c:69-74 */
    if (st.freq > g.max_freq)
        st.freq = g.max_freq;
    else if (st.freq < g.min_freq)
        st.freq = g.min_freq;
    return z;
}

/*
 * The same step written for the serial wave of rx_fused_pipe_kernel, where every instruction is on the
 * critical path: no divergent branches (the detector and the clamp are selects, the phase wrap is taken
 * only when some lane of the wave needs it), sin/cos in the Costas form.  FAST_CLAMP (host-checked:
 * min_freq < 0 < max_freq) replaces the two compare/select pairs of costas_loop.c:69-74 by one median-of-3,
 * which returns the same float for every non-NaN frequency when neither bound is a zero.
 */
template <bool FAST_CLAMP>
__device__ __forceinline__ float2 costas_step_lean(float &phase, float &freq, float alpha, float beta,
                                                   float min_freq, float max_freq, float2 d)
{
    const SinCos w = sincos_f32_costas(phase);
    float2 z;
    z.x = d.x * w.c + d.y * w.s;
    z.y = d.y * w.c - d.x * w.s;
    const float e = (z.x > 0.0f ? z.y : -z.y) - (z.y > 0.0f ? z.x : -z.x);
    float f = freq + beta * e;
    float p = phase + f + alpha * e;
    if (__builtin_expect(__any(fabsf(p) >= TAU_F), 0))
        p = phase_wrap(p);
    if (FAST_CLAMP)
        f = __builtin_amdgcn_fmed3f(f, min_freq, max_freq);
    else
        f = f > max_freq ? max_freq : (f < min_freq ? min_freq : f);
    phase = p;
    freq = f;
    return z;
}

} // namespace qpsk
#endif
