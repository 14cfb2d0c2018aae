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
#include "kernels.h"

namespace qpsk {

constexpr int NTAPS = 127;       /* rrc_fir.h:13 */
constexpr int HIST = NTAPS - 1;  /* samples of history one output needs */
constexpr double GAIN = 1.85;    /* rrc_fir.h:14, a double */
constexpr double TAU = 2.0 * 3.14159265358979323846; /* qpsk.h:29 */
constexpr float TAU_F = 0x1.921fb6p+2f; /* the float just above TAU: "phase > TAU" <=> phase >= TAU_F */
constexpr float ROT45 = 0x1.6a09e6p-1f; /* cosf((float)(M_PI/4)) == sinf(same), qpsk.h:30, qpsk.c:75 */

/* tools of the hand-ordered FIR loops (rx_fused.hip, rrc_fir_kernel) */
typedef float v2f __attribute__((ext_vector_type(2)));   /* one VGPR pair: operand type of the packed fp32 ops */

template <int I>
struct IntC { static constexpr int value = I; };

/* f(IntC<I>{}) for I = FIRST .. LAST-1, expanded at compile time (the compiler does not unroll a loop with an
 * asm statement in it) */
template <int FIRST, int LAST, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (FIRST < LAST) {
        f(IntC<FIRST>{});
        static_for<FIRST + 1, LAST>(f);
    }
}

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

/* costas_loop.c:61-67: float phase against the DOUBLE 2*pi.
 * The reference's loops are unbounded: a phase of 2^27 or more never moves ((float)((double)p - TAU) == p) and the
 * CPU process hangs; below that it takes |p| / TAU iterations.  A GPU wave must not spin like that, so the loops
 * stop after WRAP_LIMIT iterations (|phase| > ~25,000 rad: inputs some 10^5 times the modem's working amplitude),
 * `over` is raised -- the kernel reports it through its status word and the call fails with QPSK_ERR_RANGE -- and
 * the phase restarts from 0 so that the rest of the frame runs at its normal cost.  Every phase the reference
 * wraps in at most WRAP_LIMIT steps is wrapped exactly as it does. */
constexpr int WRAP_LIMIT = 4096;

__device__ __forceinline__ float phase_wrap(float p, bool &over)
{
    int n = 0;
    while (p >= TAU_F) {
        p = (float)((double)p - TAU);
        if (++n > WRAP_LIMIT) { over = true; return 0.0f; }
    }
    while (p <= -TAU_F) {
        p = (float)((double)p + TAU);
        if (++n > WRAP_LIMIT) { over = true; return 0.0f; }
    }
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
__device__ __forceinline__ float2 costas_step(Loop &st, const LoopGains &g, float2 d, bool &over)
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
    st.phase = phase_wrap(st.phase + st.freq + g.alpha * e, over);
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
                                                   float min_freq, float max_freq, float2 d, bool &over)
{
    const SinCos w = sincos_f32_costas(phase);
    float2 z;
    z.x = d.x * w.c + d.y * w.s;
    z.y = d.y * w.c - d.x * w.s;
    const float e = (z.x > 0.0f ? z.y : -z.y) - (z.y > 0.0f ? z.x : -z.x);
    float f = freq + beta * e;
    float p = phase + f + alpha * e;
    if (__builtin_expect(__any(fabsf(p) >= TAU_F), 0))
        p = phase_wrap(p, over);
    if (FAST_CLAMP)
        f = __builtin_amdgcn_fmed3f(f, min_freq, max_freq);
    else
        f = f > max_freq ? max_freq : (f < min_freq ? min_freq : f);
    phase = p;
    freq = f;
    return z;
}

/*
 * The shortest form of the step, for the serial wave of rx_fused_pipe_kernel.  Three exact rewrites:
 *
 * 1. sin/cos polynomials in Horner form (8 fp64 ops instead of 11).  Rounding differs inside the double
 *    evaluation, yet tools/check_device_sincos.cpp --horner shows the FLOAT results equal libm's for every
 *    float in [-2pi, 2pi] (2,173,837,239 arguments, 0 mismatches).
 * 2. The quadrant of the phase is NOT applied on the serial path.  With S, C the raw polynomial values
 *    (sine/cosine of the reduced argument) and q the quadrant, the de-rotated symbol of qpsk.c:197 is
 *        z = T * (-j)^q,   T = d * (C - jS)
 *    bit for bit (negation and swapping commute with every rounding involved), and the QPSK detector
 *    sgn(I)Q - sgn(Q)I (costas_loop.c:44-47) takes the same value on T as on z -- it is invariant under
 *    quarter turns -- except when a component of T is exactly zero, where the reference's sgn(0) = -1
 *    breaks the symmetry.  So the wave computes T and e(T); if any lane has T.x*T.y == 0 (exact zero, or
 *    an underflowing product: a superset) the wave redoes that step through z.  The consumer of the
 *    (T, q) record -- the FIR waves' flush -- forms z for the slicer and for costas_frame[].
 * 3. FAST_CLAMP as in costas_step_lean().
 *
 * Returns T in (tx, ty) and the quadrant bits in q (low two bits significant).
 */
/* z = T * (-j)^q:  q = 0: (T.x, T.y)   1: (T.y, -T.x)   2: (-T.x, -T.y)   3: (-T.y, T.x) */
__device__ __forceinline__ float2 apply_quadrant(float tx, float ty, unsigned q)
{
    const float a = (q & 1u) ? ty : tx;
    const float b = (q & 1u) ? tx : ty;
    float2 z;
    z.x = (q & 2u) ? -a : a;
    z.y = ((q + 1u) & 2u) ? -b : b;
    return z;
}

__device__ __forceinline__ float detector(float zx, float zy)
{
    return (zx > 0.0f ? zy : -zy) - (zy > 0.0f ? zx : -zx);
}

/*
 * costas_frame[] of one symbol (qpsk.c:197) from the phase the loop held BEFORE that symbol's step -- what the
 * consumer of a phase record computes: the library's sin/cos of that phase (raw polynomials, then the quadrant
 * applied to the PAIR (sin, cos), which costs what applying it to the product would), then the reference's four
 * products and two sums as they stand.  (The serial wave itself forms T = d (C - jS) on the raw values and leaves
 * the quadrant out -- its detector does not need it; turning T instead of (sin, cos) here would give the same
 * numbers except for the SIGN OF AN EXACT ZERO: x + (-x) is +0 whichever way round, so negating a sum that
 * cancelled does not negate it.  A zero symbol -- the pick past the end of a block at CYCLES = 4, SURVEY Q5 -- met
 * exactly that; tests/test_abi.py::test_streams_rx_pcm_host_equals_the_device_pointer_calls.)
 * FIRST = the frame's first symbol: its phase may be a loaded -0, so it takes the form that is exact there too,
 * like the step that consumed it.
 */
template <bool FIRST>
__device__ __forceinline__ float2 derotate(float phase, float2 d)
{
    const SinCos w = FIRST ? sincos_f32(phase) : sincos_from_raw(sincos_raw_horner(phase));
    return make_float2(d.x * w.c + d.y * w.s, d.y * w.c - d.x * w.s);
}

template <bool FAST_CLAMP>
__device__ __forceinline__ void costas_step_t(float &phase, float &freq, float alpha, float beta, float min_freq,
                                              float max_freq, float2 d, float &tx, float &ty, unsigned &q, bool &over)
{
    const SinCosRaw w = sincos_raw_horner(phase);
    tx = d.x * w.c + d.y * w.s;
    ty = d.y * w.c - d.x * w.s;
    q = w.n;
    float e = detector(tx, ty);
    float f = freq + beta * e;
    float p = phase + f + alpha * e;
    const bool zero = tx * ty == 0.0f;
    const bool wrap = fabsf(p) >= TAU_F;
    if (__builtin_expect(__any(zero || wrap), 0)) {
        if (zero) {
            /* sgn(0) = -1 is not rotation invariant: take the detector on z itself -- formed as derotate() forms it, from the turned
             * PAIR (sin, cos): turning T instead gives the same numbers except for the sign of a sum that cancelled to zero (see
             * there), and that sign reaches the loop when its frequency is a zero too (a loaded -0 frequency in front of zero symbols) */
            const SinCos sc = sincos_from_raw(w);
            const float2 z = make_float2(d.x * sc.c + d.y * sc.s, d.y * sc.c - d.x * sc.s);
            e = detector(z.x, z.y);
            f = freq + beta * e;
            p = phase + f + alpha * e;
        }
        p = phase_wrap(p, over);
    }
    if (FAST_CLAMP)
        f = __builtin_amdgcn_fmed3f(f, min_freq, max_freq);
    else
        f = f > max_freq ? max_freq : (f < min_freq ? min_freq : f);
    phase = p;
    freq = f;
}

/* a 16-byte load of data that is read ONCE (the input samples): nontemporal, i.e. not parked in the caches on its way --
 * worth 1.5 % of rx_lean_kernel's time at the board's power limit */
__device__ __forceinline__ float4 load_once(const float4 *p)
{
    typedef float v4f_ __attribute__((ext_vector_type(4)));
    const v4f_ v = __builtin_nontemporal_load(reinterpret_cast<const v4f_ *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}

/* NaN / Inf inputs are fenced, not reproduced: a non-finite sample reaches its loop through the filter and the loop state
 * then stays non-finite to the end of the frame (phase = phase + ...), where the kernel flags the call (STATUS_NONFINITE ->
 * QPSK_ERR_RANGE).  The reference itself spins forever in phase_wrap() on an infinite phase (costas_loop.c:61-67) and goes
 * through __mulsc3's recovery on NaN products (qpsk.c:197), which this library does not restate. */
__device__ __forceinline__ bool loop_state_finite(float phase, float freq)
{
    return fabsf(phase) <= 3.402823466e+38f && fabsf(freq) <= 3.402823466e+38f;
}

} // namespace qpsk
#endif
