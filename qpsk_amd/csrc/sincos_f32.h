/*
 * sincos_f32.h -- single-precision sin/cos with the results of the C library
 * the reference links (glibc 2.35 sinf/cosf, called through cmplx()/cmplxconj(),
 * reference qpsk.h:35-36; Costas call site qpsk.c:197).
 *
 * ROCm's own sinf/cosf round differently, and the QPSK slicer decides symbols
 * that sit on a decision boundary, so the device evaluates the
 * library's published algorithm itself: argument in fp64, n = round(x * 2/pi)
 * by a 2^24-prescaled multiply and an integer shift, r = x - n*pi/2, then a
 * degree-7 odd / degree-8 even polynomial in r, narrowed to fp32.
 *
 * Form used here: ONE path for every |x| < 120 (the library's separate
 * "|x| < pi/4" and "|x| < 2^-12" branches give the same floats as the general
 * path: n = 0 and r = x there), multiply-adds fused (the library's x86-64 FMA
 * build).  tools/check_device_sincos.c runs exactly this header on the host
 * over every float in [-120, 120] against libm: 0 mismatches; on the Costas
 * domain [-2pi, 2pi] the non-FMA library build gives the same floats too.
 *
 * Domain: |x| < 120.  The Costas phase is wrapped to [-2pi, 2pi]
 * (costas_loop.c:61-67).
 */
#ifndef QPSK_SINCOS_F32_H
#define QPSK_SINCOS_F32_H

#if defined(__HIPCC__)
#define QPSK_HD __host__ __device__ __forceinline__
#else
#define QPSK_HD static inline
#endif

namespace qpsk {

struct SinCos {
    float s, c;
};

QPSK_HD SinCos sincos_f32(float y)
{
    const double x = (double)y;
    /* n = nearest integer to x*2/pi, via the 2^24-scaled product truncated to int32 */
    const double r = x * 0x1.45F306DC9C883p+23;
    const int n = ((int)r + 0x800000) >> 24;
    const double xr = __builtin_fma(-(double)n, 0x1.921FB54442D18p0, x);
    const double x2 = xr * xr;

    /* sine polynomial (odd in xr) */
    const double x3 = xr * x2;
    const double s1 = __builtin_fma(x2, -0x1.994eb3774cf24p-13, 0x1.1107605230bc4p-7);
    const double x7 = x3 * x2;
    const double sa = __builtin_fma(x3, -0x1.555545995a603p-3, xr);
    const double S = __builtin_fma(x7, s1, sa);

    /* cosine polynomial (even) */
    const double x4 = x2 * x2;
    const double c2 = __builtin_fma(x2, 0x1.99343027bf8c3p-16, -0x1.6c087e89a359dp-10);
    const double c1 = __builtin_fma(x2, -0x1.ffffffd0c621cp-2, 1.0);
    const double x6 = x4 * x2;
    const double ca = __builtin_fma(x4, 0x1.55553e1068f19p-5, c1);
    const double Cc = __builtin_fma(x6, c2, ca);

    /* quadrant: n&3 = 0:(S,C) 1:(C,-S) 2:(-S,-C) 3:(-C,S); negation is exact */
    const float fs = (float)S, fc = (float)Cc;
    SinCos o;
    const float a = (n & 1) ? fc : fs;
    const float b = (n & 1) ? fs : fc;
    o.s = (n & 2) ? -a : a;
    o.c = ((n + 1) & 2) ? -b : b;
    /* the library returns its argument for tiny |x|; the polynomial only loses the sign of -0 */
    if (y == 0.0f)
        o.s = y;
    return o;
}

/*
 * The same values with fewer instructions, for the Costas recurrence where
 * every instruction is on the serial path.  Differences from sincos_f32():
 *  - n = round-to-nearest(x * 2/pi) by adding 1.5*2^52 in the same fma that
 *    forms the product: the integer lands in the low mantissa bits (2 ops
 *    instead of mul / cvt / add / shift / cvt).  The library truncates a
 *    2^24-scaled product instead, which can pick the other neighbour when
 *    x * 2/pi is within 2^-24 of a half-integer; tools/check_device_sincos.cpp
 *    --costas proves that over EVERY float in [-2pi, 2pi] the final floats are
 *    identical (both candidates reduce to |r| ~ pi/4, where either polynomial
 *    pair rounds to the same float).
 *  - quadrant signs by integer arithmetic on the float bits.
 *  - sin(-0) = +0 instead of -0: the loop phase can only be -0 if the caller
 *    loads it (phase + freq + alpha*e never produces -0 from a +0 start), so
 *    the kernels canonicalise a loaded -0 phase through sincos_f32().
 * Domain: |x| <= 2pi as wrapped by costas_loop.c:61-67 (valid far beyond;
 * the checker sweeps |x| < 120).
 */
QPSK_HD SinCos sincos_f32_costas(float y)
{
    const double x = (double)y;
    const double MAGIC = 0x1.8p52;
    const double r2 = __builtin_fma(x, 0x1.45F306DC9C883p-1, MAGIC);
    const double nd = r2 - MAGIC;
    unsigned long long rb;
    __builtin_memcpy(&rb, &r2, 8);
    const unsigned n = (unsigned)rb; /* low mantissa word: n mod 2^32 */
    const double xr = __builtin_fma(-nd, 0x1.921FB54442D18p0, x);
    const double x2 = xr * xr;

    const double x3 = xr * x2;
    const double s1 = __builtin_fma(x2, -0x1.994eb3774cf24p-13, 0x1.1107605230bc4p-7);
    const double x7 = x3 * x2;
    const double sa = __builtin_fma(x3, -0x1.555545995a603p-3, xr);
    const float fs = (float)__builtin_fma(x7, s1, sa);

    const double x4 = x2 * x2;
    const double c2 = __builtin_fma(x2, 0x1.99343027bf8c3p-16, -0x1.6c087e89a359dp-10);
    const double c1 = __builtin_fma(x2, -0x1.ffffffd0c621cp-2, 1.0);
    const double x6 = x4 * x2;
    const double ca = __builtin_fma(x4, 0x1.55553e1068f19p-5, c1);
    const float fc = (float)__builtin_fma(x6, c2, ca);

    /* S' = S negated in quadrants 1,2; C' = C negated in quadrants 2,3;
     * even quadrant: (sin, cos) = (S', C'), odd: (C', S') */
    unsigned sb, cb;
    __builtin_memcpy(&sb, &fs, 4);
    __builtin_memcpy(&cb, &fc, 4);
    sb += ((n + 1u) & 2u) << 30;
    cb += (n & 2u) << 30;
    const unsigned ob = (n & 1u) ? cb : sb;
    const unsigned eb = (n & 1u) ? sb : cb;
    SinCos o;
    __builtin_memcpy(&o.s, &ob, 4);
    __builtin_memcpy(&o.c, &eb, 4);
    return o;
}

/*
 * Raw form for the serial wave of rx_fused_pipe_kernel: the two polynomial values on the reduced
 * argument (Horner evaluation, 8 fp64 operations after x^2) and the quadrant number, WITHOUT the
 * quadrant fix-up (the kernel applies it off the serial path, see costas_step_t in qpsk_device.h).
 * sincos_from_raw() is that fix-up; tools/check_device_sincos.cpp --horner checks the pair against
 * libm over every float in [-2pi, 2pi] (0 mismatches; the Horner rounding differs from the library's
 * evaluation order inside the double polynomial but never in the float result on that domain).
 */
struct SinCosRaw {
    float s, c;   /* sine / cosine polynomial on the reduced argument */
    unsigned n;   /* round(x * 2/pi) mod 2^32; quadrant = n & 3 */
};

QPSK_HD SinCosRaw sincos_raw_horner(float y)
{
    const double x = (double)y;
    const double MAGIC = 0x1.8p52;
    const double r2 = __builtin_fma(x, 0x1.45F306DC9C883p-1, MAGIC);
    const double nd = r2 - MAGIC;
    unsigned long long rb;
    __builtin_memcpy(&rb, &r2, 8);
    const double xr = __builtin_fma(-nd, 0x1.921FB54442D18p0, x);
    const double x2 = xr * xr;
    const double x3 = xr * x2;
    double u = __builtin_fma(x2, -0x1.994eb3774cf24p-13, 0x1.1107605230bc4p-7);
    double t = __builtin_fma(x2, 0x1.99343027bf8c3p-16, -0x1.6c087e89a359dp-10);
    u = __builtin_fma(x2, u, -0x1.555545995a603p-3);
    t = __builtin_fma(x2, t, 0x1.55553e1068f19p-5);
    SinCosRaw o;
    o.s = (float)__builtin_fma(x3, u, xr);
    t = __builtin_fma(x2, t, -0x1.ffffffd0c621cp-2);
    o.c = (float)__builtin_fma(x2, t, 1.0);
    o.n = (unsigned)rb;
    return o;
}

/*
 * The form the serial wave's instruction stream evaluates since round 6 (costas_asm.h, QPSK_HEAD_CHAIN / QPSK_BODY / QPSK_BODY_P):
 * n = rint(fl(x * 2/pi)) -- v_mul_f64 + v_rndne_f64, a 4-byte encoding, instead of the magic-number fma + subtract -- and the reduced
 * argument by a fused multiply-add IN PLACE on x (v_fmac_f64).  The product is rounded before the integer is taken, so this is not
 * the same computation as sincos_raw_horner()'s; tools/check_device_sincos.cpp --stream shows the same n, hence the same floats, for
 * every float with |x| <= 8 except -0 (n = -0 there and the reduced argument +0: the kernels keep a -0 phase away from the stream).
 * The polynomial stages are sincos_raw_horner()'s; the paired-lane body evaluates them with per-lane coefficients:
 *   cosine lane: M = fma(xr, 0, 1) = 1, r = fma(x2, c4, c3), A = x2 * M = x2, ..., C = fma(A, r, M)
 *   sine lane:   M = fma(xr, 1, -0) = xr, r = fma(x2, 0, s3) = s3, A = x2 * M = x3, ..., S = fma(A, r, M)
 * restated here stage by stage so that the host check covers exactly what the lanes execute.
 */
QPSK_HD SinCosRaw sincos_raw_stream(float y)
{
    const double x = (double)y;
    const double t = x * 0x1.45F306DC9C883p-1;
    const double nd = __builtin_rint(t);
    const double xr = __builtin_fma(-0x1.921FB54442D18p0, nd, x);
    const double x2 = xr * xr;
    /* even lane */
    const double mc = __builtin_fma(xr, 0.0, 1.0);
    double rc = __builtin_fma(x2, 0x1.99343027bf8c3p-16, -0x1.6c087e89a359dp-10);
    const double ac = x2 * mc;
    rc = __builtin_fma(x2, rc, 0x1.55553e1068f19p-5);
    rc = __builtin_fma(x2, rc, -0x1.ffffffd0c621cp-2);
    /* odd lane */
    const double ms = __builtin_fma(xr, 1.0, -0.0);
    double rs = __builtin_fma(x2, 0.0, -0x1.994eb3774cf24p-13);
    const double as = x2 * ms;
    rs = __builtin_fma(x2, rs, 0x1.1107605230bc4p-7);
    rs = __builtin_fma(x2, rs, -0x1.555545995a603p-3);
    SinCosRaw o;
    o.c = (float)__builtin_fma(ac, rc, mc);
    o.s = (float)__builtin_fma(as, rs, ms);
    o.n = (unsigned)(long long)nd;
    return o;
}

QPSK_HD SinCos sincos_from_raw(SinCosRaw r)
{
    /* quadrant 0: (S, C)  1: (C, -S)  2: (-S, -C)  3: (-C, S) */
    const float a = (r.n & 1u) ? r.c : r.s;
    const float b = (r.n & 1u) ? r.s : r.c;
    SinCos o;
    o.s = (r.n & 2u) ? -a : a;
    o.c = ((r.n + 1u) & 2u) ? -b : b;
    return o;
}

} // namespace qpsk
#endif
