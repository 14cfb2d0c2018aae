/*
 * oracle_sincosf.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Restatement of the single-precision sine/cosine that the reference obtains
 * from its C library.  The reference calls cosf()/sinf() through the cmplx()
 * and cmplxconj() macros (reference qpsk.h:35-36; call sites qpsk.c:75,197,
 * 316,320,341,342 and rrc_fir.c:46-49,62-64).  The library is glibc 2.35
 * (Ubuntu 2.35-0ubuntu3.11), whose source is NOT under /root/reference; what
 * follows restates its published algorithm (sysdeps/ieee754/flt-32/
 * s_sincosf.h, s_sinf.c, s_cosf.c, s_sincosf_data.c -- the "optimized
 * routines" single-precision sincos: argument promoted to double, one
 * multiply/round range reduction by pi/2 with a 2^24-prescaled 2/pi, degree-7
 * sine / degree-8 cosine polynomials, result narrowed to float).
 *
 * Pin: tools/check_sincosf.c compares these functions with the container's
 * libm sinf/cosf over EVERY float in [-120, 120] (the whole reduce_fast
 * domain; the Costas phase only reaches [-2pi, 2pi]) and must report zero
 * mismatches.  x86-64 glibc selects an FMA build of the same C source at run
 * time; ORACLE_SC_FMA=1 restates that build (contracted a*b+c), 0 the plain
 * one.  The checker decides which one libm here is.
 *
 * Only |x| < 120 is restated (the slow Payne-Hanek path of the library for
 * larger arguments is unreachable on the QPSK receive path: the phase is
 * wrapped to [-2pi, 2pi] (costas_loop.c:61-67) and rrc_make arguments stay
 * below 1.35*pi*63/spb).  Out-of-domain arguments return NaN so that a misuse
 * is loud.
 */
#ifndef ORACLE_SINCOSF_H
#define ORACLE_SINCOSF_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#ifndef ORACLE_SC_FMA
#define ORACLE_SC_FMA 1
#endif

#if ORACLE_SC_FMA
#define OSC_MADD(a, b, c) __builtin_fma((a), (b), (c))
#else
/* volatile-free unfused form; this header must be compiled -ffp-contract=off */
#define OSC_MADD(a, b, c) ((a) * (b) + (c))
#endif

static inline uint32_t osc_asuint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline uint32_t osc_abstop12(float f) { return (osc_asuint(f) >> 20) & 0x7ff; }

/* polynomial coefficients, table 0 of __sincosf_table */
#define OSC_HPI_INV 0x1.45F306DC9C883p+23 /* 2/pi * 2^24 */
#define OSC_HPI 0x1.921FB54442D18p0       /* pi/2 */
#define OSC_C0 0x1p0
#define OSC_C1 -0x1.ffffffd0c621cp-2
#define OSC_C2 0x1.55553e1068f19p-5
#define OSC_C3 -0x1.6c087e89a359dp-10
#define OSC_C4 0x1.99343027bf8c3p-16
#define OSC_S1 -0x1.555545995a603p-3
#define OSC_S2 0x1.1107605230bc4p-7
#define OSC_S3 -0x1.994eb3774cf24p-13

/* sine polynomial on the reduced argument (odd: S(-x) == -S(x) bit for bit) */
static inline double osc_sin_poly(double x, double x2)
{
    double x3 = x * x2;
    double s1 = OSC_MADD(x2, OSC_S3, OSC_S2);
    double x7 = x3 * x2;
    double s = OSC_MADD(x3, OSC_S1, x);
    return OSC_MADD(x7, s1, s);
}

/* cosine polynomial; table 1 of the library is this one with every
 * coefficient negated, i.e. exactly -osc_cos_poly(). */
static inline double osc_cos_poly(double x2)
{
    double x4 = x2 * x2;
    double c2 = OSC_MADD(x2, OSC_C4, OSC_C3);
    double c1 = OSC_MADD(x2, OSC_C1, OSC_C0);
    double x6 = x4 * x2;
    double c = OSC_MADD(x4, OSC_C2, c1);
    return OSC_MADD(x6, c2, c);
}

/* both results for one argument; *sn = sinf(y), *cs = cosf(y) */
static inline void oracle_sincosf(float y, float *sn, float *cs)
{
    double x = y;
    if (osc_abstop12(y) < osc_abstop12(0x1.921FB6p-1f)) { /* "|y| < pi/4" on the top 12 bits */
        double x2 = x * x;
        if (osc_abstop12(y) < osc_abstop12(0x1p-12f)) {
            *sn = y;
            *cs = 1.0f;
            return;
        }
        *sn = (float)osc_sin_poly(x, x2);
        *cs = (float)osc_cos_poly(x2);
        return;
    }
    if (osc_abstop12(y) < osc_abstop12(120.0f)) {
        double r = x * OSC_HPI_INV;
        int n = ((int32_t)r + 0x800000) >> 24;
#if ORACLE_SC_FMA
        double xr = __builtin_fma(-(double)n, OSC_HPI, x);
#else
        double xr = x - (double)n * OSC_HPI;
#endif
        /* sign[n&3] = {1,-1,-1,1}; (n&2) selects the negated cosine table */
        double sgn = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
        double xs = xr * sgn;
        double x2 = xr * xr;
        double S = osc_sin_poly(xs, x2);
        double C = (n & 2) ? -osc_cos_poly(x2) : osc_cos_poly(x2);
        if (n & 1) {
            *sn = (float)C;
            *cs = (float)S;
        } else {
            *sn = (float)S;
            *cs = (float)C;
        }
        return;
    }
    *sn = NAN;
    *cs = NAN;
}

static inline float oracle_sinf(float y) { float s, c; oracle_sincosf(y, &s, &c); return s; }
static inline float oracle_cosf(float y) { float s, c; oracle_sincosf(y, &s, &c); return c; }

#endif
