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
 * libm sinf/cosf over EVERY float bit pattern -- all finite arguments of both
 * signs (the fast reduction below 120, the large-argument reduction above),
 * the infinities and the NaNs -- and must report zero mismatches.  x86-64
 * glibc selects an FMA build of the same C source at run time;
 * ORACLE_SC_FMA=1 restates that build (contracted a*b+c), 0 the plain one.
 * The checker decides which one libm here is.
 *
 * Both of the library's reductions are restated.  The Costas phase only
 * reaches [-2pi, 2pi] (costas_loop.c:61-67), but rrc_make() does reach the
 * large-argument one: its arguments go up to (1+alpha)*pi*63*RS/FS
 * (rrc_fir.c:46-49,62-64), which passes 120 from FS/RS <= 2.2 at alpha = .35
 * (round 4's restatement stopped at 120 and answered NaN there; the
 * reference's taps are finite).  Infinities and NaNs give NaN, as the
 * library's __math_invalidf does (the sign of that NaN is not pinned).
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

/*
 * The library's reduce_large(): |y| >= 120.  The mantissa of y (24 bits, shifted left by the low three bits of the
 * exponent) times a 96-bit window of 2/pi, picked by the exponent's upper bits from a table of 32-bit words that slide
 * one BYTE at a time over the bits of 2/pi; the 64-bit fixed-point product counts quarter turns in units of 2^-62: its top
 * two bits, rounded to nearest, are the quadrant, and what is left, a signed number in [-2^61, 2^61], times
 * (pi/2) * 2^-62 is the reduced argument in [-pi/4, pi/4].  The sign of y is ignored here; the callers fold it into the
 * quadrant.
 */
static const uint32_t osc_inv_pio4[24] = {
    /* the library calls it __inv_pio4; 2/pi = 0.a2f9836e 4e441529 fc2757d1 f534ddc0 db629599 3c439041 (hex) read through
     * a 32-bit window at byte offsets -3, -2, ... 20 (tests/test_sincos.py recomputes the digits with integer arithmetic) */
    0x000000a2, 0x0000a2f9, 0x00a2f983, 0xa2f9836e, 0xf9836e4e, 0x836e4e44, 0x6e4e4415, 0x4e441529,
    0x441529fc, 0x1529fc27, 0x29fc2757, 0xfc2757d1, 0x2757d1f5, 0x57d1f534, 0xd1f534dd, 0xf534ddc0,
    0x34ddc0db, 0xddc0db62, 0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43, 0x993c4390, 0x3c439041,
};
#define OSC_PI63 0x1.921FB54442D18p-62 /* (pi/2) * 2^-62: the library's pi63 = 2 pi / 2^64 */

static inline double osc_reduce_large(uint32_t xi, int *np)
{
    const uint32_t *arr = &osc_inv_pio4[(xi >> 26) & 15];
    const int shift = (xi >> 23) & 7;
    uint64_t n, res0, res1, res2;

    xi = (xi & 0xffffff) | 0x800000;
    xi <<= shift;

    res0 = (uint32_t)(xi * arr[0]); /* 32-bit product: the bits above are whole turns */
    res1 = (uint64_t)xi * arr[4];
    res2 = (uint64_t)xi * arr[8];
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;

    n = (res0 + (1ULL << 61)) >> 62;
    res0 -= n << 62;
    *np = (int)n;
    return (double)(int64_t)res0 * OSC_PI63;
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
    if (osc_abstop12(y) < osc_abstop12(INFINITY)) {
        /* s_sinf.c / s_cosf.c, last finite branch.  sinf: the sign bit of y is ADDED to the quadrant that picks the result's
         * sign and the negated table (sin is odd), not to the one that picks sine or cosine polynomial */
        const uint32_t xi = osc_asuint(y);
        const int sign = (int)(xi >> 31);
        int n;
        const double xr = osc_reduce_large(xi, &n);
        const int ns = n + sign;
        /* sinf: sign[(n + sign) & 3], the negated table if (n + sign) & 2 */
        {
            const double sg = ((ns & 3) == 1 || (ns & 3) == 2) ? -1.0 : 1.0;
            const double x2 = xr * xr;
            double r;
            if (n & 1)
                r = (ns & 2) ? -osc_cos_poly(x2) : osc_cos_poly(x2);
            else
                r = osc_sin_poly(xr * sg, x2);
            *sn = (float)r;
        }
        /* cosf: sign[n & 3], the negated table if n & 2 (cos is even: the sign of y plays no part), polynomial by n ^ 1 */
        {
            const double sg = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
            const double x2 = xr * xr;
            double r;
            if ((n ^ 1) & 1)
                r = (n & 2) ? -osc_cos_poly(x2) : osc_cos_poly(x2);
            else
                r = osc_sin_poly(xr * sg, x2);
            *cs = (float)r;
        }
        return;
    }
    *sn = NAN; /* inf, NaN: __math_invalidf */
    *cs = NAN;
}

static inline float oracle_sinf(float y) { float s, c; oracle_sincosf(y, &s, &c); return s; }
static inline float oracle_cosf(float y) { float s, c; oracle_sincosf(y, &s, &c); return c; }

#endif
