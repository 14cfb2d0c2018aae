/*
 * check_device_sincos.cpp -- runs qpsk_amd/csrc/sincos_f32.h (the code the
 * Costas kernel executes) on the HOST over every float with |x| < limit and
 * compares with libm sinf/cosf (the reference's source of these values,
 * reference qpsk.h:35-36).  Exit 0 iff no bit differs.
 *
 *   g++ -O2 -ffp-contract=off -fopenmp tools/check_device_sincos.cpp -o /tmp/chkdev && /tmp/chkdev [limit]
 */
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../qpsk_amd/csrc/sincos_f32.h"

static inline uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

int main(int argc, char **argv)
{
    float limit = argc > 1 ? (float)atof(argv[1]) : 120.0f;
    uint32_t top = bits(limit);
    if (top >= bits(120.0f)) top = bits(120.0f) - 1;
    unsigned long long bad = 0, n = 0;
#pragma omp parallel for schedule(static, 1 << 16) reduction(+ : bad, n)
    for (uint32_t b = 0; b <= top; b++) {
        for (int sg = 0; sg < 2; sg++) {
            uint32_t u = b | ((uint32_t)sg << 31);
            float y; memcpy(&y, &u, 4);
            qpsk::SinCos r = qpsk::sincos_f32(y);
            n++;
            if (bits(r.s) != bits(sinf(y)) || bits(r.c) != bits(cosf(y))) bad++;
        }
    }
    printf("device-form sincos vs libm: limit=%a checked=%llu mismatches=%llu\n", limit, n, bad);
    return bad ? 1 : 0;
}
