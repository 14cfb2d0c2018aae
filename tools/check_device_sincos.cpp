/*
 * check_device_sincos.cpp -- runs qpsk_amd/csrc/sincos_f32.h (the code the
 * kernels execute) on the HOST over every float with |x| < limit and compares
 * with libm sinf/cosf (the reference's source of these values, reference
 * qpsk.h:35-36).  Exit 0 iff no bit differs.
 *
 *   g++ -O2 -ffp-contract=off -fopenmp tools/check_device_sincos.cpp -o /tmp/chkdev
 *   /tmp/chkdev [limit]            sincos_f32()         (general form)
 *   /tmp/chkdev --costas [limit]   sincos_f32_costas()  (the Costas-loop form; -0 excluded, see header)
 *   /tmp/chkdev --horner [limit]   sincos_raw_horner() + sincos_from_raw()  (the pipeline kernel's form; -0 excluded)
 *   /tmp/chkdev --stream [limit]   sincos_raw_stream() + sincos_from_raw()  (the serial wave's instruction stream since round 6: rint of the
 *                                  rounded product, per-lane polynomial stages; -0 excluded) -- also against sincos_raw_horner()'s n
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
    bool costas = false, horner = false, stream = false;
    int ai = 1;
    if (argc > 1 && !strcmp(argv[1], "--costas")) { costas = true; ai = 2; }
    if (argc > 1 && !strcmp(argv[1], "--horner")) { costas = horner = true; ai = 2; }
    if (argc > 1 && !strcmp(argv[1], "--stream")) { costas = stream = true; ai = 2; }
    float limit = argc > ai ? (float)atof(argv[ai]) : 120.0f;
    uint32_t top = bits(limit);
    if (top >= bits(120.0f)) top = bits(120.0f) - 1;
    unsigned long long bad = 0, n = 0;
    uint32_t ex = 0;
#pragma omp parallel for schedule(static, 1 << 16) reduction(+ : bad, n)
    for (uint32_t b = 0; b <= top; b++) {
        for (int sg = 0; sg < 2; sg++) {
            uint32_t u = b | ((uint32_t)sg << 31);
            if (costas && u == 0x80000000u) continue; /* -0: documented exception */
            float y; memcpy(&y, &u, 4);
            qpsk::SinCos r = stream ? qpsk::sincos_from_raw(qpsk::sincos_raw_stream(y))
                             : horner ? qpsk::sincos_from_raw(qpsk::sincos_raw_horner(y))
                                      : costas ? qpsk::sincos_f32_costas(y) : qpsk::sincos_f32(y);
            n++;
            if (bits(r.s) != bits(sinf(y)) || bits(r.c) != bits(cosf(y))) { bad++; ex = u; }
            else if (stream) {      /* the raw pair too: the flush of the FIR waves recomputes it with sincos_raw_horner() from the recorded phase */
                const qpsk::SinCosRaw a = qpsk::sincos_raw_stream(y), b = qpsk::sincos_raw_horner(y);
                if (bits(a.s) != bits(b.s) || bits(a.c) != bits(b.c) || ((a.n ^ b.n) & 3u)) { bad++; ex = u; }
            }
        }
    }
    printf("%s vs libm: limit=%a checked=%llu mismatches=%llu\n", stream ? "sincos_raw_stream" : horner ? "sincos_raw_horner" : costas ? "sincos_f32_costas" : "sincos_f32", limit, n, bad);
    if (bad) { float y; memcpy(&y, &ex, 4); qpsk::SinCos r = costas ? qpsk::sincos_f32_costas(y) : qpsk::sincos_f32(y);
        printf("example x=%a: got s=%a c=%a  libm s=%a c=%a\n", y, r.s, r.c, sinf(y), cosf(y)); }
    return bad ? 1 : 0;
}
