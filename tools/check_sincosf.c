/*
 * check_sincosf.c -- pins oracle/oracle_sincosf.h against the C library the
 * reference links (glibc libm sinf/cosf, reference qpsk.h:35-36).
 *
 * Walks every float bit pattern with |x| <= LIMIT (default 120.0f = the whole
 * fast-reduction domain of the library routine) in both signs and counts the
 * arguments where the restatement and libm differ in any bit.
 *
 *   gcc -O2 -ffp-contract=off -fopenmp -DORACLE_SC_FMA=1 tools/check_sincosf.c -o /tmp/chk -lm
 *   /tmp/chk [limit]
 *
 * Exit status 0 iff zero mismatches.
 */
#include <stdio.h>
#include <stdlib.h>
#include "../oracle/oracle_sincosf.h"

int main(int argc, char **argv)
{
    float limit = argc > 1 ? (float)atof(argv[1]) : 120.0f;
    uint32_t top = osc_asuint(limit);
    if (osc_abstop12(limit) >= osc_abstop12(120.0f))
        top = osc_asuint(120.0f) - 1; /* largest float strictly inside the domain test */
    unsigned long long bad_s = 0, bad_c = 0, n = 0;
    uint32_t first_bad = 0;
    int have_first = 0;
#pragma omp parallel for schedule(static, 1 << 16) reduction(+ : bad_s, bad_c, n)
    for (uint32_t b = 0; b <= top; b++) {
        for (int sg = 0; sg < 2; sg++) {
            uint32_t u = b | ((uint32_t)sg << 31);
            float y;
            memcpy(&y, &u, 4);
            float s, c;
            oracle_sincosf(y, &s, &c);
            float ls = sinf(y), lc = cosf(y);
            n++;
            if (osc_asuint(s) != osc_asuint(ls)) {
                bad_s++;
#pragma omp critical
                if (!have_first) { have_first = 1; first_bad = u; }
            }
            if (osc_asuint(c) != osc_asuint(lc)) {
                bad_c++;
#pragma omp critical
                if (!have_first) { have_first = 1; first_bad = u; }
            }
        }
    }
    printf("variant ORACLE_SC_FMA=%d limit=%a checked=%llu sin_mismatch=%llu cos_mismatch=%llu\n",
           ORACLE_SC_FMA, limit, n, bad_s, bad_c);
    if (have_first) {
        float y;
        memcpy(&y, &first_bad, 4);
        float s, c;
        oracle_sincosf(y, &s, &c);
        printf("example x=%a: oracle sin=%a cos=%a  libm sin=%a cos=%a\n", y, s, c, sinf(y), cosf(y));
    }
    return (bad_s || bad_c) ? 1 : 0;
}
