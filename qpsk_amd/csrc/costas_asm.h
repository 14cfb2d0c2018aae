/*
 * costas_asm.h -- the Costas recurrence (qpsk.c:197-207 + costas_loop.c:44-74) as a hand-scheduled gfx950
 * instruction stream, for the serial wave of rx_fused_pipe_kernel.
 *
 * Why assembly: the recurrence runs in ONE wave per workgroup and that wave is strictly in order.  Measured
 * on MI355X (tools/ubench.hip): a lone wave issues one VALU instruction per ~5 cycles, a dependent one ~8
 * cycles after its producer.  A step is a chain of ~19 dependent operations (phase -> range reduction ->
 * cosine Horner chain -> rotate -> detector -> loop update -> phase) plus ~18 off-chain ones; its duration
 * is decided by how the off-chain operations are slotted between the dependent ones, which the compiler's
 * scheduler does not model (its version ran ~355 cycles per step).
 *
 * Arithmetic = costas_step_t() in qpsk_device.h operation for operation: Horner sin/cos polynomials in fp64
 * with fused multiply-adds (the library's form), everything in fp32 unfused (the reference is built without
 * contraction), the 2*pi wrap in fp64 (costas_loop.c:61-67), the clamp as a median of 3 (callers use this
 * stream only when min_freq < 0 < max_freq, where it equals costas_loop.c:69-74).  The parity tests compare
 * the kernel with the oracle bit for bit.
 *
 * One call runs `groups` groups of 8 steps.  Per step it reads the next decimated symbol d from LDS one
 * step ahead, and writes T = d*(C - jS) (8 bytes) and the quadrant byte; LDS operations complete in order,
 * so `s_waitcnt lgkmcnt(2)` before the first use of d leaves only the previous step's two writes in flight.
 * Cases the stream does not handle set a flag, and the group they occur in is abandoned with the loop state
 * restored to the group's start; the caller redoes that group with costas_step_t() and continues:
 *     T.x*T.y == 0   (the detector's sgn(0) = -1 asymmetry, see costas_step_t)
 *     a phase still outside [-2pi, 2pi] after ONE wrap (clamp wider than +-2pi, huge amplitudes).
 *
 * Registers: v[200:231] are scratch owned by the block (clobbered):
 *   200:201 x / d*C      202:203 magic sum (v202 bits 1:0 = quadrant)   204:205 n / x3 / d*S
 *   206:207 xr / e, zz   208:209 x2 / a, b     210:211 cos chain (v210 = C)   212:213 sin chain (v212 = S)
 *   214:215 T            216..219 beta*e, alpha*e, f2, p+f2             220:221 next d
 *   222..225 phase/freq ping-pong    226 unused   227 2pi hi   228:229 +-2pi   230,231 group-start phase/freq
 */
#ifndef QPSK_COSTAS_ASM_H
#define QPSK_COSTAS_ASM_H

#include "qpsk_device.h"

namespace qpsk {

/* 32-bit LDS byte address of a __shared__ object (what ds_read / ds_write take) */
__device__ __forceinline__ unsigned lds_addr(const void *p)
{
    return (unsigned)(__UINTPTR_TYPE__)(const __attribute__((address_space(3))) void *)p;
}

#define QPSK_STR_(x) #x
#define QPSK_STR(x) QPSK_STR_(x)

/* one step: PIN/FIN -> POUT/FOUT; DOFF = byte offset of the NEXT symbol, ZOFF/QOFF of this step's records */
#define QPSK_COSTAS_STEP(PIN, FIN, POUT, FOUT, DOFF, ZOFF, QOFF)                                              \
    "v_cvt_f64_f32 v[200:201], " PIN "\n\t"                                                                   \
    "v_fma_f64 v[202:203], v[200:201], %[k2pi], %[magic]\n\t"                                                 \
    "v_add_f64 v[204:205], v[202:203], -%[magic]\n\t"                                                         \
    "v_fma_f64 v[206:207], -v[204:205], %[hpi], v[200:201]\n\t"                                               \
    "v_mul_f64 v[208:209], v[206:207], v[206:207]\n\t"                                                        \
    "v_fma_f64 v[210:211], v[208:209], %[c4], %[c3]\n\t"                                                      \
    "v_fma_f64 v[212:213], v[208:209], %[s3], %[s2]\n\t"                                                      \
    "v_fma_f64 v[210:211], v[208:209], v[210:211], %[c2]\n\t"                                                 \
    "v_mul_f64 v[204:205], v[206:207], v[208:209]\n\t"                                                        \
    "v_fma_f64 v[210:211], v[208:209], v[210:211], %[c1]\n\t"                                                 \
    "v_fma_f64 v[212:213], v[208:209], v[212:213], %[s1]\n\t"                                                 \
    "v_fma_f64 v[210:211], v[208:209], v[210:211], 1.0\n\t"                                                   \
    "v_fma_f64 v[212:213], v[204:205], v[212:213], v[206:207]\n\t"                                            \
    "v_cvt_f32_f64 v210, v[210:211]\n\t"                                                                      \
    "v_cvt_f32_f64 v212, v[212:213]\n\t"                                                                      \
    "s_waitcnt lgkmcnt(2)\n\t"                                                                                \
    "v_pk_mul_f32 v[200:201], v[220:221], v[210:211] op_sel_hi:[1,0]\n\t"                                     \
    "v_pk_mul_f32 v[204:205], v[220:221], v[212:213] op_sel:[1,0] op_sel_hi:[0,0]\n\t"                        \
    "ds_read_b64 v[220:221], %[da] offset:" QPSK_STR(DOFF) "\n\t"                                             \
    "v_pk_add_f32 v[214:215], v[200:201], v[204:205] neg_hi:[0,1]\n\t"                                        \
    "v_cmp_lt_f32_e32 vcc, 0, v214\n\t"                                                                       \
    "v_cmp_lt_f32_e64 %[tm], 0, v215\n\t"                                                                     \
    "v_mul_f32_e32 v207, v214, v215\n\t"                                                                      \
    "v_cndmask_b32_e64 v208, -v215, v215, vcc\n\t"                                                            \
    "v_cndmask_b32_e64 v209, -v214, v214, %[tm]\n\t"                                                          \
    "v_sub_f32_e32 v206, v208, v209\n\t"                                                                      \
    "v_cmp_eq_f32_e64 %[tm], 0, v207\n\t"                                                                     \
    "v_mul_f32_e32 v216, %[be], v206\n\t"                                                                     \
    "v_mul_f32_e32 v217, %[al], v206\n\t"                                                                     \
    "v_add_f32_e32 v218, " FIN ", v216\n\t"                                                                   \
    "ds_write_b64 %[za], v[214:215] offset:" QPSK_STR(ZOFF) "\n\t"                                            \
    "v_add_f32_e32 v219, " PIN ", v218\n\t"                                                                   \
    "ds_write_b8 %[qa], v202 offset:" QPSK_STR(QOFF) "\n\t"                                                   \
    "v_add_f32_e32 " POUT ", v219, v217\n\t"                                                                  \
    "v_med3_f32 " FOUT ", v218, %[fmin], %[fmax]\n\t"                                                         \
    "s_or_b64 %[fl], %[fl], %[tm]\n\t"                                                                        \
    "v_cmp_ge_f32_e64 vcc, |" POUT "|, %[tau]\n\t"                                                            \
    "s_cbranch_vccz 1f\n\t"                                                                                   \
    "v_cvt_f64_f32 v[200:201], " POUT "\n\t"                                                                  \
    "v_bfi_b32 v229, %[absm], v227, " POUT "\n\t"                                                             \
    "v_add_f64 v[200:201], v[200:201], -v[228:229]\n\t"                                                       \
    "v_cvt_f32_f64 v204, v[200:201]\n\t"                                                                      \
    "v_cndmask_b32_e32 " POUT ", " POUT ", v204, vcc\n\t"                                                     \
    "v_cmp_ge_f32_e64 vcc, |" POUT "|, %[tau]\n\t"                                                            \
    "s_or_b64 %[fl], %[fl], vcc\n"                                                                            \
    "1:\n\t"

/*
 * Runs up to `groups` groups of 8 steps starting at LDS addresses d_addr / z_addr / q_addr (advanced on
 * return).  Returns the number of groups NOT done: 0, or -- if the flag word is nonzero -- the abandoned
 * group and everything after it, with phase/freq restored to that group's start.
 */
__device__ __forceinline__ unsigned costas_asm_run(float &phase, float &freq, unsigned &d_addr, unsigned &z_addr,
                                                   unsigned &q_addr, unsigned groups, float alpha, float beta,
                                                   float min_freq, float max_freq, unsigned long long &flags_out)
{
    unsigned long long flags, tmp;
    const double magic = 0x1.8p52, c3 = -0x1.6c087e89a359dp-10, s2 = 0x1.1107605230bc4p-7;
    asm volatile(
        "v_mov_b32 v228, 0x54442d18\n\t"        /* 2*pi = 0x401921FB54442D18 */
        "v_mov_b32 v227, 0x401921fb\n\t"
        "ds_read_b64 v[220:221], %[da]\n\t"
        "s_mov_b64 %[fl], 0\n\t"
        "s_waitcnt lgkmcnt(0)\n"
        "2:\n\t"
        "v_mov_b32 v230, %[p]\n\t"
        "v_mov_b32 v231, %[f]\n\t"
        QPSK_COSTAS_STEP("%[p]", "%[f]", "v224", "v225", 8, 0, 0)
        QPSK_COSTAS_STEP("v224", "v225", "v222", "v223", 16, 8, 1)
        QPSK_COSTAS_STEP("v222", "v223", "v224", "v225", 24, 16, 2)
        QPSK_COSTAS_STEP("v224", "v225", "v222", "v223", 32, 24, 3)
        QPSK_COSTAS_STEP("v222", "v223", "v224", "v225", 40, 32, 4)
        QPSK_COSTAS_STEP("v224", "v225", "v222", "v223", 48, 40, 5)
        QPSK_COSTAS_STEP("v222", "v223", "v224", "v225", 56, 48, 6)
        QPSK_COSTAS_STEP("v224", "v225", "%[p]", "%[f]", 64, 56, 7)
        "s_cmp_lg_u64 %[fl], 0\n\t"
        "s_cbranch_scc1 3f\n\t"
        "v_add_u32_e32 %[da], 64, %[da]\n\t"
        "v_add_u32_e32 %[za], 64, %[za]\n\t"
        "v_add_u32_e32 %[qa], 8, %[qa]\n\t"
        "s_sub_u32 %[ng], %[ng], 1\n\t"
        "s_cmp_lg_u32 %[ng], 0\n\t"
        "s_cbranch_scc1 2b\n\t"
        "s_branch 4f\n"
        "3:\n\t"
        "v_mov_b32 %[p], v230\n\t"
        "v_mov_b32 %[f], v231\n"
        "4:\n\t"
        "s_waitcnt lgkmcnt(0)"
        : [p] "+v"(phase), [f] "+v"(freq), [da] "+v"(d_addr), [za] "+v"(z_addr), [qa] "+v"(q_addr),
          [ng] "+s"(groups), [fl] "=&s"(flags), [tm] "=&s"(tmp)
        : [magic] "v"(magic), [c3] "v"(c3), [s2] "v"(s2), [fmax] "v"(max_freq), [be] "v"(beta), [al] "v"(alpha),
          [k2pi] "s"(0x1.45F306DC9C883p-1), [hpi] "s"(0x1.921FB54442D18p0), [c4] "s"(0x1.99343027bf8c3p-16),
          [s3] "s"(-0x1.994eb3774cf24p-13), [c2] "s"(0x1.55553e1068f19p-5), [s1] "s"(-0x1.555545995a603p-3),
          [c1] "s"(-0x1.ffffffd0c621cp-2), [fmin] "s"(min_freq), [tau] "s"(TAU_F), [absm] "s"(0x7fffffffu)
        : "vcc", "scc", "memory", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209",
          "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222",
          "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231");
    flags_out = flags;
    return groups;
}

} // namespace qpsk
#endif
