/*
 * costas_asm.h -- the Costas recurrence (qpsk.c:197-207 + costas_loop.c:44-74) as a hand-scheduled gfx950
 * instruction stream, for the serial wave of rx_fused_pipe_kernel.
 *
 * Why assembly: the recurrence runs in ONE wave per workgroup and that wave is strictly in order.  Measured
 * on MI355X (tools/ubench.hip, tools/gen_ubench_costas.py): a lone wave issues one VALU instruction per ~5
 * cycles and a dependent one ~8 cycles after its producer; every LDS instruction costs the wave ~12 cycles
 * of issue; a branch that waits for a VALU compare, or a TAKEN branch, stalls it for tens of cycles.  A step
 * is a chain of ~19 dependent operations (phase -> range reduction -> cosine Horner chain -> rotate ->
 * detector -> loop update -> phase); the compiler's version of it ran ~355 cycles per step, the arithmetic
 * chain alone is ~140.  What this stream does about it:
 *   - off-chain work is slotted between the dependent operations;
 *   - LDS traffic is 1.5 instructions per step: two symbols per ds_read_b128 (fetched two steps ahead into
 *     alternating register sets), one 16-byte record (T.x, T.y, n, -) per ds_write_b128;
 *   - the 2*pi wrap is out of line: the common case falls through one not-taken branch whose compare was
 *     issued several instructions earlier; the wrap block fixes the phase and jumps back;
 *   - the exact-zero test of the detector input is a running min over a group of 8 steps; with zeros out of
 *     the way sgn(T.x) T.y is |T.y| carrying the sign of T.x ^ T.y (one xor, two bit-field inserts, no compares).
 *
 * Arithmetic = costas_step_t() in qpsk_device.h operation for operation: Horner sin/cos polynomials in fp64
 * with fused multiply-adds (the library's form), everything in fp32 unfused (the reference is built without
 * contraction), the 2*pi wrap in fp64 (costas_loop.c:61-67), the clamp as a median of 3 (callers use this
 * stream only when min_freq < 0 < max_freq, where it equals costas_loop.c:69-74).  The parity tests compare
 * the kernel with the oracle bit for bit.
 *
 * One call runs `groups` groups of 8 steps; the first symbol must be an EVEN one (16-byte aligned pairs).
 * Cases the stream does not handle set a flag, and the group they occur in is abandoned with the loop state
 * restored to the group's start; the caller redoes that group with costas_step_t() and continues:
 *     min(|T.x|, |T.y|) == 0   (the detector's sgn(0) = -1 asymmetry, see costas_step_t)
 *     a phase still outside [-2pi, 2pi] after ONE wrap (clamp wider than +-2pi, huge amplitudes).
 *
 * Registers: v[100:139] (136:139 = the second pair of decimated symbols) are scratch owned by the block (clobbered; low enough for a kernel built for three
 * waves per SIMD, i.e. at most 168 VGPRs):
 *   100:101 x / d*C      102:103 beta*e, alpha*e   104:105 n / x3 / d*S     106:107 xr / e
 *   108:109 x2 / a, b    110:111 cos chain (v110 = C)   112:113 sin chain (v112 = S)
 *   114:117 the record: T.x, T.y, then the magic sum (v116 bits 1:0 = quadrant, v117 don't care)
 *   118 f2   119 p+f2    120:123 two decimated symbols   126 running min   127 2pi hi   128:129 +-2pi
 *   130,131 group-start phase/freq    132..135 phase/freq ping-pong
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

/*
 * one step: PIN/FIN -> POUT/FOUT.  DREG = the VGPR pair holding this step's symbol, WAIT = the lgkmcnt wait
 * in front of its first use, READ = the LDS fetch of the next pair (odd steps) or nothing, ZOFF = byte offset
 * of this step's record, LW/LR = the out-of-line wrap block's label and its return label.
 */
#define QPSK_COSTAS_STEP(PIN, FIN, POUT, FOUT, DREG, WAIT, READ, ZOFF, LW, LR)                                \
    "v_cvt_f64_f32 v[100:101], " PIN "\n\t"                                                                   \
    "v_fma_f64 v[116:117], v[100:101], %[k2pi], %[magic]\n\t"                                                 \
    "v_add_f64 v[104:105], v[116:117], -%[magic]\n\t"                                                         \
    "v_fma_f64 v[106:107], -v[104:105], %[hpi], v[100:101]\n\t"                                               \
    "v_mul_f64 v[108:109], v[106:107], v[106:107]\n\t"                                                        \
    "v_fma_f64 v[110:111], v[108:109], %[c4], %[c3]\n\t"                                                      \
    "v_fma_f64 v[112:113], v[108:109], %[s3], %[s2]\n\t"                                                      \
    "v_fma_f64 v[110:111], v[108:109], v[110:111], %[c2]\n\t"                                                 \
    "v_mul_f64 v[104:105], v[106:107], v[108:109]\n\t"                                                        \
    "v_fma_f64 v[110:111], v[108:109], v[110:111], %[c1]\n\t"                                                 \
    "v_fma_f64 v[112:113], v[108:109], v[112:113], %[s1]\n\t"                                                 \
    "v_fma_f64 v[110:111], v[108:109], v[110:111], 1.0\n\t"                                                   \
    "v_fma_f64 v[112:113], v[104:105], v[112:113], v[106:107]\n\t"                                            \
    "v_cvt_f32_f64 v110, v[110:111]\n\t"                                                                      \
    "v_cvt_f32_f64 v112, v[112:113]\n\t"                                                                      \
    WAIT                                                                                                      \
    "v_pk_mul_f32 v[100:101], " DREG ", v[110:111] op_sel_hi:[1,0]\n\t"                                       \
    "v_pk_mul_f32 v[104:105], " DREG ", v[112:113] op_sel:[1,0] op_sel_hi:[0,0]\n\t"                          \
    READ                                                                                                      \
    "v_pk_add_f32 v[114:115], v[100:101], v[104:105] neg_hi:[0,1]\n\t"                                        \
    "v_xor_b32_e32 v108, v114, v115\n\t"                      /* sign bit = sgn(T.x) sgn(T.y) (no zeros: v126) */ \
    "v_min3_f32 v126, v126, |v114|, |v115|\n\t"                                                               \
    "v_bfi_b32 v109, %[absm], v115, v108\n\t"                 /* sgn(T.x) T.y = |T.y| with that sign */          \
    "v_bfi_b32 v108, %[absm], v114, v108\n\t"                 /* sgn(T.y) T.x */                                  \
    "v_sub_f32_e32 v106, v109, v108\n\t"                                                                      \
    "v_pk_mul_f32 v[102:103], %[beal], v[106:107] op_sel_hi:[1,0]\n\t"                                        \
    "v_add_f32_e32 v118, " FIN ", v102\n\t"                                                                   \
    "v_add_f32_e32 v119, " PIN ", v118\n\t"                                                                   \
    "v_add_f32_e32 " POUT ", v119, v103\n\t"                                                                  \
    "v_cmp_ge_f32_e64 vcc, |" POUT "|, %[tau]\n\t"                                                            \
    "ds_write_b128 %[za], v[114:117] offset:" QPSK_STR(ZOFF) "\n\t"                                           \
    "v_med3_f32 " FOUT ", v118, %[fmin], %[fmax]\n\t"                                                         \
    "s_cbranch_vccnz " LW "f\n"                                                                               \
    LR ":\n\t"

/* the out-of-line wrap of costas_loop.c:61-67 for one step: phase -= copysign(2pi, phase) in fp64, once */
#define QPSK_COSTAS_WRAP(POUT, LW, LR)                                                                        \
    LW ":\n\t"                                                                                                \
    "v_cvt_f64_f32 v[100:101], " POUT "\n\t"                                                                  \
    "v_bfi_b32 v129, %[absm], v127, " POUT "\n\t"                                                             \
    "v_add_f64 v[100:101], v[100:101], -v[128:129]\n\t"                                                       \
    "v_cvt_f32_f64 v104, v[100:101]\n\t"                                                                      \
    "v_cndmask_b32_e32 " POUT ", " POUT ", v104, vcc\n\t"                                                     \
    "v_cmp_ge_f32_e64 vcc, |" POUT "|, %[tau]\n\t"                                                            \
    "s_or_b64 %[fl], %[fl], vcc\n\t"                                                                          \
    "s_branch " LR "b\n"

/* symbol pairs alternate between two register sets, each fetched TWO steps before its first use (the FIR waves
 * keep the LDS queue busy; one step of slack was not always enough): when a set is first used, the two record
 * writes issued since its read may still be in flight */
#define QPSK_WAIT2 "s_waitcnt lgkmcnt(2)\n\t"
#define QPSK_RDA(OFF) "ds_read_b128 v[120:123], %[da] offset:" QPSK_STR(OFF) "\n\t"
#define QPSK_RDB(OFF) "ds_read_b128 v[136:139], %[da] offset:" QPSK_STR(OFF) "\n\t"

/*
 * Runs up to `groups` groups of 8 steps starting at LDS addresses d_addr (symbols, 8 bytes each, 16-byte
 * aligned) and z_addr (records, 16 bytes each); both are advanced on return.  Returns the number of groups
 * NOT done: 0, or -- if the flag word is nonzero -- the abandoned group and everything after it, with
 * phase/freq restored to that group's start.
 */
__device__ __forceinline__ unsigned costas_asm_run(float &phase, float &freq, unsigned &d_addr, unsigned &z_addr,
                                                   unsigned groups, float alpha, float beta, float min_freq,
                                                   float max_freq, unsigned long long &flags_out)
{
    unsigned long long flags, tmp;
    const double magic = 0x1.8p52, c3 = -0x1.6c087e89a359dp-10, s2 = 0x1.1107605230bc4p-7;
    double beal;                 /* (beta, alpha) as one VGPR pair for the packed multiply by e */
    {
        const float2 ba = make_float2(beta, alpha);
        __builtin_memcpy(&beal, &ba, 8);
    }
    asm volatile(
        "v_mov_b32 v128, 0x54442d18\n\t"        /* 2*pi = 0x401921FB54442D18 */
        "v_mov_b32 v127, 0x401921fb\n\t"
        "ds_read_b128 v[120:123], %[da]\n\t"
        "s_mov_b64 %[fl], 0\n\t"
        "s_waitcnt lgkmcnt(0)\n"
        "2:\n\t"
        "v_mov_b32 v130, %[p]\n\t"
        "v_mov_b32 v131, %[f]\n\t"
        "v_mov_b32 v126, 0x7f800000\n\t"        /* running min of |T.x|, |T.y| over the group: 0 <=> some exact zero */
        QPSK_COSTAS_STEP("%[p]", "%[f]", "v132", "v133", "v[120:121]", QPSK_WAIT2, QPSK_RDB(16), 0, "10", "20")
        QPSK_COSTAS_STEP("v132", "v133", "v134", "v135", "v[122:123]", "", "", 16, "11", "21")
        QPSK_COSTAS_STEP("v134", "v135", "v132", "v133", "v[136:137]", QPSK_WAIT2, QPSK_RDA(32), 32, "12", "22")
        QPSK_COSTAS_STEP("v132", "v133", "v134", "v135", "v[138:139]", "", "", 48, "13", "23")
        QPSK_COSTAS_STEP("v134", "v135", "v132", "v133", "v[120:121]", QPSK_WAIT2, QPSK_RDB(48), 64, "14", "24")
        QPSK_COSTAS_STEP("v132", "v133", "v134", "v135", "v[122:123]", "", "", 80, "15", "25")
        QPSK_COSTAS_STEP("v134", "v135", "v132", "v133", "v[136:137]", QPSK_WAIT2, QPSK_RDA(64), 96, "16", "26")
        QPSK_COSTAS_STEP("v132", "v133", "%[p]", "%[f]", "v[138:139]", "", "", 112, "17", "27")
        "v_cmp_eq_f32_e64 %[tm], 0, v126\n\t"
        "s_or_b64 %[fl], %[fl], %[tm]\n\t"
        "s_cmp_lg_u64 %[fl], 0\n\t"
        "s_cbranch_scc1 3f\n\t"
        "v_add_u32_e32 %[da], 64, %[da]\n\t"
        "v_add_u32_e32 %[za], 0x80, %[za]\n\t"
        "s_sub_u32 %[ng], %[ng], 1\n\t"
        "s_cmp_lg_u32 %[ng], 0\n\t"
        "s_cbranch_scc1 2b\n\t"
        "s_branch 4f\n"
        "3:\n\t"
        "v_mov_b32 %[p], v130\n\t"
        "v_mov_b32 %[f], v131\n\t"
        "s_branch 4f\n"
        QPSK_COSTAS_WRAP("v132", "10", "20")
        QPSK_COSTAS_WRAP("v134", "11", "21")
        QPSK_COSTAS_WRAP("v132", "12", "22")
        QPSK_COSTAS_WRAP("v134", "13", "23")
        QPSK_COSTAS_WRAP("v132", "14", "24")
        QPSK_COSTAS_WRAP("v134", "15", "25")
        QPSK_COSTAS_WRAP("v132", "16", "26")
        QPSK_COSTAS_WRAP("%[p]", "17", "27")
        "4:\n\t"
        "s_waitcnt lgkmcnt(0)"
        : [p] "+v"(phase), [f] "+v"(freq), [da] "+v"(d_addr), [za] "+v"(z_addr), [ng] "+s"(groups),
          [fl] "=&s"(flags), [tm] "=&s"(tmp)
        : [magic] "v"(magic), [c3] "v"(c3), [s2] "v"(s2), [fmax] "v"(max_freq), [beal] "v"(beal),
          [k2pi] "s"(0x1.45F306DC9C883p-1), [hpi] "s"(0x1.921FB54442D18p0), [c4] "s"(0x1.99343027bf8c3p-16),
          [s3] "s"(-0x1.994eb3774cf24p-13), [c2] "s"(0x1.55553e1068f19p-5), [s1] "s"(-0x1.555545995a603p-3),
          [c1] "s"(-0x1.ffffffd0c621cp-2), [fmin] "s"(min_freq), [tau] "s"(TAU_F), [absm] "s"(0x7fffffffu)
        : "vcc", "scc", "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109",
          "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122",
          "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135",
          "v136", "v137", "v138", "v139");
    flags_out = flags;
    return groups;
}

} // namespace qpsk
#endif
