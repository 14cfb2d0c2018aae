/*
 * costas_asm.h -- the Costas recurrence (qpsk.c:197-207 + costas_loop.c:44-74) as a hand-scheduled gfx950
 * instruction stream, for the serial wave of the pipeline kernels (rx_lean_kernel, rx_fused_pipe_kernel, rx_pipe2_kernel,
 * costas_pipe_kernel, stream_block_kernel; rx_hist_kernel runs the generated low-register copy, costas_asm_lo.h).
 *
 * Why assembly: the recurrence runs in ONE wave per workgroup and that wave is strictly in order.  Measured
 * on MI355X (timing-only variants of this stream inside the kernel in round 1, tools/ab_libs.py; the stream alone
 * on a CU with pieces removed in round 2, tools/ubench_step.py, profiles/r02_step_cost.txt): the step is ISSUE
 * bound.  The wave pays ~4.5-5.8 cycles per VALU instruction whether or not it depends on the one before (the "8
 * cycles per dependent instruction" of round 1's notes came from one-instruction asm statements, which the compiler
 * pads with s_nop), ~16-21 per LDS instruction almost regardless of its size (16 -> 12 -> 4 bytes per step: -2.4 %,
 * then -1 %; one write per 16 steps instead of one per step: -8 %), 13-15 for the per-step wrap test's branch with
 * the wraps it takes.  (Round 6, profiles/r06_step_cost.txt: the price is per INSTRUCTION, ~4.5 cycles, not per byte -- 4-byte
 * encodings of three fp64 operations changed nothing -- and a branch waits for the compare that feeds it: 165.6 cycles per step with the
 * wrap branch right behind its test, 146.1 nine instructions on.)  The compiler's version of the step ran ~355 cycles.  What this stream
 * does about it:
 *   - 28 VALU instructions per step (26 in the paired-lane form below), as few of them 8-byte encodings as the ISA allows: the
 *     range reduction is v_mul_f64 + v_rndne_f64 (VOP1) + v_fmac_f64 (VOP2: gfx90a on) in place on the argument, the sine's last
 *     stage a v_fmac_f64 in place on the reduced argument (round 6; n = rndne(fl(x 2/pi)) equals the magic-number rounding
 *     of round 1-5 for every float in [-8, 8]: tools/check_device_sincos.cpp --stream, tests/test_sincos.py);
 *   - the only thing a step leaves behind is the PHASE it started from (the FIR waves' flush redoes sin/cos and
 *     the rotation from it, bit for bit the same operations): four steps' phases sit in v140..v143 and go to LDS
 *     in one ds_write_b128, so the wave issues 0.75 LDS instructions per step (two symbols per ds_read_b128,
 *     fetched two steps ahead into alternating register sets) instead of 1.5;
 *   - what nothing in step k+1 waits for -- step k's frequency clamp and exact-zero test -- is issued between the dependent fp64
 *     operations of step k+1's head (one-lane form) or where a DPP read needs two wait states (paired form);
 *   - the 2*pi wrap is out of line AND late: step k+1 starts from the unwrapped phase; its 2*pi test is the step's FIRST instruction and
 *     the branch stands behind the head and the first rows of the polynomial chains (round 6), and the rare wrap block corrects the phase
 *     where it stands and redoes what stood in front of the branch;
 *   - groups of 16 steps: the per-group bookkeeping (state snapshot, flag test, taken loop branch) costs
 *     ~60 cycles;
 *   - the exact-zero test of the detector input is a running min over the group; with zeros out of the way
 *     the error is e = s (|T.y| - |T.x|) with s = sgn(T.x) sgn(T.y) (negating both operands of a float
 *     subtraction negates its result), so d = |T.y| - |T.x| is one subtract with abs modifiers and s rejoins it
 *     as the +-1.0f factor of the two fused multiply-adds that update freq and phase (+-1 x float is exact, so
 *     each is the reference's unfused multiply, then add).  Only the sign of a ZERO error differs (s d = -0
 *     where the reference has +0), and that is visible only when freq is -0; freq can be -0 only as loaded
 *     state (costas_loop.c:56: x + y = -0 needs both -0, and beta e = -0 needs T = 0, which is flagged), so
 *     the caller keeps a group that starts with freq = -0 away from this stream.
 *
 * Arithmetic = costas_step_t() in qpsk_device.h: Horner sin/cos polynomials in fp64 with fused multiply-adds
 * (the library's form), everything else in fp32 unfused (the reference is built without contraction), the
 * 2*pi wrap in fp64 (costas_loop.c:61-67), the clamp as a median of 3 (callers use this stream only when
 * min_freq < 0 < max_freq, where it equals costas_loop.c:69-74).  The parity tests compare the kernel with
 * the oracle bit for bit.
 *
 * One call runs `groups` groups of COSTAS_ASM_GROUP steps; the first symbol's number must be a multiple of 4
 * (16-byte aligned symbol pairs and record quads).  Cases the stream does not handle set a flag, and the group
 * they occur in is abandoned with the loop state restored to the group's start (its records are then rewritten
 * too); the caller redoes that group with costas_step_t() and continues:
 *     min(|T.x|, |T.y|) == 0   (the detector's sgn(0) = -1 asymmetry, see costas_step_t)
 *     a phase still outside [-2pi, 2pi] after ONE wrap (clamp wider than +-2pi, huge amplitudes).
 * `ign` (a lane mask) takes lanes out of the first test.  It is for symbols that are exactly (+0, +0) -- a stream's first block,
 * a squelched input, an all-zero frame -- where T is (+-0, +-0) whatever the phase: d = |T.y| - |T.x| = +0, the two updates add
 * +-0, and x + (+-0) = x for every x but -0, so the stream's step IS the reference's (phase + freq, wrap, clamp; state at rest
 * stays at rest) as long as neither phase nor freq is -0 -- which only loaded state can be, and the sum of two floats is -0 only if
 * both are.  The caller checks symbols and state (rx_fused.hip, zero_run) before it sets a lane's bit, per stretch of symbols.
 *
 * Registers: v[100:143] are scratch owned by the block (clobbered; low enough for a kernel built for three
 * waves per SIMD, i.e. at most 168 VGPRs):
 *   100:101 d*C (paired form: M, then C | S)      102:103 beta*d, alpha*d   104:105 x 2/pi -> n / x3 / d*S     106:107 x -> xr -> S / d
 *   108:109 x2 / s       110:111 cos chain (v110 = C)   112:113 sin chain (v112 = S)
 *   114:115 T            116:117 free
 *   118 freq (clamped in place by the next step's head)    120:123, 136:139 two pairs of decimated symbols   126 running min
 *   127 2pi hi   128:129 +-2pi   130,131 group-start phase/freq
 *   140:143 the phases of four consecutive steps (step k in v140 + k % 4) = their records
 */
#ifndef QPSK_COSTAS_ASM_H
#define QPSK_COSTAS_ASM_H

#include "qpsk_device.h"

namespace qpsk {

/* steps per group of the stream; the caller hands over multiples of it */
constexpr int COSTAS_ASM_GROUP = 16;

/* 32-bit LDS byte address of a __shared__ object (what ds_read / ds_write take) */
__device__ __forceinline__ unsigned lds_addr(const void *p)
{
    return (unsigned)(__UINTPTR_TYPE__)(const __attribute__((address_space(3))) void *)p;
}

/* a wave-uniform 64-bit value the compiler cannot see as one (a ballot carried round a divergent loop): into SGPRs */
__device__ __forceinline__ unsigned long long uniform64(unsigned long long v)
{
    return ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(v >> 32)) << 32) |
           (unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)v);
}

#define QPSK_STR_(x) #x
#define QPSK_STR(x) QPSK_STR_(x)

/* head of a step: phase PIN -> x = (double)PIN in v[106:107], n = rndne(x 2/pi) (v[104:105]), xr = x - n pi/2 IN PLACE
 * (v_fmac_f64: a 4-byte VOP2 encoding, like v_rndne_f64's VOP1), xr^2 (v108:109), the first row of each polynomial chain */
#define QPSK_HEAD_CHAIN(PIN)                                                                                  \
    "v_cvt_f64_f32 v[106:107], " PIN "\n\t"                                                                   \
    "v_mul_f64 v[104:105], v[106:107], %[k2pi]\n\t"                                                           \
    "v_rndne_f64_e32 v[104:105], v[104:105]\n\t"                                                              \
    "v_fmac_f64_e32 v[106:107], %[nhpi], v[104:105]\n\t"                                                      \
    "v_mul_f64 v[108:109], v[106:107], v[106:107]\n\t"                                                        \
    "v_fma_f64 v[110:111], v[108:109], %[c4], %[c3]\n\t"                                                      \
    "v_fma_f64 v[112:113], v[108:109], %[s3], %[s2]\n\t"

/*
 * A step of the one-lane stream in front of its wrap branch: the 2*pi test of its phase PIN FIRST (the branch waits for the
 * compare's VCC: see QPSK_STEP_P_PRE), the head with the previous step's leftovers in between -- its frequency clamp (v118, in
 * place: this step's input) and its exact-zero test -- and the first rows of the two polynomial chains.  Everything here is a pure
 * function of PIN or idempotent: LW (QPSK_WRAP_HEAD) wraps PIN in place, runs it again, returns to LR.
 */
#define QPSK_CMP_P(PIN) "v_cmp_ge_f32_e64 vcc, |" PIN "|, %[tau]\n\t"
#define QPSK_STEP_1_PRE(CMP, PIN)                                                                             \
    CMP                                                                                                       \
    "v_cvt_f64_f32 v[106:107], " PIN "\n\t"                                                                   \
    "v_mul_f64 v[104:105], v[106:107], %[k2pi]\n\t"                                                           \
    "v_med3_f32 v118, v118, %[fmin], %[fmax]\n\t"                                                             \
    "v_rndne_f64_e32 v[104:105], v[104:105]\n\t"                                                              \
    "v_min3_f32 v126, v126, |v114|, |v115|\n\t"                                                               \
    "v_fmac_f64_e32 v[106:107], %[nhpi], v[104:105]\n\t"                                                      \
    "v_mul_f64 v[108:109], v[106:107], v[106:107]\n\t"                                                        \
    "v_fma_f64 v[110:111], v[108:109], %[c4], %[c3]\n\t"                                                      \
    "v_fma_f64 v[112:113], v[108:109], %[s3], %[s2]\n\t"
#define QPSK_HEAD_DEFERRED(PIN, LW, LR)                                                                      \
    QPSK_STEP_1_PRE(QPSK_CMP_P(PIN), PIN)                                                                     \
    "s_cbranch_vccnz " LW "f\n"                                                                               \
    LR ":\n\t"

/*
 * The paired-lane stream's step in front of its wrap branch (PRE) and behind it (QPSK_BODY_P).  The branch waits for the compare's VCC:
 * measured with the stream alone on a CU (tools/ubench_step.py, profiles/r06_step_cost.txt), a step costs 165.6 cycles with the branch
 * right behind the compare, 160.6 two instructions on, 148.3 five on, 146.1 nine on and 150-154 later still (where the branch falls in
 * the fetch lines moves it by +-3), so the 2*pi test is the step's FIRST instruction and the branch stands nine instructions behind it:
 * everything in between is a pure function of PIN (the out-of-line wrap corrects PIN and runs PRE again).
 */
#define QPSK_STEP_P_PRE(CMP, PIN)                                                                             \
    CMP                                                                                                       \
    "v_cvt_f64_f32 v[106:107], " PIN "\n\t"                                                                   \
    "v_mul_f64 v[104:105], v[106:107], %[k2pi]\n\t"                                                           \
    "v_rndne_f64_e32 v[104:105], v[104:105]\n\t"                                                              \
    "v_fmac_f64_e32 v[106:107], %[nhpi], v[104:105]\n\t"                                                      \
    "v_mul_f64 v[108:109], v[106:107], v[106:107]\n\t"                                                        \
    "v_fma_f64 v[100:101], v[106:107], %[km], %[kj]\n\t"                                                      \
    "v_fma_f64 v[110:111], v[108:109], %[ka], %[kb]\n\t"                                                      \
    "v_mul_f64 v[104:105], v[108:109], v[100:101]\n\t"                                                        \
    "v_fma_f64 v[110:111], v[108:109], v[110:111], %[kc]\n\t"
#define QPSK_HEAD_CHAIN_P(PIN) QPSK_STEP_P_PRE("", PIN)
#define QPSK_HEAD_DEFERRED_P(PIN, LW, LR)                                                                    \
    QPSK_STEP_P_PRE(QPSK_CMP_P(PIN), PIN)                                                                     \
    "s_cbranch_vccnz " LW "f\n"                                                                               \
    LR ":\n\t"

/*
 * the rest of a step: sin/cos polynomials -> T = symbol x conj(C + jS) -> detector -> loop update.  Leaves
 * T in v114:115 (with v116 from the head: the record), the unclamped frequency in v118, the new phase,
 * unwrapped, in POUT.  The two updates are v_fmac_f32 in its 4-byte encoding (s = +-1: the product is exact, so each is the
 * reference's unfused multiply, then add): a lone wave's issue rate is set by instruction BYTES, ~1.56 per cycle
 * (profiles/r03_ubench_fetch.txt: 4.15 cycles per 4-byte instruction, 5.14 per 8-byte one, 5.75 for v_fma_f32).  DREG = the VGPR pair holding this step's symbol, WAIT = the lgkmcnt wait in front of
 * its first use, READ = the LDS fetch of the pair after next (even steps) or nothing.
 */
/* the fp32 part of a step, from C in v110 and S in v112 */
#define QPSK_BODY_F32(PIN, POUT, DREG, WAIT, READ, QW)                                                        \
    WAIT                                                                                                      \
    "v_pk_mul_f32 v[100:101], " DREG ", v[110:111] op_sel_hi:[1,0]\n\t"                                       \
    "v_pk_mul_f32 v[104:105], " DREG ", v[112:113] op_sel:[1,0] op_sel_hi:[0,0]\n\t"                          \
    READ                                                                                                      \
    "v_pk_add_f32 v[114:115], v[100:101], v[104:105] neg_hi:[0,1]\n\t"                                        \
    QW                                                                                                        \
    "v_sub_f32_e64 v106, |v115|, |v114|\n\t"                  /* d = |T.y| - |T.x|;  e = s d */                  \
    "v_xor_b32_e32 v108, v114, v115\n\t"                      /* sign bit of s = sgn(T.x) sgn(T.y) */            \
    "v_pk_mul_f32 v[102:103], %[beal], v[106:107] op_sel_hi:[1,0]\n\t"   /* (beta d, alpha d) */                \
    "v_bfi_b32 v108, %[absm], 1.0, v108\n\t"                  /* s as +-1.0f */                                  \
    "v_fmac_f32_e32 v118, v108, v102\n\t"                     /* freq + beta e (v118 = the clamped freq) */      \
    "v_add_f32_e32 " POUT ", " PIN ", v118\n\t"                                                               \
    "v_fmac_f32_e32 " POUT ", v108, v103\n\t"                 /* (phase + freq) + alpha e */

#define QPSK_BODY(PIN, POUT, DREG, WAIT, READ, QW)                                                            \
    "v_fma_f64 v[110:111], v[108:109], v[110:111], %[c2]\n\t"                                                 \
    "v_mul_f64 v[104:105], v[106:107], v[108:109]\n\t"                                                        \
    "v_fma_f64 v[110:111], v[108:109], v[110:111], %[c1]\n\t"                                                 \
    "v_fma_f64 v[112:113], v[108:109], v[112:113], %[s1]\n\t"                                                 \
    "v_fma_f64 v[110:111], v[108:109], v[110:111], 1.0\n\t"                                                   \
    "v_fmac_f64_e32 v[106:107], v[104:105], v[112:113]\n\t"   /* S = xr + x3 q, in place on xr */                \
    "v_cvt_f32_f64 v110, v[110:111]\n\t"                                                                      \
    "v_cvt_f32_f64 v112, v[106:107]\n\t"                                                                      \
    QPSK_BODY_F32(PIN, POUT, DREG, WAIT, READ, QW)

/*
 * The paired-lane body (round 6): lanes 2f and 2f + 1 run the SAME loop and hold the same state; the two Horner chains of the
 * step -- 4 + 4 fp64 operations -- are ONE chain of per-lane coefficients: the even lane evaluates the cosine, the odd lane the sine,
 *     M  = fma(xr, km, kj)     = 1                | xr          (km, kj) = (0, 1) | (1, -0): exact
 *     r  = fma(x2, ka, kb)     = fma(x2, c4, c3)  | s3          ka = c4 | 0: x2 0 + s3 = s3 exactly
 *     A  = x2 M                = x2               | x3 = xr x2  (the product the one-lane form rounds, commuted)
 *     r  = fma(x2, r, kc), fma(x2, r, kd)          kc, kd = c2, c1 | s2, s1
 *     M += A r  (v_fmac_f64)   = fma(x2, r, 1)    | fma(x3, r, xr)
 * i.e. per value the operations of QPSK_BODY, so the exhaustive device check of the one-lane form covers it.  One DPP
 * multiply then hands each lane its partner's value: the even lane holds (C, S); the odd lane (S, -C) (sg = 1 | -1), the pair
 * of the phase turned by three quarter turns, so its T is the even lane's turned by one -- (-T.y, T.x), bit for bit: negation
 * and swapping commute with every rounding involved -- and the detector takes the same value on it (qpsk_device.h, costas_step_t;
 * the exact-zero test covers the one case where it does not, in both lanes alike).  From there both lanes run the same fp32
 * instructions on their own T and arrive at the same frequency and phase: no way back is needed.  A VALU write needs two wait
 * states before a DPP read of it: the previous step's clamp and zero test stand there.  26 VALU instructions per step.  The first
 * four rows below stand in QPSK_STEP_P_PRE, in front of the step's wrap branch; QPSK_BODY_P is the rest.
 */
#define QPSK_BODY_P(PIN, POUT, DREG, WAIT, READ, QW)                                                          \
    "v_fma_f64 v[110:111], v[108:109], v[110:111], %[kd]\n\t"                                                 \
    "v_fmac_f64_e32 v[100:101], v[104:105], v[110:111]\n\t"                                                   \
    "v_cvt_f32_f64 v110, v[100:101]\n\t"                                                                      \
    "v_med3_f32 v118, v118, %[fmin], %[fmax]\n\t"                                                             \
    "v_min3_f32 v126, v126, |v114|, |v115|\n\t"                                                               \
    "v_mul_f32_dpp v112, v110, %[sg] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"                      \
    QPSK_BODY_F32(PIN, POUT, DREG, WAIT, READ, QW)

/* the leftovers of a group's LAST step, in line: clamp -> FOUT, zero test, 2*pi test of POUT */
#define QPSK_TAIL(POUT, FOUT, LW, LR)                                                                         \
    "v_cmp_ge_f32_e64 vcc, |" POUT "|, %[tau]\n\t"                                                            \
    "v_med3_f32 " FOUT ", v118, %[fmin], %[fmax]\n\t"                                                         \
    "v_min3_f32 v126, v126, |v114|, |v115|\n\t"                                                               \
    "s_cbranch_vccnz " LW "f\n"                                                                               \
    LR ":\n\t"

/* costas_loop.c:61-67 for the lanes in vcc: P -= copysign(2pi, P) in fp64, once; still outside -> flag */
#define QPSK_WRAP_ONCE(P)                                                                                     \
    "v_cvt_f64_f32 v[100:101], " P "\n\t"                                                                     \
    "v_bfi_b32 v129, %[absm], v127, " P "\n\t"                                                                \
    "v_add_f64 v[100:101], v[100:101], -v[128:129]\n\t"                                                       \
    "v_cvt_f32_f64 v104, v[100:101]\n\t"                                                                      \
    "v_cndmask_b32_e32 " P ", " P ", v104, vcc\n\t"                                                           \
    "v_cmp_ge_f32_e64 vcc, |" P "|, %[tau]\n\t"                                                               \
    "s_or_b64 %[fl], %[fl], vcc\n\t"

/* out-of-line wrap for QPSK_HEAD_DEFERRED: wrap PIN where it stands (the record written later is the WRAPPED phase, the
 * one the step uses), then the head again */
#define QPSK_WRAP_HEAD(PIN, LW, LR)                                                                           \
    LW ":\n\t"                                                                                                \
    QPSK_WRAP_ONCE(PIN)                                                                                       \
    QPSK_STEP_1_PRE("", PIN)                                                                                  \
    "s_branch " LR "b\n"
/* ... for QPSK_HEAD_DEFERRED_P: what stood in front of its branch again */
#define QPSK_WRAP_HEAD_P(PIN, LW, LR)                                                                         \
    LW ":\n\t"                                                                                                \
    QPSK_WRAP_ONCE(PIN)                                                                                       \
    QPSK_STEP_P_PRE("", PIN)                                                                                  \
    "s_branch " LR "b\n"

/* out-of-line wrap for QPSK_TAIL */
#define QPSK_WRAP_TAIL(POUT, LW, LR)                                                                          \
    LW ":\n\t"                                                                                                \
    QPSK_WRAP_ONCE(POUT)                                                                                      \
    "s_branch " LR "b\n"

/* symbol pairs alternate between two register sets, each fetched TWO steps before its first use (the FIR waves
 * keep the LDS queue busy; one step of slack was not always enough).  LDS operations complete in order, so the
 * wait in front of a set's first use allows as many outstanding ones as were issued after its read: the record
 * write of the step before (steps 0, 4, 8, 12) or none (steps 2, 6, 10, 14) */
#define QPSK_WAIT0 "s_waitcnt lgkmcnt(0)\n\t"
#define QPSK_WAIT1 "s_waitcnt lgkmcnt(1)\n\t"
/* QPSK_DA / QPSK_ZA: the registers holding the group's symbol / record address (defined in front of each stream) */
#define QPSK_RDA(OFF) "ds_read_b128 v[120:123], " QPSK_DA " offset:" QPSK_STR(OFF) "\n\t"
#define QPSK_RDB(OFF) "ds_read_b128 v[136:139], " QPSK_DA " offset:" QPSK_STR(OFF) "\n\t"
/* the records of four steps, their starting phases v140..v143, in one write (issued by the fourth of them once its
 * own phase has been through the 2*pi test, before its update overwrites v140) */
#define QPSK_QW(OFF) "ds_write_b128 " QPSK_ZA ", v[140:143] offset:" QPSK_STR(OFF) "\n\t"

/* four steps k = 4m .. 4m+3 (k > 0): the phase of step k lives in v140 + k % 4, the clamped frequency in v133
 * (odd k) or v135 (even k); SA/SB = the symbol sets of the first and the second pair, RD1/RD2 the fetches of the
 * two even steps, QOFF the byte offset of the four records, P4 where the fourth step leaves the next phase */
#define QPSK_STEP_QUAD_(HD, BD, SA_LO, SA_HI, SB_LO, SB_HI, RD1, RD2, QOFF, P4, L1, L2, L3, L4)                \
    HD("v140", "1" L1, "2" L1)                                                                        \
    BD("v140", "v141", SA_LO, QPSK_WAIT1, RD1, "")                                                    \
    HD("v141", "1" L2, "2" L2)                                                                        \
    BD("v141", "v142", SA_HI, "", "", "")                                                             \
    HD("v142", "1" L3, "2" L3)                                                                        \
    BD("v142", "v143", SB_LO, QPSK_WAIT0, RD2, "")                                                    \
    HD("v143", "1" L4, "2" L4)                                                                        \
    BD("v143", P4, SB_HI, "", "", QPSK_QW(QOFF))
#define QPSK_STEP_QUAD(...) QPSK_STEP_QUAD_(QPSK_HEAD_DEFERRED, QPSK_BODY, __VA_ARGS__)

/*
 * Runs up to `groups` groups of COSTAS_ASM_GROUP steps starting at LDS addresses d_addr (symbols, 8 bytes each,
 * 16-byte aligned) and z_addr (records, 4 bytes each, 16-byte aligned); both are advanced on return.  Returns the number of
 * groups NOT done: 0, or -- if the flag word is nonzero -- the abandoned group and everything after it, with
 * phase/freq restored to that group's start.  freq must not be -0.0f (see the header).
 */
#define QPSK_DA "%[da]"
#define QPSK_ZA "%[za]"
__device__ __forceinline__ unsigned costas_asm_run(float &phase, float &freq, unsigned &d_addr, unsigned &z_addr,
                                                   unsigned groups, float alpha, float beta, float min_freq,
                                                   float max_freq, unsigned long long &flags_out, unsigned long long ign = 0ull)
{
    unsigned long long flags, tmp;
    ign = uniform64(ign);
    const double c3 = -0x1.6c087e89a359dp-10, s2 = 0x1.1107605230bc4p-7;
    double beal;                 /* (beta, alpha) as one VGPR pair for the packed multiply by d */
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
        "v_mov_b32 v118, %[f]\n\t"            /* the frequency accumulates in place (v_fmac) */
        "v_mov_b32 v126, 0x7f800000\n\t"        /* running min of |T.x|, |T.y| over the group: 0 <=> some exact zero */
        "v_mov_b32 v140, %[p]\n\t"
        /* steps 0..3: the first has no predecessor in the group */
        QPSK_HEAD_CHAIN("v140")
        QPSK_BODY("v140", "v141", "v[120:121]", QPSK_WAIT1, QPSK_RDB(16), "")
        QPSK_HEAD_DEFERRED("v141", "101", "201")
        QPSK_BODY("v141", "v142", "v[122:123]", "", "", "")
        QPSK_HEAD_DEFERRED("v142", "102", "202")
        QPSK_BODY("v142", "v143", "v[136:137]", QPSK_WAIT0, QPSK_RDA(32), "")
        QPSK_HEAD_DEFERRED("v143", "103", "203")
        QPSK_BODY("v143", "v140", "v[138:139]", "", "", QPSK_QW(0))
        /* steps 4..15 */
        QPSK_STEP_QUAD("v[120:121]", "v[122:123]", "v[136:137]", "v[138:139]", QPSK_RDB(48), QPSK_RDA(64), 16, "v140", "04", "05", "06", "07")
        QPSK_STEP_QUAD("v[120:121]", "v[122:123]", "v[136:137]", "v[138:139]", QPSK_RDB(80), QPSK_RDA(96), 32, "v140", "08", "09", "10", "11")
        QPSK_STEP_QUAD("v[120:121]", "v[122:123]", "v[136:137]", "v[138:139]", QPSK_RDB(112), QPSK_RDA(128), 48, "%[p]", "12", "13", "14", "15")
        QPSK_TAIL("%[p]", "%[f]", "116", "216")
        "v_cmp_eq_f32_e64 %[tm], 0, v126\n\t"
        "s_andn2_b64 %[tm], %[tm], %[ign]\n\t"   /* lanes the caller has looked at: nothing but +0.0 symbols ahead of them (below) */
        "s_or_b64 %[fl], %[fl], %[tm]\n\t"
        "s_cmp_lg_u64 %[fl], 0\n\t"
        "s_cbranch_scc1 3f\n\t"
        "v_add_u32_e32 %[da], 0x80, %[da]\n\t"
        "v_add_u32_e32 %[za], 0x40, %[za]\n\t"
        "s_sub_u32 %[ng], %[ng], 1\n\t"
        "s_cmp_lg_u32 %[ng], 0\n\t"
        "s_cbranch_scc1 2b\n\t"
        "s_branch 4f\n"
        "3:\n\t"
        "v_mov_b32 %[p], v130\n\t"
        "v_mov_b32 %[f], v131\n\t"
        "s_branch 4f\n"
        QPSK_WRAP_HEAD("v141", "101", "201")
        QPSK_WRAP_HEAD("v142", "102", "202")
        QPSK_WRAP_HEAD("v143", "103", "203")
        QPSK_WRAP_HEAD("v140", "104", "204")
        QPSK_WRAP_HEAD("v141", "105", "205")
        QPSK_WRAP_HEAD("v142", "106", "206")
        QPSK_WRAP_HEAD("v143", "107", "207")
        QPSK_WRAP_HEAD("v140", "108", "208")
        QPSK_WRAP_HEAD("v141", "109", "209")
        QPSK_WRAP_HEAD("v142", "110", "210")
        QPSK_WRAP_HEAD("v143", "111", "211")
        QPSK_WRAP_HEAD("v140", "112", "212")
        QPSK_WRAP_HEAD("v141", "113", "213")
        QPSK_WRAP_HEAD("v142", "114", "214")
        QPSK_WRAP_HEAD("v143", "115", "215")
        QPSK_WRAP_TAIL("%[p]", "116", "216")
        "4:\n\t"
        "s_waitcnt lgkmcnt(0)"
        : [p] "+v"(phase), [f] "+v"(freq), [da] "+v"(d_addr), [za] "+v"(z_addr), [ng] "+s"(groups),
          [fl] "=&s"(flags), [tm] "=&s"(tmp)
        : [c3] "v"(c3), [s2] "v"(s2), [fmax] "v"(max_freq), [beal] "v"(beal),
          [k2pi] "s"(0x1.45F306DC9C883p-1), [nhpi] "s"(-0x1.921FB54442D18p0), [c4] "s"(0x1.99343027bf8c3p-16),
          [s3] "s"(-0x1.994eb3774cf24p-13), [c2] "s"(0x1.55553e1068f19p-5), [s1] "s"(-0x1.555545995a603p-3),
          [c1] "s"(-0x1.ffffffd0c621cp-2), [fmin] "s"(min_freq), [tau] "s"(TAU_F), [absm] "s"(0x7fffffffu), [ign] "s"(ign)
        : "vcc", "scc", "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109",
          "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122",
          "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135",
          "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143");
    flags_out = flags;
    return groups;
}

#undef QPSK_DA
#undef QPSK_ZA

/*
 * The same stream running THROUGH the chunk hand-overs of the pipeline kernels' rings (rx_fused.hip): symbol ring and
 * record ring of 128 symbols = 8 groups per lane (two 64-symbol chunks), group k at ring position k % 8.  What the
 * serial wave did between two costas_asm_run() calls per chunk -- ready[] poll, acquire, address set-up, an exposed
 * LDS read of the first symbol pair, waiting out the last record write, release, consumed -- cost ~0.5-0.7 k cycles
 * per chunk, every cycle of it on the kernel's critical path (config 2 sits on this wave).  Here:
 *   - every group reads the lane's producer counter ready[] (one 4-byte LDS read per 16 steps, in the shadow of the
 *     steps); at a chunk boundary the value read at the START of the chunk's last group decides: LDS operations of
 *     a wave execute in order, so everything read after a counter read that showed the next chunk is that chunk
 *     -- including the usual fetch of the next group's first symbol pair two steps before the group ends;
 *   - consumed = chunk + 1 is one LDS write by lane 0 behind the chunk's last record write (same order argument:
 *     the FIR waves read the counter, then the records);
 *   - the stream stops at a boundary whose next chunk was not yet there (the caller waits, comes back), at kend,
 *     or inside a group it abandons (as costas_asm_run: state restored, caller redoes that group).
 * k (in/out): absolute group number, a multiple of 4 on entry unless the caller resumes behind a group it redid.
 * Registers: as costas_asm_run, plus v132 / v134 / v124 = symbol address, record address, address of the next
 * group's first symbol pair, v125 = the counter read.
 */
/* INVARIANT of the ring stream's LDS traffic (nothing checks it at compile time -- keep it when editing; the test
 * test_stream_across_ring_handovers_takes_its_fallbacks exercises every hand-over): LDS operations of a wave complete in
 * issue order and lgkmcnt counts them, so the wait in front of a group's first symbol use, `s_waitcnt lgkmcnt(2)`, is right
 * exactly while at most TWO LDS operations are issued between the fetch of that group's first pair (QPSK_RDN, in step 14
 * of the group before) and the wait: the last record write of that group (QPSK_QW(48), step 15) and this group's counter
 * read (ds_read_b32 v125).  The optional `consumed` write at a chunk boundary sits between them in program order only on
 * paths that LEAVE the stream or re-enter at label 2 after it -- there three operations follow the fetch and the wait
 * lets two of them stay outstanding, i.e. it still covers the fetch (the oldest).  One more LDS instruction anywhere
 * between QPSK_RDN and that wait needs lgkmcnt(3), one fewer lgkmcnt(1); no ordering fence is needed for the hand-over
 * itself (counter read before data reads, data writes before counter write, same wave, in order). */
#define QPSK_DA "v132"
#define QPSK_ZA "v134"
#define QPSK_RDN "ds_read_b128 v[120:123], v124\n\t"
/* where the group loop's head sits relative to the 64-byte lines, per stream (tools/ubench_step.py --align) */
#define QPSK_RING_ALIGN_1 ".p2align 6\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n"
#define QPSK_RING_ALIGN_P ".p2align 6\n\ts_nop 0\n\ts_nop 0\n"
#define QPSK_RING_TEXT(HC, HD, BD, WH, INIT, ALIGN)                                                                           \
    "v_mov_b32 v128, 0x54442d18\n\t"        /* 2*pi = 0x401921FB54442D18 */                                            \
    "v_mov_b32 v127, 0x401921fb\n\t" INIT                                                                                   \
    "s_and_b32 %[t0], %[k], 7\n\t"                                                                                     \
    "s_lshl_b32 %[t0], %[t0], 7\n\t"                                                                                   \
    "v_add_u32_e32 v132, %[t0], %[db]\n\t"                                                                             \
    "ds_read_b128 v[120:123], v132\n\t"                                                                                \
    "s_mov_b64 %[fl], 0\n\t"                                                                                           \
    "s_waitcnt lgkmcnt(0)\n"                                                                                           \
    /* the group loop's head 12 bytes behind a 64-byte boundary: a lone wave is limited by instruction fetch (header), and where \
     * the 8-byte instructions of the 16-step body fall relative to the 32-byte fetch lines is worth 4 % -- 157.2 cycles \
     * per step at this offset, 160.9 as the compiler placed it, 163.4 at the worst (tools/ubench_step.py --align,     \
     * profiles/r03_step_cost.txt; re-measure after any edit of the stream) */                                         \
    ALIGN                                                                                                              \
    "2:\n\t"                                                                                                           \
    /* ring addresses of group k and of group k + 1's first pair; the producer counter */                              \
    "s_and_b32 %[t0], %[k], 7\n\t"                                                                                     \
    "s_lshl_b32 %[t1], %[t0], 7\n\t"                                                                                   \
    "v_add_u32_e32 v132, %[t1], %[db]\n\t"                                                                             \
    "s_lshl_b32 %[t1], %[t0], 6\n\t"                                                                                   \
    "v_add_u32_e32 v134, %[t1], %[zb]\n\t"                                                                             \
    "s_add_u32 %[t0], %[k], 1\n\t"                                                                                     \
    "s_and_b32 %[t0], %[t0], 7\n\t"                                                                                    \
    "s_lshl_b32 %[t0], %[t0], 7\n\t"                                                                                   \
    "v_add_u32_e32 v124, %[t0], %[db]\n\t"                                                                             \
    "ds_read_b32 v125, %[ra]\n\t"                                                                                      \
    "v_mov_b32 v130, %[p]\n\t"                                                                                         \
    "v_mov_b32 v131, %[f]\n\t"                                                                                         \
    "v_mov_b32 v118, %[f]\n\t"            /* the frequency accumulates in place (v_fmac) */                            \
    "v_mov_b32 v126, 0x7f800000\n\t"                                                                                   \
    "v_mov_b32 v140, %[p]\n\t"                                                                                         \
    /* steps 0..3; outstanding in front of the first pair's use: the last group's record write and the counter read */ \
    HC("v140")                                                                                            \
    BD("v140", "v141", "v[120:121]", "s_waitcnt lgkmcnt(2)\n\t", QPSK_RDB(16), "")                                     \
    HD("v141", "101", "201")                                                                                           \
    BD("v141", "v142", "v[122:123]", "", "", "")                                                                       \
    HD("v142", "102", "202")                                                                                           \
    BD("v142", "v143", "v[136:137]", QPSK_WAIT0, QPSK_RDA(32), "")                                                     \
    HD("v143", "103", "203")                                                                                           \
    BD("v143", "v140", "v[138:139]", "", "", QPSK_QW(0))                                                               \
    QPSK_STEP_QUAD_(HD, BD, "v[120:121]", "v[122:123]", "v[136:137]", "v[138:139]", QPSK_RDB(48), QPSK_RDA(64), 16, "v140", "04", "05", "06", "07") \
    QPSK_STEP_QUAD_(HD, BD, "v[120:121]", "v[122:123]", "v[136:137]", "v[138:139]", QPSK_RDB(80), QPSK_RDA(96), 32, "v140", "08", "09", "10", "11") \
    QPSK_STEP_QUAD_(HD, BD, "v[120:121]", "v[122:123]", "v[136:137]", "v[138:139]", QPSK_RDB(112), QPSK_RDN, 48, "%[p]", "12", "13", "14", "15") \
    QPSK_TAIL("%[p]", "%[f]", "116", "216")                                                                            \
    "v_cmp_eq_f32_e64 %[tm], 0, v126\n\t"                                                                              \
    "s_andn2_b64 %[tm], %[tm], %[ign]\n\t"   /* lanes the caller has looked at: nothing but +0.0 symbols ahead of them (below) */ \
    "s_or_b64 %[fl], %[fl], %[tm]\n\t"                                                                                 \
    "s_cmp_lg_u64 %[fl], 0\n\t"                                                                                        \
    "s_cbranch_scc1 3f\n\t"                                                                                            \
    "s_add_u32 %[k], %[k], 1\n\t"                                                                                      \
    "s_and_b32 %[t0], %[k], 3\n\t"                                                                                     \
    "s_cmp_lg_u32 %[t0], 0\n\t"                                                                                        \
    "s_cbranch_scc1 2b\n\t"                 /* inside a chunk (kend is a multiple of 4) */                             \
    /* chunk k / 4 - 1 is done: consumed = k / 4, by lane 0, behind the record writes */                               \
    "s_lshr_b32 %[t0], %[k], 2\n\t"                                                                                    \
    "v_mov_b32 v104, %[t0]\n\t"                                                                                        \
    "s_mov_b64 %[ex], exec\n\t"                                                                                        \
    "s_mov_b64 exec, 1\n\t"                                                                                            \
    "ds_write_b32 %[ca], v104\n\t"                                                                                     \
    "s_mov_b64 exec, %[ex]\n\t"                                                                                        \
    "s_cmp_ge_u32 %[k], %[ke]\n\t"                                                                                     \
    "s_cbranch_scc1 4f\n\t"                                                                                            \
    /* next chunk there?  ready[] >= k / 4 + 1 in every lane, as read at the start of the group just done */           \
    "v_cmp_le_i32_e64 %[tm], v125, %[t0]\n\t"                                                                          \
    "s_cmp_lg_u64 %[tm], 0\n\t"                                                                                        \
    "s_cbranch_scc0 2b\n\t"                                                                                            \
    "s_branch 4f\n"                                                                                                    \
    "3:\n\t"                                                                                                           \
    "v_mov_b32 %[p], v130\n\t"                                                                                         \
    "v_mov_b32 %[f], v131\n\t"                                                                                         \
    "s_branch 4f\n"                                                                                                    \
    WH("v141", "101", "201")                                                                               \
    WH("v142", "102", "202")                                                                               \
    WH("v143", "103", "203")                                                                               \
    WH("v140", "104", "204")                                                                               \
    WH("v141", "105", "205")                                                                               \
    WH("v142", "106", "206")                                                                               \
    WH("v143", "107", "207")                                                                               \
    WH("v140", "108", "208")                                                                               \
    WH("v141", "109", "209")                                                                               \
    WH("v142", "110", "210")                                                                               \
    WH("v143", "111", "211")                                                                               \
    WH("v140", "112", "212")                                                                               \
    WH("v141", "113", "213")                                                                               \
    WH("v142", "114", "214")                                                                               \
    WH("v143", "115", "215")                                                                               \
    QPSK_WRAP_TAIL("%[p]", "116", "216")                                                                               \
    "4:\n\t"                                                                                                           \
    "s_waitcnt lgkmcnt(0)"

__device__ __forceinline__ void costas_asm_run_ring(float &phase, float &freq, unsigned d_base, unsigned z_base,
                                                    unsigned ready_addr, unsigned consumed_addr, unsigned &k, unsigned kend,
                                                    float alpha, float beta, float min_freq, float max_freq,
                                                    unsigned long long &flags_out, unsigned long long ign = 0ull)
{
    unsigned long long flags, tmp, ex;
    ign = uniform64(ign);
    kend = __builtin_amdgcn_readfirstlane(kend);
    unsigned t0, t1;
    const double c3 = -0x1.6c087e89a359dp-10, s2 = 0x1.1107605230bc4p-7;
    double beal;
    {
        const float2 ba = make_float2(beta, alpha);
        __builtin_memcpy(&beal, &ba, 8);
    }
    asm volatile(
        QPSK_RING_TEXT(QPSK_HEAD_CHAIN, QPSK_HEAD_DEFERRED, QPSK_BODY, QPSK_WRAP_HEAD, "", QPSK_RING_ALIGN_1)
        : [p] "+v"(phase), [f] "+v"(freq), [k] "+s"(k), [fl] "=&s"(flags), [tm] "=&s"(tmp), [ex] "=&s"(ex),
          [t0] "=&s"(t0), [t1] "=&s"(t1)
        : [db] "v"(d_base), [zb] "v"(z_base), [ra] "v"(ready_addr), [ca] "v"(consumed_addr), [ke] "s"(kend),
          [c3] "v"(c3), [s2] "v"(s2), [fmax] "v"(max_freq), [beal] "v"(beal),
          [k2pi] "s"(0x1.45F306DC9C883p-1), [nhpi] "s"(-0x1.921FB54442D18p0), [c4] "s"(0x1.99343027bf8c3p-16),
          [s3] "s"(-0x1.994eb3774cf24p-13), [c2] "s"(0x1.55553e1068f19p-5), [s1] "s"(-0x1.555545995a603p-3),
          [c1] "s"(-0x1.ffffffd0c621cp-2), [fmin] "s"(min_freq), [tau] "s"(TAU_F), [absm] "s"(0x7fffffffu), [ign] "s"(ign)
        : "vcc", "scc", "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109",
          "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122",
          "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135",
          "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143");
    flags_out = flags;
}

/*
 * The paired-lane form of the ring stream (QPSK_BODY_P): lanes 2f and 2f + 1 both carry loop f -- the same phase, frequency,
 * gains, ring addresses (two lanes read one address: a broadcast; both write the same records to the same address) -- and
 * `odd` tells a lane which of the two it is.  Everything else as costas_asm_run_ring.  For workgroups whose loops fill at
 * most half the wave (rx_lean_kernel up to 16 frames per workgroup: BASELINE config 2).
 */
__device__ __forceinline__ void costas_asm_run_ring_pair(float &phase, float &freq, unsigned d_base, unsigned z_base,
                                                         unsigned ready_addr, unsigned consumed_addr, unsigned &k, unsigned kend,
                                                         float alpha, float beta, float min_freq, float max_freq, bool odd,
                                                         unsigned long long &flags_out, unsigned long long ign = 0ull)
{
    unsigned long long flags, tmp, ex;
    ign = uniform64(ign);
    kend = __builtin_amdgcn_readfirstlane(kend);
    unsigned t0, t1;
    /* per-lane coefficients: the cosine's in the even lane, the sine's in the odd one (QPSK_BODY_P) */
    const double ka = odd ? 0.0 : 0x1.99343027bf8c3p-16, kb = odd ? -0x1.994eb3774cf24p-13 : -0x1.6c087e89a359dp-10;
    const double kc = odd ? 0x1.1107605230bc4p-7 : 0x1.55553e1068f19p-5, kd = odd ? -0x1.555545995a603p-3 : -0x1.ffffffd0c621cp-2;
    const double km = odd ? 1.0 : 0.0, kj = odd ? -0.0 : 1.0;
    const float sg = odd ? -1.0f : 1.0f;
    double beal;
    {
        const float2 ba = make_float2(beta, alpha);
        __builtin_memcpy(&beal, &ba, 8);
    }
    asm volatile(
        /* the first step's zero test looks at the T of "the step before": anything but a zero */
        QPSK_RING_TEXT(QPSK_HEAD_CHAIN_P, QPSK_HEAD_DEFERRED_P, QPSK_BODY_P, QPSK_WRAP_HEAD_P, "v_mov_b32 v114, 1.0\n\tv_mov_b32 v115, 1.0\n\t", QPSK_RING_ALIGN_P)
        : [p] "+v"(phase), [f] "+v"(freq), [k] "+s"(k), [fl] "=&s"(flags), [tm] "=&s"(tmp), [ex] "=&s"(ex),
          [t0] "=&s"(t0), [t1] "=&s"(t1)
        : [db] "v"(d_base), [zb] "v"(z_base), [ra] "v"(ready_addr), [ca] "v"(consumed_addr), [ke] "s"(kend),
          [ka] "v"(ka), [kb] "v"(kb), [kc] "v"(kc), [kd] "v"(kd), [km] "v"(km), [kj] "v"(kj), [sg] "v"(sg),
          [fmax] "v"(max_freq), [beal] "v"(beal),
          [k2pi] "s"(0x1.45F306DC9C883p-1), [nhpi] "s"(-0x1.921FB54442D18p0),
          [fmin] "s"(min_freq), [tau] "s"(TAU_F), [absm] "s"(0x7fffffffu), [ign] "s"(ign)
        : "vcc", "scc", "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109",
          "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122",
          "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135",
          "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143");
    flags_out = flags;
}
#undef QPSK_DA
#undef QPSK_ZA

} // namespace qpsk
#endif
