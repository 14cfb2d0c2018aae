#!/usr/bin/env python3
"""Generator of qpsk_amd/csrc/fir_r2_asm.h and fir_r4_asm.h: the RRC FIR step of the pipeline kernels' FIR waves
(R = 2 or 4 consecutive decimated outputs per lane) as ONE hand-scheduled gfx950 instruction stream (rrc_fir.c:22-26
evaluated at the decimated outputs a lane owns).

    python tools/gen_fir_asm.py 1 100 2 > qpsk_amd/csrc/fir_r2_asm.h      # depth, first VGPR, R [, step]
    python tools/gen_fir_asm.py 1 144 4 > qpsk_amd/csrc/fir_r4_asm.h
    python tools/gen_fir_asm.py 1 80 8 1 > qpsk_amd/csrc/fir_full8_asm.h   # full rate: 8 CONSECUTIVE outputs per lane (v80..v167:
                                                                            # timing_scan_kernel runs three waves per SIMD, 168 VGPRs)
    python tools/gen_fir_asm.py 1 40 8 1 sgpr > qpsk_amd/csrc/fir_full8s_asm.h   # the same sum with the 64 distinct taps of a SYMMETRIC
                                            # filter in SGPRs s36..s99 (scalar loads at the head of the stream): no tap reads, no tap
                                            # registers -- v40..v103, so that a kernel can run four or five waves per SIMD
The committed headers are checked against these command lines by tests/test_generated_headers.py.

Why a generated stream and not C++: the compiler's version of the same sum (asm-pinned product/add order) carries
~60 v_mov and ~60 s_nop per 508 packed multiply/adds, and fetches LDS only one block of 8 window positions ahead --
a FIR wave alone on its SIMD then needs ~7 cycles per packed instruction (measured), against the SIMD's 4.
Here: no moves (taps are picked out of their register pairs with op_sel), no nops, window values and taps fetched
DEPTH blocks ahead with counted lgkmcnt waits (LDS returns in order).

Arithmetic: symbol 0 of the lane sums window position t times tap t, symbol 1 position t times tap t - 8, each in
tap order 0..126 into its own packed (re, im) accumulator, multiply and add unfused (the reference is built without
contraction): exactly the C++ step it replaces; the parity tests compare every output bit with the oracle.

Window image (rx_fused.hip, namespace pipe2): position p at float2 slot p + 2*(p/16) from the lane's base, so that
positions (t, t+1), t even, are one aligned 16-byte word; taps as 128 floats in LDS (tap 127 = 0, never used).
"""
import sys

NTAPS, C = 127, 8
DEPTH = int(sys.argv[1]) if len(sys.argv) > 1 else 2
V0 = int(sys.argv[2]) if len(sys.argv) > 2 else 76   # first VGPR of the block's fixed registers (even)
R = int(sys.argv[3]) if len(sys.argv) > 3 else 2     # outputs per lane
STEP = int(sys.argv[4]) if len(sys.argv) > 4 else C  # samples between a lane's outputs: CYCLES (decimated) or 1 (full rate)
SGPR = len(sys.argv) > 5 and sys.argv[5] == "sgpr"   # taps as SGPR operands (symmetric filters only: tap k = tap 126 - k, host-checked)
TAP0 = 36                             # s36..s99: taps 0..63 (s_load_dwordx16 wants a multiple of 4; as in tools/gen_lean_asm.py)
TSTEPS = NTAPS + STEP * (R - 1)       # window positions a lane sweeps: 135 (R = 2), 151 (R = 4), 134 (full rate, R = 8)
NB = (TSTEPS + C - 1) // C            # blocks of 8 positions
REACH = (STEP * (R - 1) + C - 1) // C # tap groups an output reaches back behind the block's own
NW, NG = DEPTH + 1, (0 if SGPR else DEPTH + 1 + REACH) # live window blocks / tap groups
PAD = R * STEP                        # lanes are PAD positions apart: position p at slot p + 2*(p/PAD)
NAME = ("fir_r%d" % R if STEP == C else "fir_full%d" % R) + ("s" if SGPR else "")

W0 = V0                               # window blocks: NW x 16 dwords
T0 = W0 + 16 * NW                     # tap groups:    NG x 8 dwords
P0 = T0 + 8 * NG                      # products:      2 positions x R symbols, one pair each
VEND = P0 + 4 * R                     # first register NOT used


def slot_of(p):
    return p + 2 * (p // PAD)


def wreg(t):
    """VGPR pair holding window position t"""
    b, u = divmod(t, C)
    r = W0 + 16 * (b % NW) + 2 * u
    return "v[%d:%d]" % (r, r + 1)


def tap_operand(k):
    """(register pair, op_sel text) that broadcasts tap k to both halves of a packed multiply (tap = src0)"""
    if SGPR:
        r = TAP0 + (k if k <= 63 else 126 - k)
        if r % 2 == 0:
            return "s[%d:%d]" % (r, r + 1), "op_sel_hi:[0,1]"
        return "s[%d:%d]" % (r - 1, r), "op_sel:[1,0]"
    g, i = divmod(k, C)
    r = T0 + 8 * (g % NG) + i
    if r % 2 == 0:
        return "v[%d:%d]" % (r, r + 1), "op_sel_hi:[0,1]"
    return "v[%d:%d]" % (r - 1, r), "op_sel:[1,0]"


def fetch(b, out):
    """LDS reads of block b: its window positions (pairs, 16 bytes each) and tap group b"""
    n = 0
    for u in range(0, C, 2):
        t = b * C + u
        if t < TSTEPS:
            r = W0 + 16 * (b % NW) + 2 * u
            out.append('"ds_read_b128 v[%d:%d], %%[rd] offset:%d\\n\\t"' % (r, r + 3, 8 * slot_of(t)))
            n += 1
    if b * C < NTAPS and not SGPR:
        r = T0 + 8 * (b % NG)
        out.append('"ds_read_b128 v[%d:%d], %%[tp] offset:%d\\n\\t"' % (r, r + 3, 32 * b))
        out.append('"ds_read_b128 v[%d:%d], %%[tp] offset:%d\\n\\t"' % (r + 4, r + 7, 32 * b + 16))
        n += 2
    return n


def main():
    out = []
    reads = {}     # block -> number of LDS reads issued for it
    if SGPR:       # the 64 distinct taps: four scalar loads (the scalar cache serves every wave after the first)
        for i in range(4):
            out.append('"s_load_dwordx16 s[%d:%d], %%[tp], 0x%x\\n\\t"' % (TAP0 + 16 * i, TAP0 + 16 * i + 15, 64 * i))
    for d in range(min(DEPTH, NB)):
        reads[d] = fetch(d, out)
    out.append('"' + "".join("v_mov_b64 %%[a%d], 0\\n\\t" % r for r in range(R)) + '"')     # y = 0 (rrc_fir.c:22)
    acc = ["%%[a%d]" % r for r in range(R)]
    for b in range(NB):
        if b + DEPTH < NB:
            reads[b + DEPTH] = fetch(b + DEPTH, out)
        later = sum(reads.get(x, 0) for x in range(b + 1, min(NB, b + DEPTH + 1)))
        if SGPR and b == 0:
            later = 0      # scalar loads share lgkmcnt with the LDS reads and return out of order: only 0 covers them
        out.append('"s_waitcnt lgkmcnt(%d)\\n\\t"' % later)
        for u in range(0, C, 2):
            muls, adds = [], []
            np_ = 0
            for t in (b * C + u, b * C + u + 1):
                if t >= TSTEPS:
                    continue
                for sym in range(R):
                    k = t - STEP * sym
                    if 0 <= k < NTAPS:
                        treg, sel = tap_operand(k)
                        p = "v[%d:%d]" % (P0 + 2 * np_, P0 + 2 * np_ + 1)
                        muls.append('"v_pk_mul_f32 %s, %s, %s %s\\n\\t"' % (p, treg, wreg(t), sel))
                        adds.append('"v_pk_add_f32 %s, %s, %s\\n\\t"' % (acc[sym], acc[sym], p))
                        np_ += 1
            out += muls + adds
    # the static guard of tools/gen_lean_asm.py (round 6): behind their four scalar loads the tap SGPRs are live to the end of the stream,
    # and every register the stream writes must be one it declares as clobbered
    from gen_lean_asm import written_and_read
    taps_live = set()
    for text in out:
        for line in text.strip('"').split("\\n\\t"):
            dst, _ = written_and_read(line)
            if line.startswith("s_load_dwordx16"):
                taps_live.update(dst)
                continue
            for r in dst:
                assert r not in taps_live, "gen_fir_asm.py: `%s` writes %s, which holds a tap" % (line, r)
                assert (r[0] == "v" and V0 <= int(r[1:]) < VEND), "gen_fir_asm.py: `%s` writes %s, outside the stream's registers v%d..v%d" % (line, r, V0, VEND - 1)
    assert len(taps_live) == (64 if SGPR else 0)
    body = "\n        ".join(out)
    clob = ", ".join('"v%d"' % r for r in range(V0, VEND))
    if SGPR:
        clob += ", " + ", ".join('"s%d"' % r for r in range(TAP0, TAP0 + 64))
    nmul = sum(1 for x in out if "v_pk_mul" in x)
    args = ", ".join("v2f &acc%d" % r for r in range(R))
    decl = ", ".join("a%d" % r for r in range(R))
    outs = ", ".join('[a%d] "=&v"(a%d)' % (r, r) for r in range(R))
    copy = "\n".join("    acc%d = a%d;" % (r, r) for r in range(R))
    print('''/*
 * %(NAME)s_asm.h -- GENERATED by tools/gen_fir_asm.py %(D)d %(V0)d %(R)d %(STEP)d%(SFX)s; do not edit.
 *
 * The RRC FIR step of a FIR wave with %(R)d outputs per lane, %(STEP)d sample(s) apart (rrc_fir.c:22-26 at those
 * outputs) as one hand-scheduled gfx950 instruction stream: %(nmul)d packed multiplies and as many packed adds,
 * unfused, taps 0..126 in order into one (re, im) accumulator per symbol; window pairs (one aligned 16-byte word
 * per two positions: position p at slot p + 2 (p / %(PAD)d) from the lane's base) %(TAPDOC)s fetched from LDS
 * %(D)d block(s) of 8 positions ahead (counted lgkmcnt waits), no register moves, no nops.  See the generator.
 * Fixed registers v%(V0)d..v%(VL)d are scratch owned by the block.%(SGPRDOC)s
 */
#ifndef QPSK_%(UNAME)s_ASM_H
#define QPSK_%(UNAME)s_ASM_H

#include "qpsk_device.h"

namespace qpsk {

constexpr int %(UNAME)s_ASM_FIRST_VGPR = %(V0)d, %(UNAME)s_ASM_END_VGPR = %(VEND)d;

/* rd_addr: LDS byte address of the lane's window position 0; %(TAPARGDOC)s */
__device__ __forceinline__ void %(NAME)s_asm(unsigned rd_addr, %(TAPARG)s, %(args)s)
{
    v2f %(decl)s;
    asm volatile(
        %(body)s
        "s_waitcnt lgkmcnt(0)"
        : %(outs)s
        : [rd] "v"(rd_addr), %(TAPIN)s
        : "memory", %(clob)s);
%(copy)s
}

} // namespace qpsk
#endif''' % dict(NAME=NAME, UNAME=NAME.upper(), STEP=STEP, R=R, D=DEPTH, V0=V0, VL=VEND - 1, VEND=VEND, nmul=nmul, PAD=PAD, args=args, decl=decl, body=body, outs=outs,
                  clob=clob, copy=copy, SFX=" sgpr" if SGPR else "",
                  TAPDOC="are" if SGPR else "and tap groups",
                  SGPRDOC=("\n * The 64 distinct taps of the SYMMETRIC filter (tap k = tap 126 - k: checked by the host before a kernel with this"
                           "\n * stream is chosen) are SGPR operands s%d..s%d, loaded by four s_load_dwordx16 at the head of the stream." % (TAP0, TAP0 + 63)) if SGPR else "",
                  TAPARGDOC="taps_g: the 127 taps in global memory (64-byte aligned, at least 64 readable floats)" if SGPR
                  else "tap_addr: LDS byte address of the 128 taps",
                  TAPARG="const float *taps_g" if SGPR else "unsigned tap_addr",
                  TAPIN='[tp] "s"(taps_g)' if SGPR else '[tp] "v"(tap_addr)'))


if __name__ == "__main__":
    main()
