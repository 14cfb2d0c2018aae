#!/usr/bin/env python3
"""Builds tools/ubench_step.hip against variants of qpsk_amd/csrc/costas_asm.h (text substitutions on a copy: the
product header is not touched) -> build_ubench/step/<variant>.bin; run them on the GPU box (this script there, hipcc is
in the image: build_ubench/ is in .gpurunignore) with
    python tools/ubench_step.py [--align] && for b in build_ubench/step/*.bin; do $b $(basename $b .bin); done
What each variant removes from the serial wave's step tells what that piece costs the wave."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = open(os.path.join(ROOT, "qpsk_amd", "csrc", "costas_asm.h")).read()
OUT = os.path.join(ROOT, "build_ubench", "step")
os.makedirs(OUT, exist_ok=True)


def no_lds(s):
    s = re.sub(r'#define QPSK_RDA\(OFF\) .*', '#define QPSK_RDA(OFF) ""', s)
    s = re.sub(r'#define QPSK_RDB\(OFF\) .*', '#define QPSK_RDB(OFF) ""', s)
    s = re.sub(r'#define QPSK_QW\(OFF\) .*', '#define QPSK_QW(OFF) ""', s)
    s = re.sub(r'#define QPSK_RDN .*', '#define QPSK_RDN ""', s)
    s = s.replace('"ds_read_b32 v125, %[ra]\\n\\t"', '"v_mov_b32 v125, 0x7fffffff\\n\\t"')
    # both symbol register sets hold the first pair for good (the entry read stays)
    s = s.replace('"s_waitcnt lgkmcnt(0)\\n"', '"s_waitcnt lgkmcnt(0)\\n\\tv_mov_b32 v136, v120\\n\\tv_mov_b32 v137, v121\\n\\tv_mov_b32 v138, v122\\n\\tv_mov_b32 v139, v123\\n"', 1) if "QPSK_RING_TEXT" in s else s.replace('"s_waitcnt lgkmcnt(0)\\n"\n        "2:\\n\\t"', '"s_waitcnt lgkmcnt(0)\\n\\tv_mov_b32 v136, v120\\n\\tv_mov_b32 v137, v121\\n\\tv_mov_b32 v138, v122\\n\\tv_mov_b32 v139, v123\\n"\n        "2:\\n\\t"')
    return s


def no_branch(s):
    return s.replace('"s_cbranch_vccnz " LW "f\\n"', '"\\n"')


def no_fillers(s):
    """the stream without the previous step's leftovers (clamp in place, zero test) and without the wrap test -- timing only,
    the arithmetic is no longer the loop's"""
    s = s.replace('"v_cmp_ge_f32_e64 vcc, |" PIN "|, %[tau]\\n\\t"', '')
    assert '"v_med3_f32 v118, v118, %[fmin], %[fmax]\\n\\t"' in s
    s = s.replace('"v_med3_f32 v118, v118, %[fmin], %[fmax]\\n\\t"', '')
    s = s.replace('"v_min3_f32 v126, v126, |v114|, |v115|\\n\\t"', '')
    return s


def no_sign(s):
    s = s.replace('"v_xor_b32_e32 v108, v114, v115\\n\\t"', '')
    return s.replace('"v_bfi_b32 v108, %[absm], 1.0, v108\\n\\t"', '')


def nops_for_fillers(s):
    """the paired stream with s_nop 0 where the clamp and the zero test separate the conversion from the DPP read of it (timing only)"""
    i = s.index("#define QPSK_BODY_P(")
    j = s.index("QPSK_BODY_F32(PIN", i)
    body = s[i:j].replace('"v_med3_f32 v118, v118, %[fmin], %[fmax]\\n\\t"', '"s_nop 0\\n\\t"')
    body = body.replace('"v_min3_f32 v126, v126, |v114|, |v115|\\n\\t"', '"s_nop 0\\n\\t"')
    return s[:i] + body + s[j:]


def aligned(pad, which):
    """the group loop's head on a 64-byte boundary plus `pad` s_nops (4 bytes each): QPSK_RING_ALIGN_1 (one lane per loop) or _P (paired)"""
    def f(s):
        line = '#define QPSK_RING_ALIGN_%s ".p2align 6\\n%s"' % (which, "\\ts_nop 0\\n" * pad)
        return re.sub(r'#define QPSK_RING_ALIGN_%s .*' % which, lambda m: line, s)
    return f


STEP_P = ['"v_cvt_f64_f32 v[106:107], " PIN "\\n\\t"',
          '"v_mul_f64 v[104:105], v[106:107], %[k2pi]\\n\\t"', '"v_rndne_f64_e32 v[104:105], v[104:105]\\n\\t"',
          '"v_fmac_f64_e32 v[106:107], %[nhpi], v[104:105]\\n\\t"', '"v_mul_f64 v[108:109], v[106:107], v[106:107]\\n\\t"',
          '"v_fma_f64 v[100:101], v[106:107], %[km], %[kj]\\n\\t"', '"v_fma_f64 v[110:111], v[108:109], %[ka], %[kb]\\n\\t"',
          '"v_mul_f64 v[104:105], v[108:109], v[100:101]\\n\\t"', '"v_fma_f64 v[110:111], v[108:109], v[110:111], %[kc]\\n\\t"',
          '"v_fma_f64 v[110:111], v[108:109], v[110:111], %[kd]\\n\\t"',
          '"v_fmac_f64_e32 v[100:101], v[104:105], v[110:111]\\n\\t"', '"v_cvt_f32_f64 v110, v[100:101]\\n\\t"',
          '"v_med3_f32 v118, v118, %[fmin], %[fmax]\\n\\t"', '"v_min3_f32 v126, v126, |v114|, |v115|\\n\\t"',
          '"v_mul_f32_dpp v112, v110, %[sg] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\\n\\t"']


def split_step(n):
    """the paired step with its wrap branch behind the 2 pi test and the next n instructions (the out-of-line wrap redoes those)"""
    def f(s):
        i = s.index("#define QPSK_STEP_P_PRE(CMP, PIN)")
        j = s.index("#define QPSK_HEAD_CHAIN_P(PIN)")
        s = s[:i] + "#define QPSK_STEP_P_PRE(CMP, PIN) CMP " + " ".join(STEP_P[:n]) + "\n" + s[j:]
        i = s.index("#define QPSK_BODY_P(PIN, POUT, DREG, WAIT, READ, QW)")
        j = s.index("/* the leftovers of a group's LAST step")
        return s[:i] + "#define QPSK_BODY_P(PIN, POUT, DREG, WAIT, READ, QW) " + " ".join(STEP_P[n:]) + " QPSK_BODY_F32(PIN, POUT, DREG, WAIT, READ, QW)\n\n" + s[j:]
    return f


PADS = (0, 1, 2, 3, 4, 5, 6, 7, 9, 11, 13, 15) if "--align" in sys.argv else ()
# (name, header text, transformation, extra compiler flags)
R05 = open(os.path.join(ROOT, "tools", "ref", "costas_asm_r05.h")).read()      # round 5's stream (28 VALU, magic-number rounding), for the comparison
VARIANTS = [
    ("r05_as_shipped", R05, lambda s: s, []),
    ("0_as_shipped", SRC, lambda s: s, []),
    ("0p_paired_as_shipped", SRC, lambda s: s, ["-DPAIRED"]),
    *[("a%02d_one_lane" % k, SRC, aligned(k, "1"), []) for k in PADS],
    *[("b%02d_paired" % k, SRC, aligned(k, "P"), ["-DPAIRED"]) for k in PADS],
    *[("c%02d_paired_branch_%d_behind_the_test" % (k, k), SRC, split_step(k), ["-DPAIRED"]) for k in (5, 7, 9, 10, 11, 12, 13, 15)],
    ("1_no_lds", SRC, no_lds, []),
    ("1p_paired_no_lds", SRC, no_lds, ["-DPAIRED"]),
    ("2_no_wrap_branch", SRC, no_branch, []),
    ("2p_paired_no_wrap_branch", SRC, no_branch, ["-DPAIRED"]),
    ("3_no_lds_no_branch", SRC, lambda s: no_branch(no_lds(s)), []),
    ("3p_paired_no_lds_no_branch", SRC, lambda s: no_branch(no_lds(s)), ["-DPAIRED"]),
    ("4_no_lds_no_branch_no_clamp_zero_wrap_tests", SRC, lambda s: no_fillers(no_branch(no_lds(s))), []),
    ("4p_paired_nops_for_clamp_and_zero_test", SRC, nops_for_fillers, ["-DPAIRED"]),
    ("5_also_no_sign_instructions", SRC, lambda s: no_sign(no_fillers(no_branch(no_lds(s)))), []),
]

def build(v):
    name, src, f, flags = v
    hdr = os.path.join(OUT, "costas_asm_%s.h" % name)
    open(hdr, "w").write(f(src))
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-I", os.path.join(ROOT, "qpsk_amd", "csrc"),
           '-DCOSTAS_HEADER="%s"' % hdr, os.path.join(ROOT, "tools", "ubench_step.hip"), "-o", os.path.join(OUT, name + ".bin")] + flags
    r = subprocess.run(cmd, capture_output=True, text=True)
    return name, r.returncode, r.stderr[-2000:]


from concurrent.futures import ThreadPoolExecutor
with ThreadPoolExecutor(max_workers=min(12, os.cpu_count() or 1)) as ex:
    for name, rc, err in ex.map(build, VARIANTS):
        print(name, "ok" if rc == 0 else err)
        if rc:
            sys.exit(1)
