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
    s = s.replace('"s_waitcnt lgkmcnt(0)\\n"\n        "2:\\n\\t"', '"s_waitcnt lgkmcnt(0)\\n\\tv_mov_b32 v136, v120\\n\\tv_mov_b32 v137, v121\\n\\tv_mov_b32 v138, v122\\n\\tv_mov_b32 v139, v123\\n"\n        "2:\\n\\t"')
    return s


def no_branch(s):
    return s.replace('"s_cbranch_vccnz " LW "f\\n"', '"\\n"')


def no_fillers(s):
    """the head without the previous step's leftovers (clamp in place, zero test) and without the wrap test -- timing only,
    the arithmetic is no longer the loop's"""
    s = s.replace('"v_cmp_ge_f32_e64 vcc, |" PIN "|, %[tau]\\n\\t"', '')
    assert '"v_med3_f32 v118, v118, %[fmin], %[fmax]\\n\\t"' in s
    s = s.replace('"v_med3_f32 v118, v118, %[fmin], %[fmax]\\n\\t"', '')
    s = s.replace('"v_min3_f32 v126, v126, |v114|, |v115|\\n\\t"', '')
    return s


def no_sign(s):
    s = s.replace('"v_xor_b32_e32 v108, v114, v115\\n\\t"', '')
    return s.replace('"v_bfi_b32 v108, %[absm], 1.0, v108\\n\\t"', '')


def aligned(pad):
    """label 2 (the group loop's head) on a 64-byte boundary plus `pad` s_nops (4 bytes each) in costas_asm_run_ring (the
    shipped header has its own alignment line there: replaced)"""
    def f(s):
        i = s.index("costas_asm_run_ring(")
        s = s[:i] + re.sub(r'"\.p2align 6\\n(\\ts_nop 0\\n)*"\n', '', s[i:], count=1)
        j = s.index('"2:\\n\\t"', i)
        return s[:j] + '".p2align 6\\n\\t' + "s_nop 0\\n\\t" * pad + '"\n        ' + s[j:]
    return f


VARIANTS = {
    "0_as_shipped": lambda s: s,
    **{"a%02d_loop_head_aligned" % k: aligned(k) for k in ((0, 1, 2, 3, 4, 5, 6, 7, 9, 11, 13, 15) if "--align" in sys.argv else (2,))},
    "1_no_lds": no_lds,
    "2_no_wrap_branch": no_branch,
    "3_no_lds_no_branch": lambda s: no_branch(no_lds(s)),
    "4_no_lds_no_branch_no_clamp_zero_wrap_tests": lambda s: no_fillers(no_branch(no_lds(s))),
    "5_also_no_sign_instructions": lambda s: no_sign(no_fillers(no_branch(no_lds(s)))),
}

for name, f in VARIANTS.items():
    hdr = os.path.join(OUT, "costas_asm_%s.h" % name)
    open(hdr, "w").write(f(SRC))
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-I", os.path.join(ROOT, "qpsk_amd", "csrc"),
           '-DCOSTAS_HEADER="%s"' % hdr, os.path.join(ROOT, "tools", "ubench_step.hip"), "-o", os.path.join(OUT, name + ".bin")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    print(name, "ok" if r.returncode == 0 else r.stderr[-2000:])
    if r.returncode:
        sys.exit(1)
