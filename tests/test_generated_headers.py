"""The generated instruction-stream headers are what their generators emit with the documented command lines (the
docstring of tools/gen_fir_asm.py once named a first VGPR for fir_full8_asm.h that would not have fitted timing_scan_kernel's
168 registers; nothing guarded it).  CPU only: runs the generators and compares text."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "qpsk_amd", "csrc")

CASES = [
    (["tools/gen_fir_asm.py", "1", "100", "2"], "fir_r2_asm.h"),
    (["tools/gen_fir_asm.py", "1", "144", "4"], "fir_r4_asm.h"),
    (["tools/gen_fir_asm.py", "1", "80", "8", "1"], "fir_full8_asm.h"),
    (["tools/gen_lean_asm.py"], "fir_lean_asm.h"),
    (["tools/gen_fir_asm.py", "1", "40", "8", "1", "sgpr"], "fir_full8s_asm.h"),
    (["tools/gen_costas_lo.py"], "costas_asm_lo.h"),          # costas_asm.h's streams with the VGPR block 16 registers down (rx_hist_kernel)
]


@pytest.mark.parametrize("cmd,header", CASES, ids=[c[1] for c in CASES])
def test_header_is_what_its_generator_emits(cmd, header):
    out = subprocess.run([sys.executable] + cmd, cwd=ROOT, capture_output=True, text=True, check=True).stdout
    with open(os.path.join(CSRC, header)) as f:
        assert f.read() == out, "%s is stale: regenerate with python %s > qpsk_amd/csrc/%s" % (header, " ".join(cmd), header)


def test_generator_docstring_names_the_committed_command_lines():
    doc = open(os.path.join(ROOT, "tools", "gen_fir_asm.py")).read()
    for cmd, header in [c for c in CASES if c[0][0].endswith('gen_fir_asm.py')]:
        assert "gen_fir_asm.py %s > qpsk_amd/csrc/%s" % (" ".join(cmd[1:]), header) in doc


def _gen_lean():
    import importlib
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    return importlib.import_module("gen_lean_asm")


def test_static_guard_parses_destinations():
    """tools/gen_lean_asm.py parses every emitted instruction for the registers it writes (round 6)"""
    g = _gen_lean()
    w = g.written_and_read
    assert w("v_pk_mul_f32 v[114:115], s[36:37], v[122:123] op_sel_hi:[0,1]") == (["v114", "v115"], ["s36", "s37", "v122", "v123"])
    assert w("ds_read_b128 v[122:125], %[rd] offset:80")[0] == ["v122", "v123", "v124", "v125"]
    assert w("ds_write_b128 %[w0_00], v[64:67]")[0] == [] and w("global_store_short v1, v2, s[18:19] nt")[0] == []
    assert w("global_load_lds_dwordx4 v32, s[10:11] nt")[0] == []                 # writes LDS, not a register
    assert w("s_load_dwordx16 s[36:51], s[26:27], 0x0") == (["s%d" % r for r in range(36, 52)], ["s26", "s27"])
    assert w("v_readfirstlane_b32 s29, v52") == (["s29"], ["v52"])
    assert w("s_add_u32 s10, s10, 0x1000") == (["s10"], ["s10"]) and w("s_cmp_lt_u32 s22, s23")[0] == []
    assert w("v_cmp_gt_u32_e32 vcc, v114, %[wlim]")[0] == [] and w("v_fmac_f64_e32 v[4:5], v[6:7], v[8:9]")[1][-2:] == ["v4", "v5"]
    assert w("s_mov_b64 exec, -1")[0] == [] and w("Lloop_%=:") == ([], [])


def test_static_guard_refuses_a_write_to_a_live_register():
    g = _gen_lean()
    e = g.Emit()
    e.protect(["s26", "s27"], "the taps pointer")
    e.protect(["s10"], "a source pointer", rmw=True)
    e("s_add_u32 s10, s10, 0x1000")              # read-modify-write of an rmw register: fine
    e("v_mov_b32_e32 v1, s26")                   # reading is fine
    with pytest.raises(AssertionError, match="writes s26, which is live: the taps pointer"):
        e("v_readfirstlane_b32 s26, v52")
    with pytest.raises(AssertionError, match="writes s10"):
        e("s_mov_b32 s10, 0")                    # not a read-modify-write
    e.release(["s26", "s27"])
    e("v_readfirstlane_b32 s26, v52")


def test_static_guard_catches_round_5s_clobber():
    """the edit that cost round 5 a GPU memory access fault: the LDS-DMA stream's first-unit loads -- issued between the read of the
    taps pointer into s26:27 and the four scalar loads through it -- took s26 (ST0) as their temporary.  With that edit the generator
    must fail, and as shipped (ST3) it must not."""
    g = _gen_lean()
    try:
        g.DMA = True
        g.block(2)                               # as shipped
        st3 = g.ST3
        g.ST3 = g.ST0
        with pytest.raises(AssertionError, match="which is live: the taps pointer"):
            g.block(2)
    finally:
        g.ST3 = st3
        g.DMA = False
