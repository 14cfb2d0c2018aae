"""The reference's OWN caller against the drop-in boundary: /root/reference/qpsk.c -- untouched, or with exactly
the edits INTEGRATION.md prints -- compiles as C11 and links against libqpsk_hip.so in place of rrc_fir.c and costas_loop.c.

Build container only (marker `ref`: needs /root/reference; skipped on the GPU box).  LINK ONLY: nothing built here from the
reference travels or runs anywhere; the patched copies live in pytest's tmp_path and are never part of the repo.  The edits are
made by line number, the way INTEGRATION.md states them, and each edited line is first checked to be the line the patch means."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

pytestmark = [pytest.mark.ref,
              pytest.mark.skipif(not os.path.exists(os.path.join(REF, "qpsk.c")), reason="no /root/reference here (GPU box)")]

# what the reference's program needs from rrc_fir.c / costas_loop.c (qpsk.c:125,197-208,217,243,302,308)
PRIMITIVES = ["rrc_fir", "rrc_make", "create_control_loop", "phase_detector", "advance_loop", "phase_wrap", "frequency_limit",
              "get_phase", "get_frequency"]


def link(src, out, includes, defines=()):
    import qpsk_amd
    libdir = os.path.dirname(qpsk_amd.lib_path())
    cmd = ["gcc", "-std=c11", "-DTEST_SCATTER", "-Wall"] + ["-D" + d for d in defines] + ["-I" + i for i in includes] + [
        str(src), "-L" + libdir, "-lqpsk_hip", "-lm", "-Wl,-rpath," + libdir, "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, " ".join(cmd) + "\n" + r.stderr
    return r.stderr


def undefined_symbols(exe):
    out = subprocess.check_output(["nm", "-D", "--undefined-only", str(exe)], text=True)
    return {ln.split()[-1].split("@")[0] for ln in out.splitlines() if ln.strip()}


def ref_lines():
    return open(os.path.join(REF, "qpsk.c")).read().split("\n")


def test_untouched_caller_links_with_the_reference_headers(qpsk_lib, tmp_path):
    """Makefile:7 with `costas_loop.c rrc_fir.c` replaced by the library: the reference's headers, the reference's qpsk.c"""
    exe = tmp_path / "qpsk_ref_headers"
    warn = link(os.path.join(REF, "qpsk.c"), exe, [REF])
    und = undefined_symbols(exe)
    for name in PRIMITIVES:
        assert name in und, name                      # bound to the library, not to a definition of its own
    assert "rx_frame" not in und                      # file-static in the reference (qpsk.c:25)
    assert warn.count("warning") <= 2, warn           # the file-scope VLA of costas_frame[]


def test_patch_a_primitives_from_the_drop_in_header(qpsk_lib, tmp_path):
    """INTEGRATION.md patch A: lines 19-20 (the two #includes) become the drop-in header with QPSK_DROPIN_PRIMITIVES_ONLY; NTAPS
    (qpsk.c:36-37) now comes from qpsk_dropin.h; qpsk.c keeps its own static rx_frame() / qpsk_demod()"""
    ln = ref_lines()
    assert ln[18].strip() == '#include "costas_loop.h"' and ln[19].strip() == '#include "rrc_fir.h"'
    ln[18:20] = ["#define QPSK_DROPIN_PRIMITIVES_ONLY", '#include "qpsk_dropin.h"']
    src = tmp_path / "qpsk_patch_a.c"
    src.write_text("\n".join(ln))
    exe = tmp_path / "qpsk_patch_a"
    # qpsk.h is still the reference's; rrc_fir.h / costas_loop.h must not be reachable: only qpsk.h is copied beside the source
    (tmp_path / "qpsk.h").write_text(open(os.path.join(REF, "qpsk.h")).read())
    link(src, exe, [str(tmp_path), os.path.join(ROOT, "include")])
    und = undefined_symbols(exe)
    for name in PRIMITIVES:
        assert name in und, name
    assert "rx_frame" not in und


def test_patch_b_rx_frame_from_the_library(qpsk_lib, tmp_path):
    """INTEGRATION.md patch B: also drop the static prototypes (24-25) and the definitions of qpsk_demod() (74-79) and rx_frame()
    (88-218): main()'s read loop (qpsk.c:344-354) then calls the library's rx_frame() -- one kernel launch per block"""
    ln = ref_lines()
    assert ln[23].startswith("static void qpsk_demod(") and ln[24].startswith("static void rx_frame(")
    assert ln[73].startswith("static void qpsk_demod(") and ln[78] == "}"
    assert ln[87].startswith("static void rx_frame(") and ln[217] == "}"
    for lo, hi in ((88, 218), (74, 79), (24, 25)):        # 1-based, inclusive; from the bottom up so the numbers stay valid
        del ln[lo - 1:hi]
    assert ln[18].strip() == '#include "costas_loop.h"' and ln[19].strip() == '#include "rrc_fir.h"'
    ln[18:20] = ['#include "qpsk_dropin.h"']
    src = tmp_path / "qpsk_patch_b.c"
    src.write_text("\n".join(ln))
    (tmp_path / "qpsk.h").write_text(open(os.path.join(REF, "qpsk.h")).read())
    exe = tmp_path / "qpsk_patch_b"
    link(src, exe, [str(tmp_path), os.path.join(ROOT, "include")])
    und = undefined_symbols(exe)
    assert "rx_frame" in und and "rrc_fir" in und and "rrc_make" in und and "create_control_loop" in und


def test_header_compiles_as_cxx(qpsk_lib, tmp_path):
    """the reference's headers carry extern "C" guards; the drop-in header is includable from C++ too (pointer-taking functions
    with the compiler's _Complex types), with g++ and with hipcc's clang++"""
    src = tmp_path / "t.cpp"
    src.write_text('#include "qpsk_dropin.h"\n'
                   "int main() { static float _Complex mem[NTAPS], x[8]; static double _Complex a[NFFT], b[NFFT];\n"
                   "  rrc_make(9600.f, 2400.f, .35f); rrc_fir(mem, x, 8); fft(a, b); create_control_loop(.06f, -1.f, 1.f);\n"
                   "  return get_phase() != 0.f || GAIN != 1.85; }\n")
    import qpsk_amd
    libdir = os.path.dirname(qpsk_amd.lib_path())
    for cxx in (["g++", "-std=c++17"], ["/opt/rocm/bin/hipcc", "-x", "c++", "-std=c++17"]):
        if not (os.path.exists(cxx[0]) or cxx[0] == "g++"):
            continue
        cmd = cxx + ["-Wall", "-I", os.path.join(ROOT, "include"), str(src), "-L" + libdir, "-lqpsk_hip", "-Wl,-rpath," + libdir,
                     "-o", str(tmp_path / "tcpp")]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, " ".join(cmd) + "\n" + r.stderr
