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
