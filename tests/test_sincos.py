"""sinf/cosf: the oracle's restatement (oracle/oracle_sincosf.h) and the routine the GPU kernels run
(qpsk_amd/csrc/sincos_f32.h, compiled here for the host) against this machine's libm, which is where
the reference gets its values (qpsk.h:35-36).  The exhaustive sweeps are tools/check_sincosf.c (EVERY float
bit pattern: 0 mismatches, 11 s on 8 cores) and tools/check_device_sincos.cpp (every float in [-120, 120],
see DESIGN.md); this test runs the oracle's checker over the whole large-argument domain [120, inf] -- the
part rrc_make() reaches at low samples per symbol (rrc_fir.c:46-49,62-64) and round 4's oracle lacked --
and both programs over the smallest arguments and a dense random sample so that the CPU suite stays short."""
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


def test_oracle_sincosf_equals_libm_every_large_argument(tmp_path):
    """glibc's reduce_large path, |x| >= 120 up to and including the infinities: 2 x 1,016,070,145 arguments"""
    r = _run("check_sincosf.c", ["-DORACLE_SC_FMA=1"], str(tmp_path / "chk"), ["120", "inf"])
    assert r.returncode == 0, r.stdout + r.stderr
    assert "checked=2032140290 " in r.stdout, r.stdout


def test_two_over_pi_table_digits():
    """the 24 words of oracle_sincosf.h's osc_inv_pio4[] are 2/pi's bits through a 32-bit window sliding a byte at a time;
    2/pi recomputed here with integer arithmetic (Machin's formula)"""
    import re

    def atan_inv(x, bits):
        total, term, n, sign = 0, (1 << bits) // x, 1, 1
        while term:
            total += sign * (term // n)
            term //= x * x
            n += 2
            sign = -sign
        return total

    B = 400
    pi = 4 * (4 * atan_inv(5, B) - atan_inv(239, B))
    frac = ((2 << (2 * B)) // pi) >> (B - 192)          # the first 192 bits of 2/pi
    want = [(frac >> (192 - 8 * (k + 1))) & 0xffffffff for k in range(24)]      # word k = floor(2/pi * 2^(8 (k + 1))) mod 2^32
    src = open(os.path.join(ROOT, "oracle", "oracle_sincosf.h")).read()
    body = src[src.index("osc_inv_pio4[24] = {"):]
    body = body[:body.index("};")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    got = [int(w, 16) for w in re.findall(r"0x[0-9a-fA-F]{8}", body)]
    assert got == want


def test_device_form_equals_libm_small_range(tmp_path):
    r = _run("check_device_sincos.cpp", [], str(tmp_path / "chkdev"), ["1e-30"])
    assert r.returncode == 0, r.stdout + r.stderr


def test_stream_form_equals_libm_every_float_of_the_costas_domain(tmp_path):
    """the form the serial wave's instruction stream evaluates since round 6 (costas_asm.h: n = rint of the ROUNDED product x 2/pi,
    v_mul_f64 + v_rndne_f64, reduced argument by v_fmac_f64 in place, the two Horner chains as one chain of per-lane coefficients;
    restated stage by stage as sincos_raw_stream() in sincos_f32.h): every float with |x| <= 8 -- the loop's phase is wrapped to
    [-2 pi, 2 pi] (costas_loop.c:61-67) -- against this machine's libm AND against sincos_raw_horner(), the form the FIR waves' flush
    evaluates from the recorded phase (raw sine, raw cosine, quadrant): 2 x 1,090,519,041 arguments, 0 differences (25 s on 8 cores)"""
    r = _run("check_device_sincos.cpp", [], str(tmp_path / "chkdev"), ["--stream", "8"])
    assert r.returncode == 0, r.stdout + r.stderr
    assert "checked=2181038081 mismatches=0" in r.stdout, r.stdout


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
