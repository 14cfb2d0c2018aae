"""The C-ABI boundary without a GPU: the library builds for gfx950, loads, exports every symbol that
include/*.h declares, and refuses to compute when there is no GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"^\s*#.*$", "", src, flags=re.M)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{}]*\)\s*;", src)
    return sorted(set(n for n in names if n not in ("defined",)))


def test_headers_and_binding_lists_agree():
    from qpsk_amd.lib import API_SYMBOLS, DROPIN_SYMBOLS
    assert declared_functions("qpsk_hip.h") == sorted(API_SYMBOLS)
    assert declared_functions("qpsk_dropin.h") == sorted(DROPIN_SYMBOLS)


def test_library_exports_every_declared_symbol(qpsk_lib):
    for h in ("qpsk_hip.h", "qpsk_dropin.h"):
        for name in declared_functions(h):
            assert hasattr(qpsk_lib, name), "%s declared in include/%s is not exported" % (name, h)


def test_headers_compile_as_c11(tmp_path):
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "qpsk_dropin.h"\nint main(void){ qpsk_params p; qpsk_params_default(&p); return p.frame_size != 512; }\n')
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src), "-o", str(tmp_path / "t.o")])


def build_c_example(out_dir, name="loopback"):
    """examples/<name>.c: a C11 host that uses nothing but include/*.h and the shared library"""
    import subprocess
    import qpsk_amd
    exe = os.path.join(str(out_dir), name)
    libdir = os.path.dirname(qpsk_amd.lib_path())
    subprocess.check_call(["gcc", "-std=c11", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", name + ".c"), "-L", libdir, "-lqpsk_hip", "-lm",
                           "-Wl,-rpath," + libdir, "-o", exe])
    return exe


@pytest.mark.parametrize("name", ["loopback", "dropin_main"])
def test_c_host_example_builds_and_links(qpsk_lib, tmp_path, name):
    assert os.path.exists(build_c_example(tmp_path, name))


@pytest.mark.gpu
def test_c_dropin_main_tracks_the_transmitter_offset(qpsk_lib, tmp_path):
    """the reference's main() loop (qpsk.c:289-359) against include/qpsk_dropin.h, shipped parameters: rx_frame()
    block by block on PCM from a transmitter 50 Hz off centre; the loop's estimate settles there (the oracle fed the
    same way reads 49.9-50.0 Hz after 50 blocks)"""
    import subprocess
    exe = build_c_example(tmp_path, "dropin_main")
    r = subprocess.run([exe, "200"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    hz = float(r.stdout.rsplit("final offset estimate", 1)[1].split()[0])
    assert abs(hz - 50.0) < 1.0, r.stdout


@pytest.mark.gpu
def test_c_host_loopback_runs_on_every_gpu(qpsk_lib, tmp_path):
    """transmitter (N2) -> receive path, driven from plain C on all visible devices: no decision errors"""
    import subprocess
    exe = build_c_example(tmp_path)
    r = subprocess.run([exe, "96", "1024"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "errors 0" in r.stdout


def test_defaults_are_the_reference_literals(qpsk_lib):
    from qpsk_amd.lib import Params
    p = Params()
    qpsk_lib.qpsk_params_default(C.byref(p))
    assert (p.fs, p.rs, p.frame_size) == (9600.0, 2400.0, 512)          # qpsk.h:16-23
    assert abs(p.rrc_alpha - .35) < 1e-7 and p.min_freq == -1.0 and p.max_freq == 1.0   # qpsk.c:302,308
    import numpy as np
    assert np.float32(p.loop_bw) == np.float32(2.0 * 3.14159265358979323846 / 100.0)


def test_no_gpu_means_error_not_fallback(qpsk_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from qpsk_amd.lib import Params
    p = Params()
    qpsk_lib.qpsk_params_default(C.byref(p))
    h = C.c_void_p()
    rc = qpsk_lib.qpsk_ctx_create(C.byref(h), 0, C.byref(p), None)
    assert rc == -1 and not h.value                      # QPSK_ERR_NO_DEVICE
    assert b"no CPU path" in qpsk_lib.qpsk_last_error()
    import qpsk_amd
    with pytest.raises(qpsk_amd.QpskError):
        qpsk_amd.Modem()


def test_bad_arguments_are_rejected(qpsk_lib):
    # argument validation happens before any device work only when a device exists; without one the
    # device error wins -- either way nothing is computed
    from qpsk_amd.lib import Params
    p = Params(9600.0, 2400.0, 510, .35, .06, -1.0, 1.0, 0, 0)  # 510 % 4 != 0
    h = C.c_void_p()
    assert qpsk_lib.qpsk_ctx_create(C.byref(h), 0, C.byref(p), None) < 0
