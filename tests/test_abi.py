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


@pytest.mark.parametrize("name", ["loopback", "dropin_main", "shard_devices"])
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


@pytest.mark.gpu
def test_c_host_shards_a_batch_over_devices_and_gathers(qpsk_lib, tmp_path):
    """examples/shard_devices.c: a C host, one batch split into contiguous shards -- THREE shards on the devices of this box
    (round robin: on the one-GPU box three contexts share the GPU), an odd frame count so that the shards differ in size -- every step's
    symbols / freq / phase gathered into one host array per output, serially and with the copy-back overlapped with the next step's
    kernel: all frames locked, the gathered arrays identical across steps and schedules"""
    import subprocess
    exe = build_c_example(tmp_path, "shard_devices")
    r = subprocess.run([exe, "301", "256", "6", "3"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "not locked (|offset estimate| >= 2 Hz): 0" in r.stdout and "identical across steps and schedules" in r.stdout, r.stdout


@pytest.mark.gpu
def test_multi_job_against_the_oracle(qpsk_lib, oracle):
    """qpsk_multi_*() (include/qpsk_hip.h, MULTI) through ctypes: host frames uploaded shard by shard (two shards on device 0, 3 : 2
    frames... an odd total), two pipelined steps into both result slots, the gathered symbols / freq / phase of EVERY frame against the
    oracle bit for bit; then the error paths of the slot protocol"""
    import numpy as np
    import qpsk_amd
    from oracle.pyoracle import TIMING_FIXED
    from sigutil import bits_equal, make_frames
    fs, rs, L, F = 19200.0, 2400.0, 1024, 77
    mj = qpsk_amd.MultiJob([0, 0], fs=fs, rs=rs, frame_size=L, timing_mode=TIMING_FIXED, fixed_index=6)
    m = qpsk_amd.Modem(fs=fs, rs=rs, frame_size=L, timing_mode=TIMING_FIXED, fixed_index=6)
    x, _ = make_frames(F, L, 8, m.taps, fs, offset_hz=45.0, base_seed=5, noise=0.03)
    want = oracle.rx_batch(x, fs, rs, timing_mode=TIMING_FIXED, fixed_index=6)
    mj.load(x)
    a, b = mj.shard(0), mj.shard(1)
    assert (a["first"], a["count"], b["first"], b["count"]) == (0, 38, 38, 39)      # [r F / N, (r + 1) F / N): qpsk_amd/shard.py's rule
    out0, out1 = mj.outputs(), mj.outputs()
    mj.begin(0)
    mj.begin(1)
    mj.end(0, *out0)
    mj.end(1, *out1)
    for sym, freq, phase in (out0, out1):
        assert np.array_equal(sym, want["sym"]) and bits_equal(freq, want["freq"]) and bits_equal(phase, want["phase"])
    # packed mode: four symbols per byte come back (16 MiB -> 4 MiB per 8192-frame step); unpacked on the host they are the oracle's
    mj.set_packed(True)
    pk = mj.outputs()
    assert pk[0].shape == (F, (m.nsym + 3) // 4)
    mj.begin(1)
    mj.end(1, *pk)
    assert np.array_equal(mj.unpack(pk[0]), want["sym"]) and bits_equal(pk[1], want["freq"])
    ppin = mj.pinned_outputs()
    mj.set_direct(0, *ppin)
    mj.begin(0)
    mj.end(0)
    assert np.array_equal(ppin[0], pk[0]) and bits_equal(ppin[2], want["phase"])
    mj.set_direct(0)
    mj.set_packed(False)
    with pytest.raises(qpsk_amd.QpskError, match="nothing in flight"):
        mj.end(0)
    mj.begin(0)
    with pytest.raises(qpsk_amd.QpskError, match="still in flight"):
        mj.begin(0)
    import torch
    lent = torch.from_numpy(x[a["count"]:].view(np.float32).reshape(b["count"], L, 2).copy()).cuda()
    with pytest.raises(qpsk_amd.QpskError, match="in flight"):      # an input buffer is not swapped under a running step
        mj.use_device_input(1, lent)
    mj.end(0, *out0)
    assert np.array_equal(out0[0], want["sym"])
    # the caller's own device buffer as shard 1's input (no upload of that shard's frames by the job): same results
    mj.use_device_input(1, lent)
    mj.begin(1)
    mj.end(1, *out1)
    assert np.array_equal(out1[0], want["sym"]) and bits_equal(out1[1], want["freq"])
    mj.close()
    # a job that was never loaded has nothing to run
    mj2 = qpsk_amd.MultiJob([0], fs=fs, rs=rs, frame_size=L, timing_mode=TIMING_FIXED, fixed_index=6)
    with pytest.raises(qpsk_amd.QpskError, match="load a job first"):
        mj2.begin(0)
    mj2.close()
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("rows,nsym", [(7, 2048), (33, 37), (5, 16), (3, 1), (1000, 130)])
def test_pack_symbols(qpsk_lib, rows, nsym):
    """qpsk_pack_symbols: four 2-bit symbols per byte, rows of ceil(nsym / 4) bytes (16 symbols per thread where nsym % 16 == 0, byte by byte
    otherwise, the last byte of a row padded with zeros), and qpsk_unpack_symbols_host back"""
    import numpy as np
    import torch
    import qpsk_amd
    m = qpsk_amd.Modem(fs=19200.0, rs=2400.0, frame_size=1024)
    rng = np.random.default_rng(rows * 1000 + nsym)
    sym = rng.integers(0, 4, (rows, nsym), dtype=np.uint8)
    d = torch.from_numpy(sym).cuda()
    pb = (nsym + 3) // 4
    out = torch.full((rows * pb + 64,), 0xEE, dtype=torch.uint8, device="cuda")
    m._check(m.L.qpsk_pack_symbols(m.h, C.c_void_p(d.data_ptr()), rows, nsym, C.c_void_p(out.data_ptr())))
    m.sync()
    got = out.cpu().numpy()
    assert np.all(got[rows * pb:] == 0xEE)
    pad = np.zeros((rows, 4 * pb), np.uint8)
    pad[:, :nsym] = sym
    want = (pad[:, 0::4] | (pad[:, 1::4] << 2) | (pad[:, 2::4] << 4) | (pad[:, 3::4] << 6)).astype(np.uint8)
    assert np.array_equal(got[:rows * pb].reshape(rows, pb), want)
    back = np.empty((rows, nsym), np.uint8)
    m._check(m.L.qpsk_unpack_symbols_host(want.ctypes.data_as(C.c_void_p), rows, nsym, back.ctypes.data_as(C.c_void_p)))
    assert np.array_equal(back, sym)
    m.close()


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
    # the multi-device host layer (round 6) is no way round that: no device, no job
    devs = (C.c_int32 * 2)(0, 0)
    mj = C.c_void_p()
    assert qpsk_lib.qpsk_multi_create(C.byref(mj), devs, 2, C.byref(p)) < 0 and not mj.value
    assert qpsk_lib.qpsk_multi_create(C.byref(mj), devs, 0, C.byref(p)) == -2            # QPSK_ERR_ARG: no shard
    assert qpsk_lib.qpsk_multi_rx_begin(None, 0) == -2 and qpsk_lib.qpsk_multi_shards(None) == 0
    assert qpsk_lib.qpsk_unpack_symbols_host(None, 1, 4, None) == -2
    pk = (C.c_uint8 * 2)(0b11100100, 0b00000010)                                        # symbols 0 1 2 3 | 2
    out = (C.c_uint8 * 5)()
    assert qpsk_lib.qpsk_unpack_symbols_host(pk, 1, 5, out) == 0 and list(out) == [0, 1, 2, 3, 2]      # plain host code: runs anywhere


def test_bad_arguments_are_rejected(qpsk_lib):
    # argument validation happens before any device work only when a device exists; without one the
    # device error wins -- either way nothing is computed
    from qpsk_amd.lib import Params
    p = Params(9600.0, 2400.0, 510, .35, .06, -1.0, 1.0, 0, 0)  # 510 % 4 != 0
    h = C.c_void_p()
    assert qpsk_lib.qpsk_ctx_create(C.byref(h), 0, C.byref(p), None) < 0


@pytest.mark.gpu
def test_dropin_rx_frame_reports_a_flagged_block(qpsk_lib, tmp_path):
    """a kernel that flags its results (here: a loop phase beyond the bounded 2 pi wrap, provoked through the reference's
    own control surface, set_alpha(1e12f)) must not be swallowed by the void drop-in rx_frame(): the process aborts with
    the library's message on stderr instead of returning stale symbols (the reference itself would hang in phase_wrap())"""
    import subprocess
    src = tmp_path / "bad.c"
    src.write_text(r'''
#include <stdio.h>
#include "qpsk_dropin.h"
int main(void)
{
    static int16_t pcm[512];
    for (int i = 0; i < 512; i++) pcm[i] = (int16_t)((i * 7919) % 20000 - 10000);
    create_control_loop(0.0628318f, -1.0f, 1.0f);
    rrc_make(9600.0f, 2400.0f, .35f);
    rx_frame(pcm);                       /* ordinary block: fine */
    rx_frame(pcm);
    printf("two ordinary blocks done\n");
    fflush(stdout);
    set_alpha(1e12f);                    /* costas_loop.h control surface */
    rx_frame(pcm);
    rx_frame(pcm);
    printf("NOT REACHED\n");
    return 0;
}
''')
    import qpsk_amd
    libdir = os.path.dirname(qpsk_amd.lib_path())
    exe = str(tmp_path / "bad")
    subprocess.check_call(["gcc", "-std=c11", "-O2", "-I", os.path.join(ROOT, "include"), str(src), "-L", libdir, "-lqpsk_hip", "-lm",
                           "-Wl,-rpath," + libdir, "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert "two ordinary blocks done" in r.stdout and "NOT REACHED" not in r.stdout
    assert r.returncode != 0 and "2 pi wrap" in r.stderr, r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("block", [1, 0, 2])
def test_streams_rx_pcm_host_equals_the_device_pointer_calls(oracle, block):
    """qpsk_streams_rx_pcm_host (one upload, one download, one synchronisation per block: what the drop-in rx_frame()
    runs on) gives the bits of the oracle's modem block after block, loop state in and out included"""
    import numpy as np
    import qpsk_amd
    from sigutil import bits_equal
    fs, rs, L, S, B = 9600.0, 2400.0, 512, 3, 6
    m = qpsk_amd.Modem(fs=fs, rs=rs, frame_size=L)
    m.tune(stream_block=1 if block == 1 else 0)      # 1: one launch per block on the pinned staging buffer; 0: copies + the composition
    m.tune(stream_scan=1 if block == 2 else 0)       # 2: mixer + filter + scan as one kernel (needs whole 256-sample tiles at CYCLES = 8)
    m.streams_reset(S, 1500.0)
    rng = np.random.default_rng(3)
    om = [oracle.modem(fs, rs, L, loop_bw=np.float32(2.0 * 3.14159265358979323846 / 100.0)) for _ in range(S)]
    for o in om:
        o.set_mixer_hz(1500.0)
    N = m.nsym
    for k in range(B):
        pcm = (6000 * rng.standard_normal((S, L))).astype(np.int16)
        st = np.array([[o.phase, o.freq] for o in om], np.float32)
        if k == 3:                         # the caller edits the state between blocks (set_phase()/set_frequency())
            st[:, 0] += np.float32(0.25)
            for i, o in enumerate(om):
                o.s.loop.phase = st[i, 0]
        sym = np.zeros((S, N), np.uint8); cos = np.zeros((S, N, 2), np.float32); idx = np.zeros(S, np.int32)
        rc = m.L.qpsk_streams_rx_pcm_host(m.h, C.c_void_p(pcm.ctypes.data), C.c_void_p(st.ctypes.data), C.c_void_p(sym.ctypes.data),
                                          C.c_void_p(cos.ctypes.data), C.c_void_p(idx.ctypes.data))
        assert rc == 0, m.L.qpsk_last_error()
        for i, o in enumerate(om):
            o.rx_pcm(pcm[i])
            assert idx[i] == o.index and bits_equal(sym[i], o.symbols), (k, i)
            bad = np.nonzero(cos[i].view(np.uint32) != o.costas_frame.view(np.uint32))[0]
            assert bad.size == 0, "block %d stream %d: costas_frame differs at symbols %s: %s vs %s" % (
                k, i, bad[:8], cos[i][bad[:4]], o.costas_frame[bad[:4]])
            assert st[i, 0] == o.phase and st[i, 1] == o.freq, (k, i)


@pytest.mark.gpu
@pytest.mark.parametrize("fs,L,S", [(9600.0, 512, 1500), (19200.0, 512, 3000)])
def test_streams_rx_pcm_host_many_streams(oracle, fs, L, S):
    """the host-buffer call with the library's own kernel choice at stream counts beyond the handful the drop-in uses: 1500 streams at
    the shipped configuration (one launch per block reading the pinned staging buffer) and 3000 at CYCLES = 8 (staged copy, then
    stream_scan_kernel + the loop kernel); a sample of streams against the oracle's modems, block after block"""
    import numpy as np
    import qpsk_amd
    from sigutil import bits_equal
    rs, B = 2400.0, 3
    m = qpsk_amd.Modem(fs=fs, rs=rs, frame_size=L)
    m.streams_reset(S, 1500.0)
    rng = np.random.default_rng(S)
    check = [0, 7, S // 2, S - 1]
    om = {i: oracle.modem(fs, rs, L, loop_bw=np.float32(2.0 * 3.14159265358979323846 / 100.0)) for i in check}
    for o in om.values():
        o.set_mixer_hz(1500.0)
    N = m.nsym
    for k in range(B):
        pcm = (6000 * rng.standard_normal((S, L))).astype(np.int16)
        sym = np.zeros((S, N), np.uint8); cos = np.zeros((S, N, 2), np.float32); idx = np.zeros(S, np.int32)
        rc = m.L.qpsk_streams_rx_pcm_host(m.h, C.c_void_p(pcm.ctypes.data), None, C.c_void_p(sym.ctypes.data),
                                          C.c_void_p(cos.ctypes.data), C.c_void_p(idx.ctypes.data))
        assert rc == 0, m.L.qpsk_last_error()
        for i, o in om.items():
            o.rx_pcm(pcm[i])
            assert idx[i] == o.index and bits_equal(sym[i], o.symbols) and bits_equal(cos[i], o.costas_frame), (k, i)
    assert m.last_kernel() == ("stream_block_kernel" if fs == 9600.0 else "stream_scan_kernel + costas_pipe_kernel")
