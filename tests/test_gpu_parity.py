"""Parity of the HIP path (through the C ABI, include/qpsk_hip.h) with the CPU oracle on the same
inputs.  Bar: symbols, costas_frame, loop phase/frequency and every intermediate BIT-EXACT
(the 1e-5 relative tolerance BASELINE.json allows for phase/frequency is not used: 0 is achieved)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import golden
from oracle.pyoracle import TAU, TIMING_FIXED, TIMING_HIST
from sigutil import bits_equal, make_frames, random_frames

pytestmark = pytest.mark.gpu
BW = np.float32(TAU / 100.0)


def modem(**kw):
    import qpsk_amd
    return qpsk_amd.Modem(**kw)


def cpu(t):
    return t.cpu().numpy()


def assert_batch_equal(got, want, keys=("sym", "phase", "freq", "index", "hz", "costas")):
    for k in keys:
        if k in want and got.get(k) is not None:
            g = cpu(got[k])
            assert bits_equal(g, want[k].astype(g.dtype)), "%s differs (%d mismatching elements)" % (
                k, int(np.sum(g.reshape(-1).view(np.uint8) != want[k].astype(g.dtype).reshape(-1).view(np.uint8))))


# ------------------------------------------------------------------ sin/cos on the device
def test_device_sincos_equals_oracle_exhaustive_costas_domain(oracle):
    """every float in [-2pi, 2pi] (2.17e9 arguments) through the device routine, hashed, against the
    same hash of the oracle's restatement (itself pinned to libm exhaustively)"""
    import subprocess, tempfile
    m = modem()
    top = np.float32(6.2831855).view(np.uint32)
    dev = m.sincos_hash(0, int(top) + 1)
    src = r'''
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include "oracle_sincosf.h"
int main(void){ uint32_t top=%uu; unsigned long long h=0;
#pragma omp parallel for reduction(+:h) schedule(static,65536)
 for(uint32_t u=0;u<=top;u++) for(int sg=0;sg<2;sg++){ uint32_t b=u|((uint32_t)sg<<31); float y,s,c; memcpy(&y,&b,4);
  oracle_sincosf(y,&s,&c); uint32_t sb,cb; memcpy(&sb,&s,4); memcpy(&cb,&c,4);
  unsigned long long v=((unsigned long long)sb<<32)|cb; v^=(unsigned long long)b*0x9E3779B97F4A7C15ull; v*=0xD6E8FEB86659FD93ull; v^=v>>32; h+=v; }
 printf("%%llu\n",h); return 0; }''' % int(top)
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "h.c"), "w").write(src)
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-I", os.path.join(root, "oracle"),
                               os.path.join(d, "h.c"), "-o", os.path.join(d, "h"), "-lm"])
        host = int(subprocess.check_output([os.path.join(d, "h")]).decode())
    assert dev == host


# ------------------------------------------------------------------ configuration numbers
def test_taps_and_gains_equal_oracle(oracle):
    for fs, rs, a in [(9600.0, 2400.0, .35), (19200.0, 2400.0, .35), (9600.0, 1200.0, .5)]:
        m = modem(fs=fs, rs=rs, frame_size=int(fs / rs) * 64, rrc_alpha=a)
        assert bits_equal(m.taps, oracle.rrc_make(fs, rs, np.float32(a)))
        from oracle.pyoracle import Costas
        c = Costas()
        oracle.lib.qo_costas_create(C.byref(c), BW, -1.0, 1.0)
        al, be = m.gains
        assert al == np.float32(c.alpha) and be == np.float32(c.beta)


# ------------------------------------------------------------------ the hot path
@pytest.mark.parametrize("fs,rs,L,F", [(19200.0, 2400.0, 1024, 37), (19200.0, 2400.0, 16384, 24), (9600.0, 2400.0, 512, 70),
                                       (9600.0, 1200.0, 4096, 9), (19200.0, 2400.0, 8, 3), (19200.0, 2400.0, 136, 5)])
@pytest.mark.parametrize("index", [0, 3, 7])
def test_rx_batch_fixed_index(oracle, fs, rs, L, F, index):
    m = modem(fs=fs, rs=rs, frame_size=L, timing_mode=TIMING_FIXED, fixed_index=index)
    x, _ = make_frames(F, L, m.cycles, m.taps, fs, offset_hz=50.0, base_seed=L + index, noise=0.02)
    want = oracle.rx_batch(x, fs, rs, loop_bw=BW, timing_mode=TIMING_FIXED, fixed_index=index, want_costas=True)
    got = m.rx_batch(x, want_costas=True)
    m.sync()
    assert_batch_equal(got, want)


@pytest.mark.parametrize("fs,rs,L,F", [(19200.0, 2400.0, 1024, 40), (19200.0, 2400.0, 16384, 6), (9600.0, 2400.0, 512, 33),
                                       (9600.0, 1200.0, 4096, 5)])
def test_rx_batch_reference_timing(oracle, fs, rs, L, F):
    """the reference's own histogram timing estimate in front of the fused kernel"""
    m = modem(fs=fs, rs=rs, frame_size=L, timing_mode=TIMING_HIST)
    x, _ = make_frames(F, L, m.cycles, m.taps, fs, offset_hz=-35.0, base_seed=L, noise=0.05)
    x[-1] = random_frames(1, L, seed=1)[0]      # not a modem signal
    x[-2] = 0.0                                  # all-zero block -> index 1 (Q4)
    want = oracle.rx_batch(x, fs, rs, loop_bw=BW, timing_mode=TIMING_HIST, want_costas=True)
    assert want["index"][-2] == 1
    got = m.rx_batch(x, want_costas=True)
    m.sync()
    assert_batch_equal(got, want)


@pytest.mark.parametrize("fs,rs,L,F", [(19200.0, 2400.0, 1024, 40), (19200.0, 2400.0, 16384, 6), (9600.0, 2400.0, 2048, 33),
                                       (4800.0, 2400.0, 1024, 21)])
def test_rx_batch_fft_timing(oracle, fs, rs, L, F):
    """config 3: the FFT timing estimate (new design on top of the reference's radix-2 FFT) in front of the
    fused kernel -- parity with the oracle's restatement of the same definition (unpinned by the reference)"""
    from oracle.pyoracle import TIMING_FFT
    m = modem(fs=fs, rs=rs, frame_size=L, timing_mode=TIMING_FFT)
    x, _ = make_frames(F, L, m.cycles, m.taps, fs, offset_hz=20.0, base_seed=L + 1, noise=0.05)
    x[-1] = random_frames(1, L, seed=2)[0]
    x[-2] = 0.0
    want = oracle.rx_batch(x, fs, rs, loop_bw=BW, timing_mode=TIMING_FFT, want_costas=True)
    got = m.rx_batch(x, want_costas=True)
    m.sync()
    assert_batch_equal(got, want)
    # on clean modem frames the estimate is the eye centre: TX + RX group delay 126 samples = 126 mod CYCLES
    assert np.all(want["index"][:F - 2] == 126 % m.cycles)


INLINE = " (FFT timing estimate inside the launch)"


@pytest.mark.parametrize("tune,kernel", [(dict(pipe_g=32), "rx_lean_kernel"), (dict(pipe_g=20), "rx_lean_kernel" + INLINE), (dict(pipe_g=8), "rx_lean_kernel" + INLINE),
                                         (dict(pipe_v=2), "rx_pipe2_kernel"), (dict(pipe_v=3), "rx_lean_kernel" + INLINE),
                                         (dict(pipe_v=1), "rx_fused_pipe_kernel" + INLINE), (dict(pipe_v=1, pipe_nf=2), "rx_fused_pipe_kernel"),
                                         (dict(fused_generic=1), "rx_fused_kernel"), (dict(lean_dma=0), "rx_lean_kernel" + INLINE),
                                         (dict(fft_fused=0), "rx_lean_kernel"), (dict(pipe_v=1, pipe_dbg=128), "rx_fused_pipe_kernel"),
                                         (dict(), "rx_lean_kernel" + INLINE)])
def test_fft_timing_under_every_geometry_key(oracle, tune, kernel):
    """No tuning key may change a result (include/qpsk_hip.h).  Round 4's host code restated the kernel choice to decide whether the
    FFT estimate runs inside the receive launch and missed QPSK_PIPE_G: with G above 16 the batch went to rx_lean_kernel, which has no
    estimate, with no index computed -- silently wrong symbols.  The plan is now made once and the estimate's placement reads it: the
    FFT-timed batch under every key that moves it to another kernel, against the oracle, the index included.  (QPSK_PIPE_G = 32 fills the
    LDS: no room for the estimate's 640 bytes; 20 frames per workgroup leave a last partial workgroup: until round 5 that went to another
    kernel and the estimate ran in front of both, since round 6 the partial workgroup rides in rx_lean_kernel's launch, estimate included.)"""
    from oracle.pyoracle import TIMING_FFT
    fs, rs, L, F = 19200.0, 2400.0, 1024, 4096        # 4096 frames = 16 per CU: the in-launch estimate's shape when nothing is set
    m = modem(fs=fs, rs=rs, frame_size=L, timing_mode=TIMING_FFT, fixed_index=3)     # a fixed_index that is NOT the estimate (6)
    x, _ = make_frames(F, L, m.cycles, m.taps, fs, offset_hz=30.0, base_seed=77, noise=0.03)
    x[5] = random_frames(1, L, seed=9)[0]
    want = oracle.rx_batch(x, fs, rs, loop_bw=BW, timing_mode=TIMING_FFT, fixed_index=3)
    m.tune(**tune)
    got = m.rx_batch(x)
    m.sync()
    assert m.last_kernel() == kernel, m.last_kernel()
    assert_batch_equal(got, want)
    assert np.all(want["index"][:5] == 6)


@pytest.mark.parametrize("F", [1024, 2560, 3584, 4608, 5632, 8192])
def test_fft_timing_inside_rx_lean_kernel_at_every_workgroup_size(oracle, F):
    """the FFT estimate inside rx_lean_kernel's launch at 4, 10, 14, 18 and 22 frames per workgroup (one window per FIR wave, a window per
    unit, eight to twelve hardware waves sharing the estimate) -- and at 32, where the LDS has no room for it and it runs in front:
    per-frame offsets of every value (frames delayed by 0..7 samples), against the oracle"""
    from oracle.pyoracle import TIMING_FFT
    fs, rs, L = 19200.0, 2400.0, 1024
    m = modem(fs=fs, rs=rs, frame_size=L, timing_mode=TIMING_FFT)
    x, _ = make_frames(F, L + 8, 8, m.taps, fs, offset_hz=30.0, base_seed=13, noise=0.05)
    x = np.stack([x[f, (f % 8):(f % 8) + L] for f in range(F)])
    x[3] = random_frames(1, L, seed=6)[0]
    got = m.rx_batch(x)
    m.sync()
    assert m.last_kernel() == ("rx_lean_kernel" if F == 8192 else "rx_lean_kernel" + INLINE), m.last_kernel()
    want = oracle.rx_batch(x, fs, rs, loop_bw=BW, timing_mode=TIMING_FFT, threads=min(16, os.cpu_count() or 1))
    assert len(np.unique(want["index"])) == 8
    assert_batch_equal(got, want, keys=("sym", "phase", "freq", "index", "hz"))


def test_rx_batch_tilings_agree(oracle):
    """results do not depend on how frames are grouped into workgroups / chunks"""
    fs, rs, L, F = 19200.0, 2400.0, 2048, 50
    m = modem(fs=fs, rs=rs, frame_size=L, timing_mode=TIMING_FIXED, fixed_index=4)
    x, _ = make_frames(F, L, 8, m.taps, fs, base_seed=3)
    want = oracle.rx_batch(x, fs, rs, loop_bw=BW, timing_mode=TIMING_FIXED, fixed_index=4, want_costas=True)
    for G, S in [(1, 8), (3, 24), (16, 32), (64, 8), (7, 64)]:
        m.tune(fused_g=G)
        m.tune(fused_s=S)
        got = m.rx_batch(x, want_costas=True)
        m.sync()
        assert_batch_equal(got, want)


@pytest.mark.parametrize("L,mode", [(2048, TIMING_FIXED), (1000, TIMING_FIXED), (2048, TIMING_HIST), (16384, TIMING_FIXED)])
def test_pipeline_geometries_agree(oracle, L, mode):
    """every layout of the pipeline kernels gives the oracle's bits: rx_pipe2_kernel (two-frame units, 1..32 frames
    per workgroup, 1..9 FIR waves owning one or two units each) and the lane mappings of rx_fused_pipe_kernel (4 or 2
    symbols per FIR lane in one 16-frame workgroup); frame counts are ragged against all of them"""
    fs, rs, F = 19200.0, 2400.0, 75
    m = modem(fs=fs, rs=rs, frame_size=L, timing_mode=mode, fixed_index=5)
    x, _ = make_frames(F, L, 8, m.taps, fs, base_seed=L, noise=0.05)
    x[7] = 0.0
    want = oracle.rx_batch(x, fs, rs, loop_bw=BW, timing_mode=mode, fixed_index=5, want_costas=True)
    m.tune(pipe_v=2)
    # frames per workgroup (two-frame units dealt to up to nine FIR waves by the library) ...
    for G in (None, 1, 2, 3, 5, 16, 17, 18, 31, 32):
        m.tune(pipe_g=G)
        got = m.rx_batch(x, want_costas=True)
        m.sync()
        assert_batch_equal(got, want)
        got = m.rx_batch(x[:33], want_costas=False)
        m.sync()
        assert bits_equal(cpu(got["sym"]), want["sym"][:33]) and bits_equal(cpu(got["freq"]), want["freq"][:33])
    # ... and explicit wave layouts (4 bits per hardware wave = units it owns, waves 1-5 / 6-11): one wave with two
    # units; a FIR wave beside the serial wave (hardware wave 4); 32 frames on nine waves with SIMD 0 carrying two units
    for G, lo, hi in ((4, 0x00002, 0), (6, 0x01011, 0), (10, 0x11111, 0), (32, 0x22222, 0x011022), (32, 0x20222, 0x112022), (24, 0x20222, 0x000022)):
        m.tune(pipe_g=G, pipe_layout_lo=lo, pipe_layout_hi=hi)
        got = m.rx_batch(x, want_costas=True)
        m.sync()
        assert_batch_equal(got, want)
    m.tune(pipe_layout_lo=None, pipe_layout_hi=None)
    m.tune(pipe_v=1, pipe_g=None, pipe_nf=None)
    for nf in (1, 2, 3, 4):
        m.tune(pipe_nf=nf)
        got = m.rx_batch(x, want_costas=True)
        m.sync()
        assert_batch_equal(got, want)
        got = m.rx_batch(x[:33], want_costas=False)
        m.sync()
        assert bits_equal(cpu(got["sym"]), want["sym"][:33]) and bits_equal(cpu(got["freq"]), want["freq"][:33])
    # nf = 4 above ran with two lane mappings in one workgroup (16 frames: 12 at 4 symbols per lane, 4 at 2);
    # the A/B variant with one mapping for all four FIR waves
    m.tune(pipe_nf=4)
    m.tune(pipe_dbg=128)
    got = m.rx_batch(x, want_costas=True)
    m.sync()
    assert_batch_equal(got, want)


@pytest.mark.parametrize("L,mode,index", [(2048, TIMING_FIXED, 6), (1024, TIMING_FIXED, 3), (4096, TIMING_FIXED, 7),
                                          (2048, TIMING_HIST, 0), (16384, TIMING_FIXED, 0)])
def test_lean_kernel_equals_oracle(oracle, L, mode, index):
    """rx_lean_kernel (the FIR waves' whole chunk loop as one hand-written stream, taps in SGPRs): the oracle's bits
    for every even workgroup size, for explicit wave layouts (one and two units per wave, a FIR wave beside the serial
    wave), even, odd and per-frame decimation offsets (histogram timing), all-zero frames, and batches whose last
    partial workgroup goes to the older kernels"""
    fs, rs, F = 19200.0, 2400.0, 77
    m = modem(fs=fs, rs=rs, frame_size=L, timing_mode=mode, fixed_index=index)
    x, _ = make_frames(F, L, 8, m.taps, fs, base_seed=L + index, noise=0.05)
    x[7] = 0.0
    x[40, : L // 2] = 0.0
    want = oracle.rx_batch(x, fs, rs, loop_bw=BW, timing_mode=mode, fixed_index=index)
    m.tune(pipe_v=3)
    for G in (2, 4, 6, 10, 16, 18, 30, 32):
        m.tune(pipe_g=G)
        for n in (F, F - F % G, G):
            got = m.rx_batch(x[:n])
            m.sync()
            assert m.last_kernel() == "rx_lean_kernel", (G, n, m.last_kernel())
            for k in ("sym", "phase", "freq", "hz", "index"):
                assert bits_equal(cpu(got[k]), want[k][:n].astype(cpu(got[k]).dtype)), (k, G, n)
    for G, lo, hi in ((4, 0x00002, 0), (6, 0x01011, 0), (10, 0x11111, 0), (32, 0x22222, 0x011022), (32, 0x20222, 0x112022),
                      (24, 0x20222, 0x000022), (32, 0x21222, 0x012022)):
        m.tune(pipe_g=G, pipe_layout_lo=lo, pipe_layout_hi=hi)
        got = m.rx_batch(x)
        m.sync()
        assert m.last_kernel() == "rx_lean_kernel", (G, lo, hi)
        for k in ("sym", "phase", "freq", "hz"):
            assert bits_equal(cpu(got[k]), want[k]), (k, G, hex(lo), hex(hi))
    # shapes it does not serve stay with the other kernels: a costas_frame[] dump, an odd workgroup
    m.tune(pipe_layout_lo=None, pipe_layout_hi=None, pipe_g=32)
    got = m.rx_batch(x, want_costas=True)
    m.sync()
    assert m.last_kernel() != "rx_lean_kernel"
    assert_batch_equal(got, oracle.rx_batch(x, fs, rs, loop_bw=BW, timing_mode=mode, fixed_index=index, want_costas=True))


def test_lean_kernel_wraps_and_loop_variants(oracle):
    """carrier offsets that wrap the phase every few symbols, a wide clamp and an asymmetric one (the serial wave's
    fallbacks) through rx_lean_kernel"""
    fs, rs, L, F = 19200.0, 2400.0, 4096, 64
    for offset_hz, bw, lim in ((270.0, np.float32(0.6), (-1.0, 1.0)), (-250.0, np.float32(0.3), (-7.0, 7.0)),
                               (120.0, BW, (0.0, 1.0))):
        m = modem(fs=fs, rs=rs, frame_size=L, timing_mode=TIMING_FIXED, fixed_index=6, loop_bw=bw, min_freq=lim[0],
                  max_freq=lim[1])
        x, _ = make_frames(F, L, 8, m.taps, fs, offset_hz=offset_hz, base_seed=int(abs(offset_hz)), noise=0.1)
        want = oracle.rx_batch(x, fs, rs, loop_bw=bw, min_freq=lim[0], max_freq=lim[1], timing_mode=TIMING_FIXED, fixed_index=6)
        m.tune(pipe_v=3, pipe_g=32)
        got = m.rx_batch(x)
        m.sync()
        assert m.last_kernel() == "rx_lean_kernel"
        for k in ("sym", "phase", "freq", "hz"):
            assert bits_equal(cpu(got[k]), want[k]), (k, offset_hz)


def test_rx_batch_golden_vectors():
    """straight against the reference's own outputs (tests/golden, generated from the reference)"""
    for name in ("c1small", "c1", "c5small_bw200"):
        g = golden("independent_%s.npz" % name)
        m = modem(fs=float(g["fs"]), rs=float(g["rs"]), frame_size=int(g["frame_size"]),
                  loop_bw=np.float32(g["loop_bw"]), timing_mode=TIMING_HIST)
        got = m.rx_batch(g["x"], want_costas=True)
        m.sync()
        assert_batch_equal(got, {k: g[k] for k in ("sym", "costas", "phase", "freq", "hz", "index")})


def test_extreme_inputs(oracle):
    """large amplitudes (several phase wraps per step), tiny amplitudes (denormal products), clamp active"""
    fs, rs, L = 19200.0, 2400.0, 1024
    for scale, lo, hi, bw in [(50.0, -1.0, 1.0, BW), (1e-20, -1.0, 1.0, BW), (1.0, -0.01, 0.005, BW),
                              (3.0, -6.0, 6.0, np.float32(0.9)), (1e-38, -1.0, 1.0, BW)]:
        m = modem(fs=fs, rs=rs, frame_size=L, loop_bw=bw, min_freq=lo, max_freq=hi, timing_mode=TIMING_FIXED, fixed_index=2)
        x = random_frames(20, L, seed=int(abs(np.log10(scale)) + 1), scale=scale)
        want = oracle.rx_batch(x, fs, rs, loop_bw=bw, min_freq=lo, max_freq=hi, timing_mode=TIMING_FIXED, fixed_index=2, want_costas=True)
        got = m.rx_batch(x, want_costas=True)
        m.sync()
        assert_batch_equal(got, want)


@pytest.mark.parametrize("offset_hz,bw", [(200.0, BW), (-250.0, np.float32(0.3)), (270.0, np.float32(0.6))])
def test_carrier_offsets_wrap_often(oracle, offset_hz, bw):
    """loops that settle at 0.5 .. 1 rad/symbol, of either sign and up to the clamp (costas_loop.c:69-74): the phase
    runs through 2 pi every 6 to 12 symbols, so the out-of-line wrap (costas_loop.c:61-67) of every step position of
    the hand-scheduled stream is taken many times per frame"""
    fs, rs, L, F = 19200.0, 2400.0, 8192, 24
    m = modem(fs=fs, rs=rs, frame_size=L, loop_bw=bw, timing_mode=TIMING_FIXED, fixed_index=6)
    x, _ = make_frames(F, L, 8, m.taps, fs, offset_hz=offset_hz, base_seed=int(abs(offset_hz)), noise=0.1)
    want = oracle.rx_batch(x, fs, rs, loop_bw=bw, timing_mode=TIMING_FIXED, fixed_index=6, want_costas=True)
    got = m.rx_batch(x, want_costas=True)
    m.sync()
    assert_batch_equal(got, want)
    assert np.abs(want["freq"]).max() > 0.5


def test_loop_bandwidth_sweep(oracle):
    """README.md:12: loop bandwidth TAU/100 .. TAU/200, several loops sharing one FIR pass (config 5)"""
    fs, rs, L, F = 9600.0, 1200.0, 8192, 7
    bws = [np.float32(TAU / d) for d in range(100, 201, 10)]
    m = modem(fs=fs, rs=rs, frame_size=L, timing_mode=TIMING_FIXED, fixed_index=4)
    x, _ = make_frames(F, L, 8, m.taps, fs, offset_hz=12.0, base_seed=9, noise=0.03)
    want = oracle.rx_batch_bw(x, fs, rs, bws, timing_mode=TIMING_FIXED, fixed_index=4)
    got = m.rx_batch_bw(x, bws)
    m.sync()
    assert_batch_equal(got, want, keys=("sym", "phase", "freq"))
    # and the single-loop entry still uses the context's own gains afterwards
    one = m.rx_batch(x)
    m.sync()
    assert bits_equal(cpu(one["sym"]), want["sym"][:, 0]) and bits_equal(cpu(one["freq"]), want["freq"][:, 0])


def test_full_size_config2_properties(oracle):
    """BASELINE config 2 at full size (4096 frames x 16384 samples, 512 MiB): the oracle cannot run all of it
    in seconds, so: (a) a spread sample of frames is compared with the oracle bit for bit, (b) a second launch
    reproduces every output bit, (c) frames are independent: reversing the batch reverses the outputs,
    (d) every frame's loop ends on the +50 Hz offset the generator applied (qpsk.c:320 vs 342)."""
    import torch
    import bench
    fs, rs, L, F = bench.FS, bench.RS, 16384, 4096
    m = modem(fs=fs, rs=rs, frame_size=L, timing_mode=TIMING_FIXED, fixed_index=bench.FIXED_INDEX)
    x = bench.synth_frames_gpu(torch, torch.device("cuda", 0), F, m.taps, seed=7)
    a = m.rx_batch(x)
    m.sync()
    pick = np.unique(np.concatenate([np.arange(0, F, 97), [1, 15, 16, 17, F - 17, F - 16, F - 1]]))
    want = oracle.rx_batch(x[torch.from_numpy(pick).cuda()].cpu().numpy(), fs, rs, loop_bw=BW, timing_mode=TIMING_FIXED,
                           fixed_index=bench.FIXED_INDEX)
    for k in ("sym", "phase", "freq", "hz"):
        assert bits_equal(cpu(a[k])[pick], want[k]), k
    b = m.rx_batch(x)
    m.sync()
    for k in ("sym", "phase", "freq"):
        assert bits_equal(cpu(a[k]), cpu(b[k])), k
    r = m.rx_batch(torch.flip(x, dims=[0]).contiguous())
    m.sync()
    for k in ("sym", "phase", "freq"):
        assert bits_equal(cpu(a[k]), cpu(r[k])[::-1].copy()), k
    assert np.all(np.abs(cpu(a["hz"]) - 50.0) < 2.0)


@pytest.mark.parametrize("mode", ["fixed_even", "fixed_odd", "per_frame"])
def test_lean_kernel_window_staging_by_dma_and_through_registers(oracle, mode):
    """rx_lean_kernel stages a FIR wave's windows by LDS-DMA when every decimation offset of the wave's four frames is even (16-byte pairs
    of the window image are 16-byte pairs of the input) and through registers otherwise -- a per-wave choice, so one launch mixes both
    when the offsets come per frame (histogram timing: random frames get every index).  Every case against the oracle, and DMA off
    (QPSK_LEAN_DMA = 0) equal to DMA on."""
    from oracle.pyoracle import TIMING_HIST as TH
    fs, rs, L, F = 19200.0, 2400.0, 1536, 8192 + 64
    if mode == "per_frame":
        m = modem(fs=fs, rs=rs, frame_size=L, timing_mode=TH)
        # noise, clean modem frames, tones and decaying bursts: their amplitude histograms put the reference's index (qpsk.c:173-180)
        # on every value 0..7, so FIR waves with four even offsets (DMA) and waves with an odd one (registers) sit side by side
        x = random_frames(F, L, seed=31)
        clean, _ = make_frames(F // 2, L, 8, m.taps, fs, offset_hz=25.0, base_seed=8, noise=0.05)
        x[::2] = clean
        rng, n = np.random.default_rng(5), np.arange(L)
        for f in range(1, F, 4):
            z = np.exp(1j * rng.uniform(0.01, 0.5) * n) * rng.uniform(0.2, 3)
            x[f, :, 0], x[f, :, 1] = z.real, z.imag
        for f in range(3, F, 8):
            z = np.exp(-n / rng.uniform(50, 800)) * (rng.standard_normal(L) + 1j * rng.standard_normal(L))
            x[f, :, 0], x[f, :, 1] = z.real, z.imag
        want = oracle.rx_batch(x, fs, rs, loop_bw=BW, timing_mode=TH, threads=min(16, os.cpu_count() or 1))
        assert len(np.unique(want["index"])) >= 4 and np.any(want["index"] % 2 == 1) and np.any(want["index"] % 2 == 0)
    else:
        ix = 4 if mode == "fixed_even" else 3
        m = modem(fs=fs, rs=rs, frame_size=L, timing_mode=TIMING_FIXED, fixed_index=ix)
        x, _ = make_frames(F, L, 8, m.taps, fs, offset_hz=25.0, base_seed=9, noise=0.05)
        want = oracle.rx_batch(x, fs, rs, loop_bw=BW, timing_mode=TIMING_FIXED, fixed_index=ix, threads=min(16, os.cpu_count() or 1))
    import torch
    xd = torch.from_numpy(x).cuda()
    m.tune(hist_onepass=0)      # this test is about rx_lean_kernel's staging (the one-pass histogram route has its own test)
    for dma in (None, 0):
        m.tune(lean_dma=dma)
        got = m.rx_batch(xd)
        m.sync()
        assert m.last_kernel() == "rx_lean_kernel", m.last_kernel()
        assert_batch_equal(got, want, keys=("sym", "phase", "freq", "index", "hz"))


@pytest.mark.parametrize("F,pipe_g", [(4096, None), (8192, None), (1024, None), (3072, None), (1280, 10)])
def test_lean_kernel_paired_serial_lanes(oracle, F, pipe_g):
    """rx_lean_kernel's serial wave with ONE lane per loop and with TWO (round 6: lanes 2f, 2f + 1 share the step's sine / cosine polynomial
    chains, costas_asm.h QPSK_BODY_P; QPSK_LEAN_PAIR 0 / 1 = up to 16 frames per workgroup / 2 = up to 32): the same bits, at 4-32 frames
    per workgroup, on frames that send the stream through every one of its exits -- clean modem frames (a 2 pi wrap every 48 steps or
    so), noise (wraps at random), silent frames and silent stretches inside frames (the zero test trips, lanes are excused), real-only
    frames (T.x or T.y exactly zero whenever the phase is: the C++ step takes the group), tiny and large levels."""
    import torch
    fs, rs, L = 19200.0, 2400.0, 1536
    m = modem(fs=fs, rs=rs, frame_size=L, timing_mode=TIMING_FIXED, fixed_index=6)
    x, _ = make_frames(F, L, 8, m.taps, fs, offset_hz=50.0, base_seed=61, noise=0.03)
    x[1::16] = random_frames(len(x[1::16]), L, seed=62)
    x[5::64] = 0.0
    x[7::32, 300:900] = 0.0
    x[9::32, :, 1] = 0.0
    x[11::64] *= 1e-18
    x[13::64] *= 300.0
    want = oracle.rx_batch(x, fs, rs, loop_bw=BW, timing_mode=TIMING_FIXED, fixed_index=6, threads=min(16, os.cpu_count() or 1))
    xd = torch.from_numpy(x).cuda()
    m.tune(pipe_g=pipe_g)
    for pair in (0, 1, 2, None):
        m.tune(lean_pair=pair)
        got = m.rx_batch(xd)
        m.sync()
        assert m.last_kernel() == "rx_lean_kernel", m.last_kernel()
        assert_batch_equal(got, want, keys=("sym", "phase", "freq", "index", "hz"))


@pytest.mark.parametrize("F,mode", [(4097, "fixed"), (5001, "fixed"), (8191, "fixed"), (8193, "fixed"), (8200, "hist"), (1031, "fft"), (1033, "fixed"),
                                    (4099, "hist")])
def test_ragged_batches_ride_in_one_launch(oracle, F, mode):
    """A batch that is not whole workgroups: round 5 sent the remainder to a second, serialized launch (one more 2048-step serial chain);
    since round 6 rx_lean_kernel's last workgroup carries pad frames -- they read the batch's last frame, their symbols go to a pad
    buffer, their lanes of the serial wave are off, and the last frame of an ODD batch (half a two-frame unit) is copied to its place
    behind the launch.  Every frame against the oracle, with per-frame timing offsets (histogram mode) and the in-launch FFT estimate
    too; the outputs sit in a buffer with a canary behind them."""
    import torch
    from oracle.pyoracle import TIMING_FFT, TIMING_HIST as TH
    fs, rs, L = 19200.0, 2400.0, 1024
    tm = {"fixed": TIMING_FIXED, "hist": TH, "fft": TIMING_FFT}[mode]
    m = modem(fs=fs, rs=rs, frame_size=L, timing_mode=tm, fixed_index=6)
    x, _ = make_frames(F, L, 8, m.taps, fs, offset_hz=40.0, base_seed=300 + F, noise=0.04)
    x[2::7] = random_frames(len(x[2::7]), L, seed=F)
    x[-1] = random_frames(1, L, seed=F + 1)[0]          # the frame the pad frames read
    want = oracle.rx_batch(x, fs, rs, loop_bw=BW, timing_mode=tm, fixed_index=6, threads=min(16, os.cpu_count() or 1))
    xd = torch.from_numpy(x).cuda()
    N = m.nsym
    sym = torch.full((F * N + 4096,), 0xEE, dtype=torch.uint8, device="cuda")
    fr = torch.full((F + 64,), 7.0, dtype=torch.float32, device="cuda")
    ph = torch.full((F + 64,), 7.0, dtype=torch.float32, device="cuda")
    m.rx_batch_raw(xd, F, sym, fr, ph)
    m.sync()
    assert m.last_kernel().startswith("rx_lean_kernel"), m.last_kernel()
    assert np.array_equal(cpu(sym[:F * N]).reshape(F, N), want["sym"])
    assert bits_equal(cpu(fr[:F]), want["freq"]) and bits_equal(cpu(ph[:F]), want["phase"])
    assert bool((sym[F * N:] == 0xEE).all()) and bool((fr[F:] == 7.0).all()) and bool((ph[F:] == 7.0).all()), "wrote behind the batch"
    got = m.rx_batch(xd)
    m.sync()
    assert_batch_equal(got, want, keys=("sym", "phase", "freq", "index", "hz"))


def test_full_size_ragged_batch_properties(oracle):
    """config 2's frame size with a batch that is not whole workgroups (4097 and 8191 frames x 16384 samples, one launch of rx_lean_kernel
    with pad frames): frames are independent (qpsk.c:36-53), so (a) every frame of the ragged batch equals the same frame of the whole
    batch it extends / is cut from, bit for bit, (b) its last frames equal what a small batch of them gives, (c) a spread sample equals
    the oracle, (d) nothing is written behind the outputs."""
    import torch
    import bench
    fs, rs, L = bench.FS, bench.RS, 16384
    m = modem(fs=fs, rs=rs, frame_size=L, timing_mode=TIMING_FIXED, fixed_index=6)
    N = m.nsym
    dev = torch.device("cuda", 0)
    x = bench.synth_frames_gpu(torch, dev, 8192, m.taps, seed=21)

    def run(first, count, lean=True):
        sym = torch.full((count * N + 4096,), 0xEE, dtype=torch.uint8, device=dev)
        fr = torch.full((count + 64,), 7.0, dtype=torch.float32, device=dev)
        ph = torch.full((count + 64,), 7.0, dtype=torch.float32, device=dev)
        m.rx_batch_raw(x[first:first + count], count, sym, fr, ph)
        m.sync()
        assert not lean or m.last_kernel() == "rx_lean_kernel", m.last_kernel()
        assert bool((sym[count * N:] == 0xEE).all()) and bool((fr[count:] == 7.0).all()) and bool((ph[count:] == 7.0).all()), "wrote behind the batch"
        return cpu(sym[:count * N]).reshape(count, N), cpu(fr[:count]), cpu(ph[:count])

    whole = run(0, 8192)
    for first, count in ((0, 4097), (1, 8191), (4095, 4097)):
        got = run(first, count)
        for g, w in zip(got, whole):
            assert bits_equal(g, w[first:first + count]), (first, count)
    tail = run(8192 - 5, 5, lean=False)      # (whatever kernel serves five frames)
    for g, w in zip(tail, whole):
        assert bits_equal(g, w[8192 - 5:])
    pick = np.unique(np.concatenate([np.arange(0, 8192, 397), [4095, 4096, 8190, 8191]]))
    want = oracle.rx_batch(x[torch.from_numpy(pick).to(dev)].cpu().numpy(), fs, rs, loop_bw=BW, timing_mode=TIMING_FIXED, fixed_index=6)
    assert np.array_equal(whole[0][pick], want["sym"]) and bits_equal(whole[1][pick], want["freq"]) and bits_equal(whole[2][pick], want["phase"])


def test_stream_calls_are_all_or_poisoned(oracle):
    """include/qpsk_hip.h, STREAMS, error contract (no test reached these paths): a kernel of a stream call reports that it
    gave up a bounded wait (injected through qpsk_test_inject_status: the status word a kernel would have written) -> the call that
    synchronises next fails with QPSK_ERR_HIP, every stream call after it -- the loop-state accessors included -- is refused with
    QPSK_ERR_STATE, qpsk_streams_reset() makes the streams usable again and what they then compute is the oracle's.  A flagged NUMBER
    (QPSK_ERR_RANGE) does not poison, and neither does a failing BATCH call while no stream work is in flight."""
    import qpsk_amd
    fs, rs, L, n = 19200.0, 2400.0, 1024, 6
    m = modem(fs=fs, rs=rs, frame_size=L, timing_mode=TIMING_FIXED, fixed_index=6)
    x, _ = make_frames(3 * n, L, 8, m.taps, fs, offset_hz=35.0, base_seed=91, noise=0.03)
    blocks = x.reshape(3, n, L, 2)

    def run_blocks():
        m.streams_reset(n)
        out = [m.streams_rx_cplx(blocks[b], want_costas=False) for b in range(3)]
        m.sync()
        return out

    def rc_of(fn):
        try:
            fn()
            return 0
        except qpsk_amd.QpskError as e:
            return e.args[0] if e.args and isinstance(e.args[0], int) else str(e)

    ref = run_blocks()
    want_state = m.streams_loop_state()
    # 1. a stream kernel "gave up": the next synchronisation reports it, then everything is refused until the reset
    m.streams_reset(n)
    m.streams_rx_cplx(blocks[0], want_costas=False)
    m._check(m.L.qpsk_test_inject_status(m.h, 1))
    assert "timed out" in str(rc_of(m.sync)), "the injected pipeline time-out was not reported"
    for call in (lambda: m.streams_rx_cplx(blocks[1], want_costas=False), m.streams_loop_state,
                 lambda: m._check(m.L.qpsk_streams_set_loop_state(m.h, (C.c_float * (2 * n))()))):
        msg = str(rc_of(call))
        assert "qpsk_streams_reset" in msg, msg
    # 2. the reset clears it, and the streams compute the oracle's bits again, block after block
    m.streams_reset(n)
    om = [oracle.modem(fs, rs, L, loop_bw=BW, timing_mode=TIMING_FIXED, fixed_index=6) for _ in range(n)]
    for b in range(3):
        o = m.streams_rx_cplx(blocks[b], want_costas=False)
        m.sync()
        for s_ in range(n):
            om[s_].rx_cplx(blocks[b][s_])
            assert bits_equal(cpu(o["sym"][s_]), om[s_].symbols), (b, s_)
            assert cpu(o["phase"])[s_] == om[s_].phase and cpu(o["freq"])[s_] == om[s_].freq, (b, s_)
            assert bits_equal(cpu(o["sym"][s_]), cpu(ref[b]["sym"][s_]))
    assert bits_equal(m.streams_loop_state(), want_state)
    # 3. a flagged number does not poison
    m.streams_reset(n)
    m.streams_rx_cplx(blocks[0], want_costas=False)
    m._check(m.L.qpsk_test_inject_status(m.h, 3))
    assert "NaN" in str(rc_of(m.sync))
    m.streams_rx_cplx(blocks[1], want_costas=False)          # not refused
    m.sync()
    # 4. a batch call's time-out with no stream work in flight leaves the streams alone
    m.streams_reset(n)
    m.rx_batch(blocks[0])
    m._check(m.L.qpsk_test_inject_status(m.h, 1))
    assert "timed out" in str(rc_of(m.sync))
    out = [m.streams_rx_cplx(blocks[b], want_costas=False) for b in range(3)]
    m.sync()
    for a, b in zip(out, ref):
        for k in ("sym", "freq", "phase"):
            assert bits_equal(cpu(a[k]), cpu(b[k])), k


ONEPASS = "rx_hist_kernel (one pass on the guessed index) + rx_fused_kernel (fall-back list)"


@pytest.mark.parametrize("F", [4096, 1000, 37])
def test_histogram_mode_in_one_pass_on_a_guessed_index(oracle, F):
    """Round 6: histogram timing (the reference's own, qpsk.c:127-191) reads the batch ONCE when the context has a guess -- the majority
    index of its previous histogram-mode batch: rx_hist_kernel runs the receive path on the guess inside the scan kernel's workgroup,
    frames whose true index differs go to a list that a fall-back pass redoes.  A modem's first call has no guess (two launches);
    then (a) the same kind of batch again: the guess holds for the clean frames, the noise / tone frames mixed in (other indices)
    exercise the fall-back; (b) a batch delayed by three samples: the guess is wrong for EVERY clean frame; (c) the route forced
    (QPSK_HIST_ONEPASS = 1) on a batch of all indices; (d) the route off.  Every frame of every call against the oracle, index
    included; ragged batch sizes (not whole 16-frame workgroups) too."""
    import torch
    from oracle.pyoracle import TIMING_HIST as TH
    fs, rs, L = 19200.0, 2400.0, 2048
    m = modem(fs=fs, rs=rs, frame_size=L, timing_mode=TH)
    nthr = min(16, os.cpu_count() or 1)

    def batch(seed, delay=0, mixed=False):
        x, _ = make_frames(F, L + 8, 8, m.taps, fs, offset_hz=40.0, base_seed=seed, noise=0.03)
        x = np.ascontiguousarray(x[:, delay:delay + L])
        step = 2 if mixed else 29              # (below an eighth of the batch: more misses than that put the route on hold)
        x[1::step] = random_frames(len(x[1::step]), L, seed=seed + 1)
        rng, n = np.random.default_rng(seed), np.arange(L)
        for f in range(3, F, 5 if mixed else 31):
            z = np.exp(1j * rng.uniform(0.01, 0.5) * n) * rng.uniform(0.2, 3)
            x[f, :, 0], x[f, :, 1] = z.real, z.imag
        return x

    def check(x, kernel):
        want = oracle.rx_batch(x, fs, rs, loop_bw=BW, timing_mode=TH, threads=nthr)
        got = m.rx_batch(torch.from_numpy(x).cuda())
        m.sync()
        if kernel is not None:
            assert (m.last_kernel() == ONEPASS) == kernel, (m.last_kernel(), kernel)
        assert_batch_equal(got, want, keys=("sym", "phase", "freq", "index", "hz"))
        return want

    w0 = check(batch(10), False)                     # no guess yet
    m.tune(hist_onepass=1)                           # the route whenever a guess exists
    w1 = check(batch(20), True)                      # (a)
    assert np.bincount(w1["index"], minlength=8).argmax() == np.bincount(w0["index"], minlength=8).argmax()
    w2 = check(batch(30, delay=3), True)             # (b): every clean frame misses
    assert np.bincount(w2["index"], minlength=8).argmax() != np.bincount(w1["index"], minlength=8).argmax()
    check(batch(50, mixed=True), True)               # (c)
    check(batch(60, delay=5), True)
    m.tune(hist_onepass=0)
    check(batch(70), False)                          # (d)
    # the library's own choice: the route only while every frame of the last batch sat on the batch's majority index
    m.tune(hist_onepass=None)
    st = (C.c_int32 * 5)()
    xc, _ = make_frames(F, L, 8, m.taps, fs, offset_hz=40.0, base_seed=80, noise=0.0)
    for k in range(3):
        want = oracle.rx_batch(xc, fs, rs, loop_bw=BW, timing_mode=TH, threads=nthr)
        m._check(m.L.qpsk_test_hist_state(m.h, st))
        expect = st[3] > 0 and st[4] == 0                  # what the host will read
        check(xc, expect)
        uniform = len(np.unique(want["index"])) == 1
        m._check(m.L.qpsk_test_hist_state(m.h, st))
        assert (st[4] == 0) == uniform, (list(st), np.bincount(want["index"], minlength=8))
    check(batch(90, mixed=True), None)                   # a batch of mixed indices ...
    check(batch(100), False)                             # ... keeps the next call off the route


def test_full_size_config2_bench_stimulus_every_frame(oracle):
    """the batch bench.py TIMES: config 2 at full size built by the library's own transmit chain (bench.tx_frames_gpu, the default
    --stimulus tx, rank 0's seed) -- EVERY one of the 4096 frames against the oracle, bit for bit (the oracle's fixed-offset path runs
    56 Msamples/s per core); the other full-size tests use the torch stimulus and a spread sample"""
    import torch
    import bench
    import qpsk_amd
    fs, rs, L, F = bench.FS, bench.RS, 16384, 4096
    dev = torch.device("cuda", 0)
    m = modem(fs=fs, rs=rs, frame_size=L, timing_mode=TIMING_FIXED, fixed_index=bench.FIXED_INDEX)
    x = bench.tx_frames_gpu(torch, dev, qpsk_amd, F, seed=1000)
    a = m.rx_batch(x)
    m.sync()
    assert m.last_kernel() == "rx_lean_kernel"          # config 2's kernel since round 5 (16-frame workgroups, LDS-DMA staging)
    want = oracle.rx_batch(x.cpu().numpy(), fs, rs, loop_bw=BW, timing_mode=TIMING_FIXED, fixed_index=bench.FIXED_INDEX,
                           threads=min(16, os.cpu_count() or 1))
    for k in ("sym", "phase", "freq", "hz"):
        assert bits_equal(cpu(a[k]), want[k]), k
    assert np.all(np.abs(cpu(a["hz"]) - 50.0) < 2.0)


def test_full_size_config2_histogram_timing(oracle):
    """the reference's own timing estimate at config 2's full size (4096 x 16384) through the fused scan kernel:
    (a) a spread sample of frames against the oracle in TIMING_HIST mode, bit for bit, index included; (b) every
    frame's index equals what the three-kernel path (rrc_fir -> scan) finds; (c) the frames that got
    index i (an amplitude-bin number; nearly all frames share one) must give, bit for bit, what a batch of
    them gives with that index fixed; (d) round 6: the one-pass route on the same batch equals the two-launch route in every output."""
    import torch
    import bench
    fs, rs, L, F = bench.FS, bench.RS, 16384, 4096
    mh = modem(fs=fs, rs=rs, frame_size=L, timing_mode=TIMING_HIST)
    x = bench.synth_frames_gpu(torch, torch.device("cuda", 0), F, mh.taps, seed=9)
    a = mh.rx_batch(x)
    mh.sync()
    pick = np.unique(np.concatenate([np.arange(0, F, 211), [1, 15, 16, 17, F - 16, F - 1]]))
    want = oracle.rx_batch(x[torch.from_numpy(pick).cuda()].cpu().numpy(), fs, rs, loop_bw=BW, timing_mode=TIMING_HIST)
    for k in ("sym", "phase", "freq", "hz", "index"):
        assert bits_equal(cpu(a[k])[pick], want[k]), k
    # round 6: the same batch again on the ONE-PASS route (rx_hist_kernel on the first call's majority index, the frames off it through
    # the fall-back pass): every output of every frame as the two-launch route left it
    assert "rx_hist_kernel" not in mh.last_kernel()
    mh.tune(hist_onepass=1)
    a1 = mh.rx_batch(x)
    mh.sync()
    assert "rx_hist_kernel" in mh.last_kernel(), mh.last_kernel()
    for k in ("sym", "phase", "freq", "hz", "index"):
        assert bits_equal(cpu(a1[k]), cpu(a[k])), k
    mh.tune(hist_onepass=0)
    mh.tune(hist_generic=1)
    b = mh.rx_batch(x)
    mh.sync()
    assert np.array_equal(cpu(a["index"]), cpu(b["index"]))
    idx = cpu(a["index"])
    for ix in np.unique(idx):
        sel = np.flatnonzero(idx == ix)
        mf = modem(fs=fs, rs=rs, frame_size=L, timing_mode=TIMING_FIXED, fixed_index=int(ix))
        c = mf.rx_batch(x[torch.from_numpy(sel).cuda()].contiguous())
        mf.sync()
        for k in ("sym", "phase", "freq"):
            assert bits_equal(cpu(a[k])[sel], cpu(c[k])), (k, int(ix))


def test_full_size_config4_shard_properties(oracle):
    """BASELINE config 4's per-GPU share (8192 frames x 16384 samples, 1 GiB), which rx_lean_kernel takes as 256
    workgroups of 32 frames (ten FIR waves, 160 KB of LDS): (a) a spread sample of frames equals the oracle bit for bit, (b) the 16-frame workgroups
    of rx_fused_pipe_kernel (two rounds) give the same bits everywhere, (c) every loop ends on the +50 Hz offset."""
    import torch
    import bench
    fs, rs, L, F = bench.FS, bench.RS, 16384, 8192
    m = modem(fs=fs, rs=rs, frame_size=L, timing_mode=TIMING_FIXED, fixed_index=bench.FIXED_INDEX)
    x = bench.synth_frames_gpu(torch, torch.device("cuda", 0), F, m.taps, seed=11)
    a = m.rx_batch(x)
    m.sync()
    assert m.last_kernel() == "rx_lean_kernel"
    pick = np.unique(np.concatenate([np.arange(0, F, 211), [1, 31, 32, 33, F - 33, F - 32, F - 1]]))
    want = oracle.rx_batch(x[torch.from_numpy(pick).cuda()].cpu().numpy(), fs, rs, loop_bw=BW, timing_mode=TIMING_FIXED,
                           fixed_index=bench.FIXED_INDEX)
    for k in ("sym", "phase", "freq", "hz"):
        assert bits_equal(cpu(a[k])[pick], want[k]), k
    for v in (1, 2):     # rx_fused_pipe_kernel in two rounds, rx_pipe2_kernel: the same bits everywhere
        m.tune(pipe_v=v)
        b = m.rx_batch(x)
        m.sync()
        for k in ("sym", "phase", "freq"):
            assert bits_equal(cpu(a[k]), cpu(b[k])), (k, v)
    assert np.all(np.abs(cpu(a["hz"]) - 50.0) < 2.0)


def test_config4_whole_job_shard_after_shard(oracle):
    """BASELINE config 4 AS A JOB: 65,536 frames x 16,384 samples sharded over 8 ranks (qpsk_amd.shard.shard_range, the ranges and seeds
    bench.py's ranks use).  There is one GPU here, so the eight shards run one after the other on it: every frame of the job goes
    through rx_lean_kernel, every loop must end on the +50 Hz offset, the shards must tile the job exactly, and a spread sample of
    every shard equals the oracle bit for bit.  (The eight-GPU run is the driver's; this is the same work, serially.)"""
    import torch
    import bench
    import qpsk_amd
    from qpsk_amd.shard import shard_range
    fs, rs, L, TOTAL, WORLD = bench.FS, bench.RS, 16384, 65536, 8
    dev = torch.device("cuda", 0)
    m = modem(fs=fs, rs=rs, frame_size=L, timing_mode=TIMING_FIXED, fixed_index=bench.FIXED_INDEX)
    covered, hashes = 0, set()
    for rank in range(WORLD):
        lo, hi = shard_range(TOTAL, rank, WORLD)
        assert lo == covered and hi - lo == bench.FRAMES_PER_GPU_SHARDED
        covered = hi
        x = bench.tx_frames_gpu(torch, dev, qpsk_amd, hi - lo, seed=1000 + rank)
        a = m.rx_batch(x)
        m.sync()
        assert m.last_kernel() == "rx_lean_kernel"
        assert np.all(np.abs(cpu(a["hz"]) - 50.0) < 2.0), rank
        pick = np.unique(np.concatenate([np.arange(rank, hi - lo, 521), [0, 31, 32, hi - lo - 1]]))
        want = oracle.rx_batch(x[torch.from_numpy(pick).cuda()].cpu().numpy(), fs, rs, loop_bw=BW, timing_mode=TIMING_FIXED,
                               fixed_index=bench.FIXED_INDEX)
        for k in ("sym", "phase", "freq", "hz"):
            assert bits_equal(cpu(a[k])[pick], want[k]), (rank, k)
        hashes.add(hash(cpu(a["sym"])[:4].tobytes()))
        del x, a
    assert covered == TOTAL and len(hashes) == WORLD        # eight different shards (different seeds), the whole job


@pytest.mark.parametrize("mode", [TIMING_FIXED, TIMING_HIST])
def test_smallest_frames(oracle, mode):
    """frames of one symbol up to just over one chunk, one to three frames per call, both pipeline geometries:
    the ragged ends of every loop in the kernels"""
    fs, rs = 19200.0, 2400.0
    for L in (8, 16, 24, 136, 264, 520):
        m = modem(fs=fs, rs=rs, frame_size=L, timing_mode=mode, fixed_index=7)
        for F in (1, 3):
            x = random_frames(F, L, seed=L + F, scale=0.7)
            want = oracle.rx_batch(x, fs, rs, loop_bw=BW, timing_mode=mode, fixed_index=7, want_costas=True)
            for v in (2, 1):
                m.tune(pipe_v=v)
                got = m.rx_batch(x, want_costas=True)
                m.sync()
                assert_batch_equal(got, want)


def test_randomised_configurations(oracle):
    """a fixed-seed sweep over frame sizes, frame counts, amplitudes, loop bandwidths, clamps, timing modes and
    pipeline geometries: every output bit against the oracle (rare paths: 2 pi wraps, active clamp, exact zeros,
    ragged chunks and workgroups)"""
    rng = np.random.default_rng(20261004)
    fs, rs = 19200.0, 2400.0
    for case in range(48):
        L = 8 * int(rng.integers(1, 400))
        F = int(rng.integers(1, 80))
        scale = float(10.0 ** rng.uniform(-3, 1.5))
        bw = np.float32(TAU / rng.uniform(20, 400))
        lo, hi = (-1.0, 1.0) if case % 3 else (-float(rng.uniform(0.002, 0.3)), float(rng.uniform(0.002, 0.3)))
        mode = TIMING_HIST if case % 4 == 0 else TIMING_FIXED
        idx = int(rng.integers(0, 8))
        if case % 2:
            x = random_frames(F, L, seed=case, scale=scale)
        else:
            m0 = modem(fs=fs, rs=rs, frame_size=L)
            x, _ = make_frames(F, L, 8, m0.taps, fs, offset_hz=float(rng.uniform(-200, 200)), base_seed=case,
                               amplitude=scale, noise=0.05 * scale)
        if case % 5 == 0:
            x[rng.integers(0, F)] = 0.0
        m = modem(fs=fs, rs=rs, frame_size=L, loop_bw=bw, min_freq=lo, max_freq=hi, timing_mode=mode, fixed_index=idx)
        want = oracle.rx_batch(x, fs, rs, loop_bw=bw, min_freq=lo, max_freq=hi, timing_mode=mode, fixed_index=idx,
                               want_costas=True)
        if case % 3 == 0:     # a third of the cases through rx_fused_pipe_kernel's two geometries
            m.tune(pipe_v=1, pipe_nf=1 + case % 4)
        else:
            m.tune(pipe_v=2, pipe_g=1 + (case * 7) % 32)
        got = m.rx_batch(x, want_costas=True)
        m.sync()
        try:
            assert_batch_equal(got, want)
        except AssertionError as e:
            raise AssertionError("case %d (L %d, F %d, scale %g, bw %g, clamp %g..%g, mode %d, index %d): %s" % (
                case, L, F, scale, bw, lo, hi, mode, idx, e))


def test_two_loops_per_frame_in_both_pipeline_kernels(oracle):
    """several loops per frame (bandwidth sweeps, config 5): one lane of the serial wave and one record ring per
    (frame, loop); the host sheds frames per workgroup until the rings fit the LDS"""
    fs, rs, L, F = 19200.0, 2400.0, 2048, 45
    bws = [np.float32(TAU / 100.0), np.float32(TAU / 170.0)]
    m = modem(fs=fs, rs=rs, frame_size=L, timing_mode=TIMING_FIXED, fixed_index=3)
    x, _ = make_frames(F, L, 8, m.taps, fs, offset_hz=20.0, base_seed=77, noise=0.05)
    want = oracle.rx_batch_bw(x, fs, rs, bws, timing_mode=TIMING_FIXED, fixed_index=3)
    for v, g in ((1, None), (2, None), (2, 32), (2, 7)):
        m.tune(pipe_v=v, pipe_g=g)
        got = m.rx_batch_bw(x, bws)
        m.sync()
        assert_batch_equal(got, want, keys=("sym", "phase", "freq"))
    bws11 = [np.float32(TAU / (100.0 + 10.0 * i)) for i in range(11)]       # config 5's sweep TAU/100 .. TAU/200
    want = oracle.rx_batch_bw(x, fs, rs, bws11, timing_mode=TIMING_FIXED, fixed_index=3)
    for v in (1, 2):
        m.tune(pipe_v=v, pipe_g=None)
        got = m.rx_batch_bw(x, bws11)
        m.sync()
        assert_batch_equal(got, want, keys=("sym", "phase", "freq"))


def test_empty_and_bad_calls_are_rejected():
    import torch
    m = modem(fs=19200.0, rs=2400.0, frame_size=1024)
    x = torch.zeros((1, 1024, 2), dtype=torch.float32, device="cuda")
    sym = torch.zeros((1, 128), dtype=torch.uint8, device="cuda")
    L = m.L
    assert L.qpsk_rx_batch(m.h, C.c_void_p(x.data_ptr()), 0, C.c_void_p(sym.data_ptr()), None, None, None, None, None) == -2
    assert b"nframes" in L.qpsk_last_error()
    assert L.qpsk_rx_batch(m.h, None, 1, C.c_void_p(sym.data_ptr()), None, None, None, None, None) == -2
    assert L.qpsk_fft_batch(m.h, C.c_void_p(x.data_ptr()), C.c_void_p(x.data_ptr()), 1, 24, 0) == -2      # not a power of two
    assert L.qpsk_streams_rx_cplx(m.h, C.c_void_p(x.data_ptr()), None, None, None, None, None) == -5    # no qpsk_streams_reset yet
    import qpsk_amd
    # the fused scan kernel's conditions are checked, not assumed: whole 256-sample tiles, CYCLES = 8, aligned input
    m2 = modem(fs=19200.0, rs=2400.0, frame_size=1000)
    with pytest.raises(qpsk_amd.QpskError, match="frame_size"):
        m2.timing_scan(np.zeros((2, 1000, 2), np.float32))
    m3 = modem(fs=9600.0, rs=2400.0, frame_size=1024)
    with pytest.raises(qpsk_amd.QpskError, match="CYCLES"):
        m3.timing_scan(np.zeros((2, 1024, 2), np.float32))
    xo = torch.zeros((2 * 1024 * 2 + 2,), dtype=torch.float32, device="cuda")[2:]                        # 8 bytes off a 16-byte boundary
    idx = torch.zeros((2,), dtype=torch.int32, device="cuda")
    assert L.qpsk_timing_scan_batch(m.h, C.c_void_p(xo.data_ptr()), 2, C.c_void_p(idx.data_ptr()), None) == -2
    with pytest.raises(qpsk_amd.QpskError):
        qpsk_amd.Modem(fs=9600.0, rs=2400.0, frame_size=510)                                              # 510 % 4 != 0
    with pytest.raises(qpsk_amd.QpskError):
        qpsk_amd.Modem(fs=19200.0, rs=2400.0, frame_size=1024, timing_mode=TIMING_FIXED, fixed_index=9)


def test_config5_long_frames_bandwidth_sweep(oracle):
    """BASELINE config 5 at its real frame length: 1200 baud, 8x oversample, 1,048,576 samples per frame
    (131,072 serial Costas steps), loop bandwidths TAU/100 .. TAU/200 sharing one FIR pass; a handful of
    frames (the frame COUNT is the only thing reduced: the oracle needs ~1 s per frame)."""
    fs, rs, L, F = 9600.0, 1200.0, 1 << 20, 5
    bws = [np.float32(TAU / d) for d in range(100, 201, 10)]
    m = modem(fs=fs, rs=rs, frame_size=L, timing_mode=TIMING_FIXED, fixed_index=6)
    x, _ = make_frames(F, L, 8, m.taps, fs, offset_hz=3.0, base_seed=55, noise=0.02)
    want = oracle.rx_batch_bw(x, fs, rs, bws, timing_mode=TIMING_FIXED, fixed_index=6)
    got = m.rx_batch_bw(x, bws)
    m.sync()
    assert_batch_equal(got, want, keys=("sym", "phase", "freq"))
    # every loop of the sweep has locked onto the 3 Hz offset by the end of such a frame (README.md:12)
    hz = cpu(got["freq"]).astype(np.float64) * rs / (2 * np.pi)
    assert np.all(np.abs(hz - 3.0) < 0.5)


# ------------------------------------------------------------------ stages
@pytest.mark.parametrize("kernel", ["stream", "generic", "asymmetric"])
def test_rrc_fir_batch_with_delay_lines(oracle, kernel):
    """rrc_fir() on batches of delay lines: the generated stream with the taps in SGPRs (firstream.hip: symmetric filters), the
    compiler-scheduled kernel (QPSK_FIR_GENERIC = 1), and a tap set that is NOT symmetric (always the latter)"""
    import torch
    m = modem(fs=19200.0, rs=2400.0, frame_size=1024)
    rng = np.random.default_rng(2)
    if kernel == "generic":
        m.tune(fir_generic=1)
    if kernel == "asymmetric":
        m.set_taps(rng.standard_normal(127).astype(np.float32))
    for n in (1, 5, 126, 127, 128, 511, 512, 513, 1000, 1024, 4097):
        F = 5
        x = rng.standard_normal((F, n, 2)).astype(np.float32)
        mem = rng.standard_normal((F, 127, 2)).astype(np.float32)
        d_mem = torch.from_numpy(mem.copy()).cuda()
        y = m.rrc_fir(x, d_mem)
        m.sync()
        for f in range(F):
            ym, mm = x[f].copy(), mem[f].copy()
            oracle.rrc_fir(m.taps, mm, ym)
            assert bits_equal(cpu(y[f]), ym), (n, f)
            assert bits_equal(cpu(d_mem[f]), mm), (n, f)
    # zero history, no write-back
    x = rng.standard_normal((3, 300, 2)).astype(np.float32)
    y = m.rrc_fir(x, None)
    m.sync()
    for f in range(3):
        ym, mm = x[f].copy(), np.zeros((127, 2), np.float32)
        oracle.rrc_fir(m.taps, mm, ym)
        assert bits_equal(cpu(y[f]), ym)
    # one long delay line (cut into runs that re-read the 126 samples in front of them), many short ones, and input
    # that does not start on a 16-byte boundary
    for F, n in ((1, 70000), (2, 33333), (700, 640)):
        x = rng.standard_normal((F, n, 2)).astype(np.float32)
        mem = rng.standard_normal((F, 127, 2)).astype(np.float32)
        buf = torch.zeros((F * n + 1, 2), dtype=torch.float32, device="cuda")
        buf[1:] = torch.from_numpy(x).cuda().reshape(-1, 2)
        for xin in (x, buf[1:].reshape(F, n, 2)):
            d_mem = torch.from_numpy(mem.copy()).cuda()
            y = m.rrc_fir(xin, d_mem)
            m.sync()
            for f in sorted(set((0, F // 2, F - 1))):
                ym, mm = x[f].copy(), mem[f].copy()
                oracle.rrc_fir(m.taps, mm, ym)
                assert bits_equal(cpu(y[f]), ym), (F, n, f)
                assert bits_equal(cpu(d_mem[f]), mm), (F, n, f)


def _ulp_distance(a, b):
    """distance in float32 ulps (monotone integer mapping of the bit patterns)"""
    ia = a.astype(np.float32).view(np.int32).astype(np.int64)
    ib = b.astype(np.float32).view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7fffffff), ia)
    ib = np.where(ib < 0, -(ib & 0x7fffffff), ib)
    return np.abs(ia - ib)


def test_rrc_fir_fast_error_bounds(oracle):
    """qpsk_rrc_fir_batch_fast (overlap-save, 512-point fp32 FFTs) is NOT a parity path: its agreement
    with the exact filter (the oracle's rrc_fir() = the reference's, rrc_fir.c:17-30) is stated here as numbers.  Asserted:
    maximum error relative to the frame's peak <= 4e-6 (observed 1.3e-6); among outputs of at least 1 % of the peak the
    maximum ulp distance <= 1024 (observed 176; near the zero crossings ulps mean nothing, which is why the exact path
    exists); the delay line -- a copy of input samples -- bit for bit; every block seam (386-sample hops), ragged lengths
    and carried delay lines included.  And the exact entry point is untouched by it."""
    import torch
    fs, rs = 19200.0, 2400.0
    m = modem(fs=fs, rs=rs, frame_size=1024)
    rng = np.random.default_rng(12)
    worst_rel, worst_ulp = 0.0, 0
    for n, kind in ((1, "noise"), (100, "noise"), (386, "noise"), (387, "noise"), (1024, "noise"), (4097, "noise"), (16384, "modem")):
        F = 4
        if kind == "modem":
            x, _ = make_frames(F, n, 8, m.taps, fs, base_seed=5, noise=0.02)
        else:
            x = rng.standard_normal((F, n, 2)).astype(np.float32)
        mem = rng.standard_normal((F, 127, 2)).astype(np.float32)
        d_mem = torch.from_numpy(mem.copy()).cuda()
        y = cpu(m.rrc_fir(x, d_mem, fast=True))
        m.sync()
        for f in range(F):
            ym, mm = x[f].copy(), mem[f].copy()
            oracle.rrc_fir(m.taps, mm, ym)
            assert bits_equal(cpu(d_mem[f]), mm), (n, f)
            peak = float(np.abs(ym).max())
            rel = float(np.abs(y[f] - ym).max()) / peak
            big = np.abs(ym) >= 0.01 * peak
            ulp = int(_ulp_distance(y[f][big], ym[big]).max())
            worst_rel, worst_ulp = max(worst_rel, rel), max(worst_ulp, ulp)
            assert rel <= 4e-6, (n, f, rel)
            assert ulp <= 1024, (n, f, ulp)
    print("qpsk_rrc_fir_batch_fast: max error %.2e of the peak, max %d ulps among outputs >= 1 %% of the peak" % (worst_rel, worst_ulp))
    # zero history, no write-back; in-place is refused
    import qpsk_amd
    x = rng.standard_normal((3, 900, 2)).astype(np.float32)
    y = cpu(m.rrc_fir(x, None, fast=True))
    for f in range(3):
        ym, mm = x[f].copy(), np.zeros((127, 2), np.float32)
        oracle.rrc_fir(m.taps, mm, ym)
        assert float(np.abs(y[f] - ym).max()) <= 4e-6 * float(np.abs(ym).max())
    d = torch.from_numpy(x).cuda()
    rc = m.L.qpsk_rrc_fir_batch_fast(m.h, None, d.data_ptr(), d.data_ptr(), 3, 900)
    assert rc != 0
    # other taps: the tables follow qpsk_ctx_set_taps
    t2 = (np.asarray(m.taps) * np.float32(0.5)).astype(np.float32)
    t2[10] += np.float32(0.01)
    m.set_taps(t2)
    y = cpu(m.rrc_fir(x, None, fast=True))
    ym, mm = x[0].copy(), np.zeros((127, 2), np.float32)
    oracle.rrc_fir(t2, mm, ym)
    assert float(np.abs(y[0] - ym).max()) <= 4e-6 * float(np.abs(ym).max())


def test_fir_golden():
    import torch
    g = golden("fir.npz")
    m = modem()
    m.set_taps(g["taps"])
    d_mem = torch.from_numpy(g["mem0"].copy()[None]).cuda()
    for i in range(6):
        y = m.rrc_fir(g["x%d" % i][None], d_mem)
        m.sync()
        assert bits_equal(cpu(y[0]), g["y%d" % i]) and bits_equal(cpu(d_mem[0]), g["m%d" % i])


@pytest.mark.parametrize("cycles,L,generic", [(8, 1024, 0), (8, 1024, 1), (4, 512, 0), (5, 1000, 0), (8, 16384, 0),
                                               (8, 16384, 1), (8, 1032, 0), (8, 128, 0)])
def test_timing_histogram(oracle, cycles, L, generic):
    """index AND the summed histograms hist_i + hist_q (qpsk.c:175) for both scans (the CYCLES = 8 kernel and the
    generic one, which also takes the frames that are not whole 128-sample tiles)"""
    rs = 2400.0
    m = modem(fs=rs * cycles, rs=rs, frame_size=L)
    if generic:
        m.tune(hist_generic=2)
    x, _ = make_frames(37, L, cycles, m.taps, rs * cycles, noise=0.1, base_seed=cycles)
    x[0] = 0.0
    x[1] = random_frames(1, L, seed=5)[0]
    x[2] = np.abs(random_frames(1, L, seed=6)[0]) * np.linspace(0.01, 3.0, L)[:, None]     # the running max keeps moving
    x[3, L // 2:] = 0.0
    x[4] *= 1e-30                                                                            # averages near the denormals
    idx, hist = m.timing_hist(x, want_hist=True)
    idx, hist = cpu(idx), cpu(hist)
    for f in range(x.shape[0]):
        want_idx, want_hist = oracle.timing_hist(x[f], cycles)
        assert np.array_equal(hist[f], want_hist), (f, hist[f], want_hist)
        assert idx[f] == want_idx, f


def test_costas_batch_with_state(oracle):
    import torch
    from oracle.pyoracle import Costas
    m = modem(fs=19200.0, rs=2400.0, frame_size=1024)
    rng = np.random.default_rng(3)
    F, N = 70, 300
    d = rng.standard_normal((F, N, 2)).astype(np.float32)
    st0 = np.stack([rng.uniform(-6, 6, F), rng.uniform(-1, 1, F)], -1).astype(np.float32)
    st0[0] = [-0.0, 0.0]
    st0[1] = [6.2831855, 1.0]
    st = torch.from_numpy(st0.copy()).cuda()
    sym, z = m.costas(d, st)
    m.sync()
    for f in range(F):
        c = Costas()
        oracle.lib.qo_costas_create(C.byref(c), BW, -1.0, 1.0)
        c.phase, c.freq = float(st0[f, 0]), float(st0[f, 1])
        zr, zi = C.c_float(), C.c_float()
        for i in range(N):
            s = oracle.lib.qo_costas_step(C.byref(c), float(d[f, i, 0]), float(d[f, i, 1]), C.byref(zr), C.byref(zi))
            assert s == int(sym[f, i])
            assert np.float32(zr.value).view(np.uint32) == cpu(z[f, i, 0]).view(np.uint32)
            assert np.float32(zi.value).view(np.uint32) == cpu(z[f, i, 1]).view(np.uint32)
        assert bits_equal(cpu(st[f]), np.array([c.phase, c.freq], np.float32))


def test_costas_zero_error_and_signed_zero_state(oracle):
    """|T.x| == |T.y| makes the detector's error an exact zero whose SIGN reaches freq and phase when those are -0
    (costas_loop.c:44-59: -0 + +0 = +0); the hand-scheduled stream carries |T.y| - |T.x| and a +-1 factor instead,
    so these steps must take its exact fallback.  Symbols on the diagonals with the loop at phase -0, alone and
    scattered among random ones."""
    import torch
    from oracle.pyoracle import Costas
    m = modem(fs=19200.0, rs=2400.0, frame_size=1024)
    rng = np.random.default_rng(11)
    F, N = 24, 128
    d = rng.standard_normal((F, N, 2)).astype(np.float32)
    diag = rng.uniform(.25, 2, (F, N)).astype(np.float32)
    sx = rng.choice(np.float32([-1, 1]), (F, N))
    sy = rng.choice(np.float32([-1, 1]), (F, N))
    on_diag = np.zeros((F, N), bool)
    on_diag[:8] = True                                   # whole frames: the loop never leaves phase -0
    on_diag[8:16, :17] = True                            # a run at the start, then random symbols
    on_diag[16:] = rng.random((F - 16, N)) < .1
    d[on_diag, 0] = (diag * sx)[on_diag]
    d[on_diag, 1] = (diag * sy)[on_diag]
    st0 = np.zeros((F, 2), np.float32)
    st0[0::2] = [-0.0, -0.0]
    st0[1::4] = [-0.0, 0.0]
    st = torch.from_numpy(st0.copy()).cuda()
    sym, z = m.costas(d, st)
    m.sync()
    for f in range(F):
        c = Costas()
        oracle.lib.qo_costas_create(C.byref(c), BW, -1.0, 1.0)
        c.phase, c.freq = float(st0[f, 0]), float(st0[f, 1])
        zr, zi = C.c_float(), C.c_float()
        want = np.empty((N, 2), np.float32)
        for i in range(N):
            s = oracle.lib.qo_costas_step(C.byref(c), float(d[f, i, 0]), float(d[f, i, 1]), C.byref(zr), C.byref(zi))
            assert s == int(sym[f, i]), (f, i)
            want[i] = zr.value, zi.value
        assert bits_equal(cpu(z[f]), want), f
        assert bits_equal(cpu(st[f]), np.array([c.phase, c.freq], np.float32)), (f, cpu(st[f]), c.phase, c.freq)


def test_stream_across_ring_handovers_takes_its_fallbacks(oracle):
    """the Costas stream that runs across the chunk hand-overs (costas_asm_run_ring) abandons a group that meets an
    exact-zero detector input and the C++ step redoes it -- wherever in a 64-symbol chunk the group sits (first, inner,
    last: the last one also owes the `consumed` hand-over), also several in a row and whole zero chunks; a frame whose
    symbol count is not a whole number of chunks ends chunk by chunk.  Both forms (QPSK_PIPE_DBG=16: chunk by chunk
    everywhere) must give the oracle's bits."""
    import torch
    from oracle.pyoracle import Costas
    rng = np.random.default_rng(23)
    for N in (512, 456):
        m = modem(fs=19200.0, rs=2400.0, frame_size=N * 8)
        F = 20
        d = rng.standard_normal((F, N, 2)).astype(np.float32)
        zero_at = {0: [64], 1: [79], 2: [112], 3: [127], 4: [128, 191, 192], 5: list(range(176, 208)), 6: list(range(192, 320)),
                   7: [255, 256, 257], 8: [N - 1], 9: [N - 17], 10: list(range(64, N)), 11: [100, 200, 300, 400]}
        for f, where in zero_at.items():
            d[f, where] = 0.0
        d[12, 130] = (1.5, -1.5)          # |T.x| == |T.y| needs phase 0: not here; a plain diagonal symbol instead
        d[13] *= 1e-20                    # tiny but nonzero: no fallback
        st0 = np.zeros((F, 2), np.float32)
        st0[14] = [-0.0, -0.0]
        st0[15] = [3.0, 0.9]              # wraps every few steps
        st0[16] = [-6.2, -0.99]
        for dbg in (0, 16):
            m.tune(pipe_dbg=dbg)
            st = torch.from_numpy(st0.copy()).cuda()
            sym, z = m.costas(d, st)
            m.sync()
            for f in range(F):
                c = Costas()
                oracle.lib.qo_costas_create(C.byref(c), BW, -1.0, 1.0)
                c.phase, c.freq = float(st0[f, 0]), float(st0[f, 1])
                zr, zi = C.c_float(), C.c_float()
                want = np.empty((N, 2), np.float32)
                wsym = np.empty(N, np.uint8)
                for i in range(N):
                    wsym[i] = oracle.lib.qo_costas_step(C.byref(c), float(d[f, i, 0]), float(d[f, i, 1]), C.byref(zr), C.byref(zi))
                    want[i] = zr.value, zi.value
                assert np.array_equal(cpu(sym[f]), wsym), (N, dbg, f)
                assert bits_equal(cpu(z[f]), want), (N, dbg, f)
                assert bits_equal(cpu(st[f]), np.array([c.phase, c.freq], np.float32)), (N, dbg, f)


@pytest.mark.parametrize("pipe_v,G", [(1, None), (2, 32), (2, 5)])
def test_zero_runs_inside_frames_both_pipeline_kernels(oracle, pipe_v, G):
    """silence inside a frame (runs of zero samples longer than the filter) gives exact-zero symbols in the middle of
    the rings' chunks: the receive kernels' serial wave leaves its stream there and comes back, per frame at different
    places, while the other frames of the workgroup carry on"""
    fs, rs, L, F = 19200.0, 2400.0, 8192, 40
    m = modem(fs=fs, rs=rs, frame_size=L, timing_mode=TIMING_FIXED, fixed_index=6)
    x, _ = make_frames(F, L, 8, m.taps, fs, base_seed=77, noise=0.02)
    rng = np.random.default_rng(5)
    for f in range(0, F, 2):
        for _ in range(1 + f % 3):
            a = int(rng.integers(0, L - 200))
            x[f, a:a + int(rng.integers(140, 1500))] = 0.0
    x[3, 512 * 3 - 130:512 * 5] = 0.0        # zero symbols exactly from a chunk boundary on
    x[5, :4096] = 0.0
    want = oracle.rx_batch(x, fs, rs, loop_bw=BW, timing_mode=TIMING_FIXED, fixed_index=6, want_costas=True)
    m.tune(pipe_v=pipe_v, pipe_g=G)
    for dbg in (0, 16):
        m.tune(pipe_dbg=dbg)
        got = m.rx_batch(x, want_costas=True)
        m.sync()
        assert_batch_equal(got, want)


def test_fft_batch(oracle):
    rng = np.random.default_rng(4)
    m = modem()
    # up to 8192 points one workgroup holds the transform in LDS; above, two passes over global memory (the frame
    # lengths of the BASELINE configs: 16384 and 2^20; fft.c:110-136 take any power of two)
    for n, nb in ((1, 5), (2, 5), (8, 5), (512, 5), (2048, 5), (8192, 5), (16384, 3), (65536, 2), (1 << 20, 2), (1 << 21, 1)):
        x = rng.standard_normal((nb, n)) + 1j * rng.standard_normal((nb, n))
        X = cpu(m.fft(x))
        Xi = cpu(m.fft(x, inverse=True))
        m.sync()
        for b in range(nb):
            assert bits_equal(X[b], oracle.fftn(x[b])), n       # same host libm builds both twiddle sets
            assert bits_equal(Xi[b], oracle.ifftn(x[b])), n
    import torch
    xin = torch.from_numpy(x).cuda()                            # in place (the reference copies in -> out first, fft.c:99-101)
    assert m.L.qpsk_fft_batch(m.h, C.c_void_p(xin.data_ptr()), C.c_void_p(xin.data_ptr()), 1, 1 << 21, 0) == 0
    m.sync()
    assert bits_equal(cpu(xin)[0], X[0])
    assert m.L.qpsk_fft_batch(m.h, C.c_void_p(xin.data_ptr()), C.c_void_p(xin.data_ptr()), 1, 1 << 22, 0) == -2
    g = golden("fft_bits.npz")
    np.testing.assert_allclose(cpu(m.fft(g["x512"][None]))[0], g["fft512"], rtol=0, atol=1e-15)
    assert np.all(cpu(m.fft(np.eye(1, 512, dtype=np.complex128)))[0] == 2.0 ** -9)


# ------------------------------------------------------------------ bit-level stages
def test_bit_stages(oracle):
    """crc16 / golden-prime interleaver / DVB scrambler, batched: against the oracle on random packets and against
    the reference's recorded outputs (tests/golden/fft_bits.npz), incl. the only known answers the reference
    carries: CRC("123456789") = 0x29B1 and the interleaver strings of interleave.c:100-102."""
    m = modem()
    g = golden("fft_bits.npz")
    rng = np.random.default_rng(12)
    for nbytes in (1, 2, 8, 22, 40, 43, 64, 255):
        pk = rng.integers(0, 256, size=(37, nbytes)).astype(np.uint8)
        crc = m.crc16(pk)
        for p in range(pk.shape[0]):
            assert int(crc[p]) == oracle.crc16(pk[p].tobytes())
        fwd, back = cpu(m.interleave(pk, 0)), cpu(m.interleave(pk, 1))
        for p in range(0, pk.shape[0], 9):
            assert bits_equal(fwd[p], oracle.interleave(pk[p], 0)) and bits_equal(back[p], oracle.interleave(pk[p], 1))
    assert int(m.crc16(np.frombuffer(b"123456789", np.uint8)[None].copy())[0]) == 0x29B1
    assert bits_equal(cpu(m.interleave(g["il_in"][None], 0))[0], g["il_out"])
    assert bits_equal(cpu(m.interleave(g["il_out"][None], 1))[0], g["il_back"])
    assert bits_equal(cpu(m.interleave(g["il22_in"][None], 0))[0], g["il22_out"])
    sc = cpu(m.scramble(np.repeat(g["scr_in"][None], 5, 0)))
    for p in range(5):
        assert bits_equal(sc[p], g["scr_out"])                          # register reloaded per frame
    assert bits_equal(cpu(m.scramble(sc)), np.repeat(g["scr_in"][None], 5, 0))   # additive: twice = identity
    syms = rng.integers(0, 4, size=(3, 2048)).astype(np.uint8)
    sc2 = cpu(m.scramble(syms))
    for p in range(3):
        assert bits_equal(sc2[p], oracle.scramble_stream(syms[p]))


# ------------------------------------------------------------------ the reference's own entry points (qpsk_dropin.h)
def test_dropin_reference_signatures_golden():
    """rrc_make / create_control_loop / rx_frame / rrc_fir / fft exactly as the reference's main() calls them
    (qpsk.c:302,308,344-354), against the outputs recorded from the reference (tests/golden)."""
    import qpsk_amd
    from qpsk_amd.lib import Params
    L = qpsk_amd.load()
    g = golden("stream_pcm_shipped.npz")
    p = Params()
    L.qpsk_params_default(C.byref(p))
    L.qpsk_dropin_configure.argtypes = [C.POINTER(Params), C.c_double]
    assert L.qpsk_dropin_configure(C.byref(p), 1500.0) == 0
    L.create_control_loop.argtypes = [C.c_float] * 3
    L.rrc_make.argtypes = [C.c_float] * 3
    L.create_control_loop(np.float32(g["loop_bw"]), -1.0, 1.0)
    L.rrc_make(9600.0, 2400.0, 0.35)
    for fn in ("get_phase", "get_frequency", "get_alpha", "get_beta", "qpsk_dropin_offset_freq"):
        getattr(L, fn).restype = C.c_float
    L.qpsk_dropin_costas_frame.restype = C.POINTER(C.c_float)
    L.qpsk_dropin_symbols.restype = C.POINTER(C.c_uint8)
    fs, N = 512, 128
    for k in range(g["sym"].shape[0]):
        blk = np.ascontiguousarray(g["pcm"][k * fs:(k + 1) * fs])
        L.rx_frame(blk.ctypes.data_as(C.POINTER(C.c_int16)))
        sym = np.ctypeslib.as_array(L.qpsk_dropin_symbols(), shape=(N,)).copy()
        cf = np.ctypeslib.as_array(L.qpsk_dropin_costas_frame(), shape=(N, 2)).copy()
        assert bits_equal(sym, g["sym"][k]) and bits_equal(cf, g["costas"][k]), k
        assert np.float32(L.get_phase()) == g["phase"][k] and np.float32(L.get_frequency()) == g["freq"][k]
        assert np.float32(L.qpsk_dropin_offset_freq()) == g["hz"][k] and L.qpsk_dropin_timing_index() == g["index"][k]
    # rrc_fir(memory, sample, length) in place with a caller-owned delay line (rrc_fir.c:17-30)
    gf = golden("fir.npz")
    mem = gf["mem0"].copy()
    for i in range(6):
        y = gf["x%d" % i].copy()
        L.rrc_fir(mem.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), y.shape[0])
        assert bits_equal(y, gf["y%d" % i]) and bits_equal(mem, gf["m%d" % i]), i
    # fft(in, out): NFFT = 512 (fft.h:44)
    gb = golden("fft_bits.npz")
    x = gb["x512"].copy()
    out = np.zeros(512, np.complex128)
    L.fft(x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    np.testing.assert_allclose(out, gb["fft512"], rtol=0, atol=1e-15)
    L.qpsk_dropin_shutdown()


# ------------------------------------------------------------------ transmitters (qpsk.c:225-285)
def dibits(bits):
    """tx_bits[] -> the symbol value qpsk_packet_mod() looks up: (tx_bits[s] << 1) | tx_bits[s+1] (qpsk.c:277-281)"""
    bits = np.asarray(bits, np.uint8)
    return ((bits[..., 0::2] << 1) | bits[..., 1::2]).astype(np.uint8)


@pytest.mark.parametrize("name", ["shipped", "c1small"])
def test_tx_golden(name):
    g = golden("tx_%s.npz" % name)
    fs, rs = float(g["fs"]), float(g["rs"])
    m = modem(fs=fs, rs=rs, frame_size=int(fs / rs) * 64)
    m.tx_reset(3, float(g["tx_hz"]))
    for bits, pcm, bb in zip(g["bits"], g["pcm"], g["baseband"]):
        o = m.tx_symbols(np.repeat(dibits(bits)[None], 3, 0), want_pcm=True, want_baseband=True)
        m.sync()
        for s in range(3):
            assert np.array_equal(cpu(o["pcm"][s]), pcm)
            assert bits_equal(cpu(o["baseband"][s]), bb)


@pytest.mark.parametrize("fs,rs,S", [(19200.0, 2400.0, 37), (9600.0, 2400.0, 16), (9600.0, 1200.0, 5)])
def test_tx_vs_oracle_many_transmitters(oracle, fs, rs, S):
    """every transmitter has its own symbols, block lengths are ragged (1 symbol .. several tiles), state carried"""
    cycles = int(fs / rs)
    m = modem(fs=fs, rs=rs, frame_size=cycles * 64)
    m.tx_reset(S, 1550.0)
    otx = [oracle.tx(fs, rs, np.float32(.35), 1550.0) for _ in range(S)]
    rng = np.random.default_rng(41)
    for nsym in (1, 7, 64, 129, 1000, 3):
        bits = rng.integers(0, 2, size=(S, 2 * nsym)).astype(np.int32)
        o = m.tx_symbols(dibits(bits))
        m.sync()
        got = cpu(o["pcm"])
        for s in range(S):
            assert np.array_equal(got[s], otx[s].symbols(bits[s])), (nsym, s)
    with pytest.raises(Exception):
        m.tx_symbols(np.zeros((S + 1, 4), np.uint8))


def test_tx_requires_reset():
    m = modem(fs=9600.0, rs=2400.0, frame_size=512)
    with pytest.raises(Exception):
        m.tx_symbols(np.zeros((1, 4), np.uint8))


# ------------------------------------------------------------------ streams (consecutive rx_frame calls)
@pytest.mark.parametrize("block", [1, 0, 2])
@pytest.mark.parametrize("name", ["shipped", "c1small"])
def test_streams_pcm_golden(name, block):
    """consecutive rx_frame() calls against the reference's recordings: block = 1 the one-launch-per-block kernel
    (streamblock.hip: what few short streams get), 0 the composition for few / long streams (mixer, filter, scan and loop kernels),
    2 mixer + filter + scan as one kernel (streamscan.hip: what thousands of streams get; needs whole 256-sample tiles at CYCLES = 8)"""
    g = golden("stream_pcm_%s.npz" % name)
    L = int(g["frame_size"])
    m = modem(fs=float(g["fs"]), rs=float(g["rs"]), frame_size=L, loop_bw=np.float32(g["loop_bw"]))
    m.tune(stream_block=1 if block == 1 else 0)
    m.tune(stream_scan=1 if block == 2 else 0)
    m.streams_reset(3, 1500.0)
    for k in range(g["sym"].shape[0]):
        blk = np.repeat(g["pcm"][k * L:(k + 1) * L][None], 3, 0)
        o = m.streams_rx_pcm(blk)
        m.sync()
        for s in range(3):
            assert cpu(o["index"])[s] == g["index"][k]
            assert bits_equal(cpu(o["sym"][s]), g["sym"][k]) and bits_equal(cpu(o["costas"][s]), g["costas"][k])
            assert cpu(o["phase"])[s] == g["phase"][k] and cpu(o["freq"])[s] == g["freq"][k]
    assert (m.last_kernel() == "stream_block_kernel") == (block == 1)


def test_streams_with_fft_timing(oracle):
    """the FFT timing estimate in the streaming mode: per block, from the raw block (it starts at sample 128 and
    so never touches the carried delay line); everything downstream carries state as usual"""
    from oracle.pyoracle import TIMING_FFT
    fs, rs, L, S = 19200.0, 2400.0, 2048, 5
    m = modem(fs=fs, rs=rs, frame_size=L, timing_mode=TIMING_FFT)
    m.streams_reset(S)
    om = [oracle.modem(fs, rs, L, loop_bw=BW, timing_mode=TIMING_FFT) for _ in range(S)]
    x, _ = make_frames(S, L * 4, 8, m.taps, fs, offset_hz=-40.0, base_seed=23, noise=0.03)
    for k in range(4):
        blk = np.ascontiguousarray(x[:, k * L:(k + 1) * L])
        o = m.streams_rx_cplx(blk)
        m.sync()
        for s in range(S):
            om[s].rx_cplx(blk[s])
            assert cpu(o["index"])[s] == om[s].index
            assert bits_equal(cpu(o["sym"][s]), om[s].symbols) and bits_equal(cpu(o["costas"][s]), om[s].costas_frame)
            assert cpu(o["phase"])[s] == om[s].phase and cpu(o["freq"])[s] == om[s].freq


@pytest.mark.parametrize("generic,L,S", [(0, 1024, 9), (1, 1024, 9), (2, 1024, 9), (2, 2048, 70), (2, 1000, 3), (0, 4096, 5), (4, 1024, 9),
                                         (4, 2048, 40), (4, 256, 17)])
def test_streams_cplx_vs_oracle(oracle, generic, L, S):
    """generic = 1: the barrier-synchronised kernels (decimate_kernel + costas_kernel) instead of the pipeline; 0: the
    composition of filter, scan and loop kernels; 2: the one-launch-per-block kernel (the library's own choice for few short
    streams), also with a frame that is not whole 512-sample tiles or whole 16-symbol groups; 4: filter + scan as one kernel with the
    filtered block left planar for the loop kernel's picks (streamscan.hip, complex input: what thousands of streams get)"""
    fs, rs = 19200.0, 2400.0
    m = modem(fs=fs, rs=rs, frame_size=L)
    if generic == 1:
        m.tune(fused_generic=1)
    m.tune(stream_block=1 if generic == 2 else 0)
    m.tune(stream_scan=1 if generic == 4 else 0)
    m.streams_reset(S)
    om = [oracle.modem(fs, rs, L, loop_bw=BW) for _ in range(S)]
    x, _ = make_frames(S, L * 5, 8, m.taps, fs, offset_hz=30.0, base_seed=17, noise=0.02)
    for k in range(5):
        blk = np.ascontiguousarray(x[:, k * L:(k + 1) * L])
        o = m.streams_rx_cplx(blk)
        m.sync()
        for s in range(S):
            om[s].rx_cplx(blk[s])
            assert cpu(o["index"])[s] == om[s].index
            assert bits_equal(cpu(o["sym"][s]), om[s].symbols) and bits_equal(cpu(o["costas"][s]), om[s].costas_frame)
            assert cpu(o["phase"])[s] == om[s].phase and cpu(o["freq"])[s] == om[s].freq
    assert (m.last_kernel() == "stream_block_kernel") == (generic == 2)


@pytest.mark.parametrize("fs,L,S,fixed", [(12000.0, 1000, 3, None), (19200.0, 520, 2, None), (9600.0, 2048, 40, None), (9600.0, 1020, 2, None),
                                          (19200.0, 512, 5, 3), (9600.0, 28, 4, None)])
@pytest.mark.parametrize("carrier", [1, 0])
def test_stream_block_kernel_shapes(oracle, fs, L, S, fixed, carrier):
    """the one-launch-per-block kernel on PCM streams block after block against the oracle's modem: CYCLES = 5 (a true division in
    the scan), blocks that are not whole 512-sample tiles, 16-symbol groups or 16-byte rows of PCM, an odd length, more streams,
    a block shorter than the filter, fixed timing; every block's index, symbols, costas_frame[], loop state"""
    from oracle.pyoracle import TIMING_HIST
    rs = 2400.0
    mode = TIMING_FIXED if fixed is not None else TIMING_HIST
    kw = dict(timing_mode=mode, fixed_index=fixed) if fixed is not None else dict(timing_mode=mode)
    m = modem(fs=fs, rs=rs, frame_size=L, **kw)
    m.tune(stream_block=1)
    m.tune(stream_carrier=carrier)      # 1: the streams' one carrier from the table (carrier.h), 0: every stream's own recurrence
    m.streams_reset(S, 1500.0)
    om = [oracle.modem(fs, rs, L, loop_bw=BW, **kw) for _ in range(S)]
    for o in om:
        o.set_mixer_hz(1500.0)
    rng = np.random.default_rng(int(fs) + L)
    for k in range(5):
        pcm = (5000 * rng.standard_normal((S, L))).astype(np.int16)
        if k == 2:
            pcm[0] = 0                                     # an all-zero block: index 1, zero symbols next block
        o = m.streams_rx_pcm(pcm)
        m.sync()
        assert m.last_kernel() == "stream_block_kernel"
        for s_ in range(S):
            om[s_].rx_pcm(pcm[s_])
            assert cpu(o["index"])[s_] == om[s_].index, (k, s_)
            assert bits_equal(cpu(o["sym"][s_]), om[s_].symbols) and bits_equal(cpu(o["costas"][s_]), om[s_].costas_frame), (k, s_)
            assert cpu(o["phase"])[s_] == om[s_].phase and cpu(o["freq"])[s_] == om[s_].freq, (k, s_)


@pytest.mark.parametrize("fixed", [None, 5])
@pytest.mark.parametrize("carrier", [1, 0])
@pytest.mark.parametrize("L,S", [(2048, 47), (256, 33), (1024, 16), (4096, 5)])
def test_streams_pcm_mixer_filter_and_scan_in_one_kernel(oracle, L, S, carrier, fixed):
    """stream_scan_kernel (PCM in; the scan fed from LDS, the filtered block left planar by decimation phase for the loop kernel's
    picks) block after block against the oracle's modems: several workgroups and a ragged last one, one to sixteen tiles per block,
    state carried through five blocks (delay lines, carrier phase, loop, picks), an all-zero block; and equal to the four kernels
    apart bit for bit.  carrier = 1: the streams' ONE carrier from the table the block before left (a spare wave of workgroup 0 runs
    the next block's); 0: the carrier recurrences of a workgroup's 16 streams by its mixer wave, a tile ahead of its filter waves"""
    fs, rs = 19200.0, 2400.0
    kw = dict(timing_mode=TIMING_FIXED, fixed_index=fixed) if fixed is not None else {}      # fixed timing rides on the same kernel
    m = modem(fs=fs, rs=rs, frame_size=L, **kw)
    m2 = modem(fs=fs, rs=rs, frame_size=L, **kw)
    for mm, scan in ((m, 1), (m2, 0)):
        mm.tune(stream_block=0)
        mm.tune(stream_scan=scan)
        mm.tune(stream_carrier=carrier)
        mm.streams_reset(S, 1500.0)
    om = [oracle.modem(fs, rs, L, loop_bw=BW, **kw) for _ in range(S)]
    for o in om:
        o.set_mixer_hz(1500.0)
    rng = np.random.default_rng(L + S)
    for k in range(5):
        pcm = (6000 * rng.standard_normal((S, L))).astype(np.int16)
        if k == 3:
            pcm[1] = 0
        o1, o2 = m.streams_rx_pcm(pcm), m2.streams_rx_pcm(pcm)
        m.sync(); m2.sync()
        for key in ("sym", "costas", "phase", "freq", "index"):
            assert bits_equal(cpu(o1[key]), cpu(o2[key])), (k, key)
        for s_ in sorted(set((0, 1, S // 2, S - 2, S - 1))):
            om[s_].rx_pcm(pcm[s_])
            assert cpu(o1["index"])[s_] == om[s_].index, (k, s_)
            assert bits_equal(cpu(o1["sym"][s_]), om[s_].symbols) and bits_equal(cpu(o1["costas"][s_]), om[s_].costas_frame), (k, s_)
            assert cpu(o1["phase"])[s_] == om[s_].phase and cpu(o1["freq"])[s_] == om[s_].freq, (k, s_)


def test_stream_block_kernel_flags_bad_input(oracle):
    """NaN PCM cannot exist (int16), but a complex block can carry one: the one-launch kernel flags the call like the batch kernels do
    (QPSK_ERR_RANGE at the next synchronisation) and the context stays usable"""
    import qpsk_amd
    fs, rs, L, S = 19200.0, 2400.0, 1024, 3
    m = modem(fs=fs, rs=rs, frame_size=L)
    m.tune(stream_block=1)
    m.streams_reset(S)
    x, _ = make_frames(S, L * 3, 8, m.taps, fs, base_seed=2)
    blk = np.ascontiguousarray(x[:, :L])
    blk[1, 100, 0] = np.float32("nan")
    m.streams_rx_cplx(blk)             # the NaN reaches stream 1's picks ...
    m.sync()
    m.streams_rx_cplx(np.ascontiguousarray(x[:, L:2 * L]))      # ... and its loop in the next block
    with pytest.raises(qpsk_amd.QpskError, match="-6"):
        m.sync()
    m.streams_reset(S)
    om = [oracle.modem(fs, rs, L, loop_bw=BW) for _ in range(S)]
    for k in range(2):
        b = np.ascontiguousarray(x[:, k * L:(k + 1) * L])
        o = m.streams_rx_cplx(b)
        m.sync()
        for s_ in range(S):
            om[s_].rx_cplx(b[s_])
            assert bits_equal(cpu(o["sym"][s_]), om[s_].symbols)


# ------------------------------------------------------------------ robustness of the product library
def test_environment_cannot_change_results(oracle, monkeypatch):
    """the ablation bits of QPSK_PIPE_DBG (1: skip the filter arithmetic, 2: skip the recurrence) exist only in the
    measurement build; the product library returns the oracle's bits whatever the environment or the tuning call says"""
    fs, rs, L, F = 19200.0, 2400.0, 2048, 70
    monkeypatch.setenv("QPSK_PIPE_DBG", "3")       # read by qpsk_ctx_create()
    m = modem(fs=fs, rs=rs, frame_size=L, timing_mode=TIMING_FIXED, fixed_index=6)
    x, _ = make_frames(F, L, 8, m.taps, fs, base_seed=5, noise=0.03)
    want = oracle.rx_batch(x, fs, rs, loop_bw=BW, timing_mode=TIMING_FIXED, fixed_index=6, want_costas=True)
    got = m.rx_batch(x, want_costas=True)
    m.sync()
    assert_batch_equal(got, want)
    m.tune(pipe_dbg=3)
    got = m.rx_batch(x, want_costas=True)
    m.sync()
    assert_batch_equal(got, want)
    import qpsk_amd
    with pytest.raises(qpsk_amd.QpskError):
        m.tune(no_such_knob=1)


@pytest.mark.parametrize("generic", [0, 1])
def test_huge_amplitude_is_an_error_not_a_hang(oracle, generic):
    """frames of amplitude 1e12: alpha*e reaches ~1e11, where the reference's unbounded phase_wrap() (costas_loop.c:61-67)
    never returns ((float)((double)p - TAU) == p from 2^27 on).  The kernels bound the wrap and the call fails with
    QPSK_ERR_RANGE at the next synchronisation; the context stays usable and ordinary frames still give the oracle's bits"""
    import qpsk_amd
    fs, rs, L, F = 19200.0, 2400.0, 1024, 20
    m = modem(fs=fs, rs=rs, frame_size=L, timing_mode=TIMING_FIXED, fixed_index=6)
    if generic:
        m.tune(fused_generic=1)
    x, _ = make_frames(F, L, 8, m.taps, fs, base_seed=8)
    bad = x.copy()
    bad[3] *= np.float32(1e12)
    m.rx_batch(bad)
    with pytest.raises(qpsk_amd.QpskError, match="-6"):
        m.sync()
    got = m.rx_batch(x)
    m.sync()                                         # the flag was cleared by the failed sync
    want = oracle.rx_batch(x, fs, rs, loop_bw=BW, timing_mode=TIMING_FIXED, fixed_index=6)
    assert_batch_equal(got, want, keys=("sym", "phase", "freq"))
    # phases the reference does wrap (a few hundred turns per step) are wrapped exactly as it does
    big = (x * np.float32(3e3)).astype(np.float32)
    got = m.rx_batch(big)
    m.sync()
    want = oracle.rx_batch(big, fs, rs, loop_bw=BW, timing_mode=TIMING_FIXED, fixed_index=6)
    assert_batch_equal(got, want, keys=("sym", "phase", "freq"))


@pytest.mark.parametrize("kernel", ["pipe", "lean", "generic"])
@pytest.mark.parametrize("bad", ["nan", "inf", "-inf"])
def test_nonfinite_input_is_an_error(oracle, kernel, bad):
    """NaN / Inf samples are fenced: the call RETURNS (no wave spins: every wait and wrap is bounded) and
    fails with QPSK_ERR_RANGE at the next synchronisation -- a NaN keeps its loop's state NaN to the end of the frame
    (STATUS_NONFINITE), an infinity runs into the bounded 2 pi wrap (STATUS_PHASE_RANGE; the reference hangs there,
    costas_loop.c:61-67).  The frames WITHOUT such a sample in the same call still carry the oracle's bits, and the context
    stays usable."""
    import qpsk_amd
    fs, rs = 19200.0, 2400.0
    L, F = (1024, 64) if kernel != "lean" else (1024, 512)       # lean: whole even workgroups of whole 64-symbol chunks
    m = modem(fs=fs, rs=rs, frame_size=L, timing_mode=TIMING_FIXED, fixed_index=6)
    if kernel == "generic":
        m.tune(fused_generic=1)
    if kernel == "lean":
        m.tune(pipe_v=3)
    x, _ = make_frames(F, L, 8, m.taps, fs, base_seed=31, noise=0.02)
    want = oracle.rx_batch(x, fs, rs, loop_bw=BW, timing_mode=TIMING_FIXED, fixed_index=6)
    xb = x.copy()
    hit = [3, F // 2 + 1]
    xb[hit[0], 300, 0] = np.float32(bad)
    xb[hit[1], L - 5, 1] = np.float32(bad)
    got = m.rx_batch(xb)
    with pytest.raises(qpsk_amd.QpskError, match="-6"):
        m.sync()
    if kernel == "lean":
        assert m.last_kernel() == "rx_lean_kernel"
    clean = [f for f in range(F) if f not in hit]
    for k in ("sym", "phase", "freq"):
        assert bits_equal(got[k].cpu().numpy()[clean], want[k][clean]), k
    # symbols of a hit frame before the sample's first filter output are untouched too
    assert bits_equal(got["sym"].cpu().numpy()[hit[0], :(300 - 6) // 8 - 1], want["sym"][hit[0], :(300 - 6) // 8 - 1])
    got = m.rx_batch(x)
    m.sync()
    assert_batch_equal(got, want, keys=("sym", "phase", "freq"))


@pytest.mark.parametrize("kernel,L,F,extra", [("pipe", 1024, 40, 64), ("lean", 1024, 512, 512), ("lean", 2048, 96, 34), ("generic", 1000, 9, 2)])
def test_pitched_frames(oracle, kernel, L, F, extra):
    """qpsk_rx_batch_pitched: frames frame_size + extra samples apart (the gap filled with NaN: never read) give the packed
    call's bits on every receive kernel and in every timing mode; a pitch below frame_size is refused."""
    import torch
    import qpsk_amd
    fs, rs = 19200.0, 2400.0
    cyc = 8 if kernel != "generic" else 5
    fs = rs * cyc
    m = modem(fs=fs, rs=rs, frame_size=L, timing_mode=TIMING_FIXED, fixed_index=3)
    if kernel == "lean":
        m.tune(pipe_v=3)
    x, _ = make_frames(F, L, cyc, m.taps, fs, base_seed=77, noise=0.03)
    want = oracle.rx_batch(x, fs, rs, loop_bw=BW, timing_mode=TIMING_FIXED, fixed_index=3)
    buf = torch.full((F, L + extra, 2), float("nan"), dtype=torch.float32, device="cuda")
    buf[:, :L] = torch.from_numpy(x).cuda()
    sym = torch.zeros((F, m.nsym), dtype=torch.uint8, device="cuda")
    fr = torch.zeros((F,), dtype=torch.float32, device="cuda")
    ph = torch.zeros_like(fr)
    m.rx_batch_raw(buf, F, sym, fr, ph, pitch=L + extra)
    m.sync()
    if kernel == "lean":
        assert m.last_kernel() == "rx_lean_kernel"
    assert_batch_equal(dict(sym=sym, freq=fr, phase=ph), want, keys=("sym", "phase", "freq"))
    with pytest.raises(qpsk_amd.QpskError, match="frame_pitch"):
        m.rx_batch_raw(buf, F, sym, fr, ph, pitch=L - 2)
    # the estimating timing modes read the same pitched frames (fused scan kernel where the shape allows, else rrc_fir + scan; FFT)
    from oracle.pyoracle import TIMING_FFT
    for mode in (TIMING_HIST, TIMING_FFT) if cyc == 8 else (TIMING_HIST,):
        if mode == TIMING_FFT and L < 1024:
            continue
        mt = modem(fs=fs, rs=rs, frame_size=L, timing_mode=mode)
        if kernel == "lean":
            mt.tune(pipe_v=3)
        n = min(F, 64)
        wt = oracle.rx_batch(x[:n], fs, rs, loop_bw=BW, timing_mode=mode)
        mt.rx_batch_raw(buf, n, sym, fr, ph, pitch=L + extra)
        mt.sync()
        assert_batch_equal(dict(sym=sym[:n], freq=fr[:n], phase=ph[:n]), wt, keys=("sym", "phase", "freq"))


def test_caller_stream_ordering_contract(oracle):
    """include/qpsk_hip.h, "Stream ordering": the library enqueues on the context's stream and nothing else orders it
    against the caller's other streams.  Here the context runs on a NON-default torch stream, buffers are produced and
    recycled by torch's stream-ordered allocator under that same stream (the contract kept), with allocations in between
    calls: results are the oracle's.  (Round 1's abort in bench.py was this contract broken: a private library stream
    wrote output buffers whose memory torch had recycled from tensors that queued torch kernels still had to read.)"""
    import torch
    fs, rs, L, F = 19200.0, 2400.0, 2048, 64
    m = modem(fs=fs, rs=rs, frame_size=L, timing_mode=TIMING_FIXED, fixed_index=6)
    xh, _ = make_frames(F, L, 8, m.taps, fs, base_seed=21, noise=0.02)
    want = oracle.rx_batch(xh, fs, rs, loop_bw=BW, timing_mode=TIMING_FIXED, fixed_index=6)
    side = torch.cuda.Stream()
    m.set_stream(side)
    with torch.cuda.stream(side):
        for it in range(4):
            x = torch.from_numpy(xh).cuda(non_blocking=True)          # produced on `side`
            junk = [torch.randn(1 << 20, device="cuda") for _ in range(3)]   # allocator traffic on `side`
            got = m.rx_batch(x)
            del junk, x                                                # recycled in `side` order: safe
            junk2 = torch.zeros(1 << 22, device="cuda")
            side.synchronize()
            assert_batch_equal(got, want, keys=("sym", "phase", "freq"))
            del junk2
    m.sync()
    m.set_stream(None)


# ------------------------------------------------------------------ config 3: what is pinned inside the FFT timing estimate
def test_fft_timing_internals_are_the_pinned_stages(oracle):
    """The FFT timing estimate has no reference counterpart (the reference never calls fft.c), but everything below
    its final argmax IS reference code: its 512 filtered samples are rrc_fir() outputs (rrc_fir.c:17-30: equal to
    qpsk_rrc_fir_batch, which the reference fixtures pin) and its spectrum is fftn() (fft.c:110-120: equal to
    qpsk_fft_batch, pinned likewise).  Only "index = first argmax of Re(X_k e^{+j 2 pi i / CYCLES})" is this build's own
    definition, and that rule is checked against the oracle's restatement."""
    from oracle.pyoracle import TIMING_FFT
    fs, rs, L, F = 19200.0, 2400.0, 2048, 24
    m = modem(fs=fs, rs=rs, frame_size=L, timing_mode=TIMING_FFT)
    x, _ = make_frames(F, L, 8, m.taps, fs, offset_hz=35.0, base_seed=3, noise=0.1)
    x[5] = random_frames(1, L, seed=9)[0]
    idx, y, X = m.timing_fft(x, want_internals=True)
    m.sync()
    full = cpu(m.rrc_fir(x))                                   # (F, L, 2): the full-rate filter, fresh delay lines
    assert bits_equal(cpu(y), full[:, 128:128 + 512])
    p = (cpu(y)[..., 0].astype(np.float64) ** 2 + cpu(y)[..., 1].astype(np.float64) ** 2).astype(np.complex128)
    assert bits_equal(cpu(X), cpu(m.fft(p)))                   # same transform, same host-built twiddles
    Xh = cpu(X)[:, 512 // 8]
    import math                                                # libm's cos/sin, like the host table (numpy's may round differently)
    cand = np.stack([Xh.real * math.cos(TAU * i / 8.0) - Xh.imag * math.sin(TAU * i / 8.0) for i in range(8)], axis=1)
    assert np.array_equal(cpu(idx), np.argmax(cand, axis=1))   # the rule itself (first maximum)
    want = oracle.rx_batch(x, fs, rs, loop_bw=BW, timing_mode=TIMING_FFT)
    assert np.array_equal(cpu(idx), want["index"])
    # the kernel qpsk_rx_batch runs evaluates only the butterflies that bin depends on: same filter outputs, and the bin is
    # the full transform's bit for bit
    idx2, y2, xk = m.timing_fft_bin(x)
    m.sync()
    assert bits_equal(cpu(y2), cpu(y)) and bits_equal(cpu(xk), Xh) and np.array_equal(cpu(idx2), cpu(idx))


@pytest.mark.parametrize("fs,F", [(19200.0, 5000), (9600.0, 2300), (4800.0, 2049)])
def test_fft_timing_pruned_bin_many_frames(oracle, fs, F):
    """more frames than one wave per frame of a full grid (a wave then takes several, one after the other, the next one's
    samples in flight), a ragged last workgroup, CYCLES = 8, 4, 2 (bins 64, 128, 256), arbitrary (non-modem) samples, an input
    that is not 16-byte aligned: the pruned bin equals the full transform's, the index the oracle's"""
    import torch
    from oracle.pyoracle import TIMING_FFT
    rs, L = 2400.0, 640
    m = modem(fs=fs, rs=rs, frame_size=L, timing_mode=TIMING_FFT)
    if m.cycles == 2:      # rrc_make() (rrc_fir.c:32-76) yields NaN taps at two samples per symbol: the estimator is exercised on other taps
        m.set_taps(modem(fs=19200.0, rs=rs, frame_size=L).taps)
    x = random_frames(F, L, seed=int(fs))
    idx, y, X = m.timing_fft(x, want_internals=True)
    idx2, y2, xk = m.timing_fft_bin(x)
    m.sync()
    assert bits_equal(cpu(y2), cpu(y)) and bits_equal(cpu(xk), cpu(X)[:, 512 // m.cycles]) and np.array_equal(cpu(idx2), cpu(idx))
    pick = np.unique(np.concatenate([np.arange(0, F, 97), [1, F - 1]]))
    taps = m.taps
    for f in pick:
        assert int(cpu(idx2)[f]) == oracle.timing_fft_index(taps, x[f], m.cycles), f
    # unaligned input (frames start 8 bytes off a 16-byte boundary)
    buf = torch.zeros((F * L + 1, 2), dtype=torch.float32, device="cuda")
    buf[1:] = torch.from_numpy(x).cuda().reshape(-1, 2)
    idx3, y3, xk3 = m.timing_fft_bin(buf[1:].reshape(F, L, 2))
    m.sync()
    assert bits_equal(cpu(xk3), cpu(xk)) and np.array_equal(cpu(idx3), cpu(idx2))


@pytest.mark.parametrize("F", [4096, 3500, 3329])
def test_fft_timing_inside_the_receive_launch(oracle, F):
    """BASELINE config 3's shape (batches that fill rx_fused_pipe_kernel's 16-frame workgroups) runs the FFT estimate INSIDE the
    receive launch: same indices and same bits as the estimator launched in front (QPSK_FFT_FUSED = 0) and as the oracle, with
    per-frame timing offsets that differ (frames delayed by 0..7 samples), a ragged last workgroup, and a non-modem frame"""
    import torch
    from oracle.pyoracle import TIMING_FFT
    fs, rs, L = 19200.0, 2400.0, 1024
    m = modem(fs=fs, rs=rs, frame_size=L, timing_mode=TIMING_FFT)
    x, _ = make_frames(F, L + 8, 8, m.taps, fs, offset_hz=30.0, base_seed=11, noise=0.05)
    x = np.stack([x[f, (f % 8):(f % 8) + L] for f in range(F)])          # every decimation offset occurs
    x[7] = random_frames(1, L, seed=4)[0]
    got = m.rx_batch(x, want_costas=True)
    m.sync()
    assert m.last_kernel() == "rx_fused_pipe_kernel (FFT timing estimate inside the launch)"
    m2 = modem(fs=fs, rs=rs, frame_size=L, timing_mode=TIMING_FFT)
    m2.tune(fft_fused=0)
    sep = m2.rx_batch(x, want_costas=True)
    m2.sync()
    assert m2.last_kernel() == "rx_fused_pipe_kernel"
    for k in ("sym", "phase", "freq", "index", "hz", "costas"):
        assert bits_equal(cpu(got[k]), cpu(sep[k])), k
    idx = cpu(got["index"])
    assert len(np.unique(idx)) == 8
    pick = np.unique(np.concatenate([np.arange(0, F, 61), [7, F - 1]]))
    want = oracle.rx_batch(x[pick], fs, rs, loop_bw=BW, timing_mode=TIMING_FFT, want_costas=True)
    for k in ("sym", "phase", "freq", "index", "hz", "costas"):
        assert bits_equal(cpu(got[k])[pick], want[k]), k
    # without an index array the kernel leaves none (the path bench.py times)
    sym = torch.zeros((F, m.nsym), dtype=torch.uint8, device="cuda")
    fr = torch.zeros((F,), dtype=torch.float32, device="cuda")
    ph = torch.zeros_like(fr)
    m.rx_batch_raw(torch.from_numpy(x).cuda(), F, sym, fr, ph)
    m.sync()
    assert bits_equal(cpu(sym), cpu(got["sym"])) and bits_equal(cpu(fr), cpu(got["freq"]))


def test_full_size_config3_properties(oracle):
    """BASELINE config 3 at full size (4096 x 16384 with the FFT timing estimate in front): (a) every clean frame's
    estimate is the eye centre 126 mod 8 = 6, (b) so the whole batch equals the fixed-index batch bit for bit, (c) a spread
    sample of frames equals the oracle's FFT-timing result, (d) every loop ends on the +50 Hz offset"""
    import torch
    import bench
    from oracle.pyoracle import TIMING_FFT
    fs, rs, L, F = bench.FS, bench.RS, 16384, 4096
    m = modem(fs=fs, rs=rs, frame_size=L, timing_mode=TIMING_FFT)
    x = bench.synth_frames_gpu(torch, torch.device("cuda", 0), F, m.taps, seed=5)
    a = m.rx_batch(x)
    m.sync()
    ix = int(cpu(a["index"])[0])
    assert np.all(cpu(a["index"]) == ix)
    mf = modem(fs=fs, rs=rs, frame_size=L, timing_mode=TIMING_FIXED, fixed_index=ix)
    b = mf.rx_batch(x)
    mf.sync()
    for k in ("sym", "phase", "freq", "hz"):
        assert bits_equal(cpu(a[k]), cpu(b[k])), k
    pick = np.unique(np.concatenate([np.arange(0, F, 257), [1, F - 1]]))
    want = oracle.rx_batch(x[torch.from_numpy(pick).cuda()].cpu().numpy(), fs, rs, loop_bw=BW, timing_mode=TIMING_FFT)
    for k in ("sym", "phase", "freq", "index"):
        assert bits_equal(cpu(a[k])[pick], want[k].astype(cpu(a[k]).dtype)), k
    assert np.all(np.abs(cpu(a["hz"]) - 50.0) < 2.0)


# ------------------------------------------------------------------ histogram timing without a filtered block in memory
@pytest.mark.parametrize("L,F", [(256, 5), (1024, 37), (16384, 33)])
def test_timing_scan_fused(oracle, L, F):
    """qpsk_timing_scan_batch (full-rate FIR + amplitude-histogram scan fused through LDS, what QPSK_TIMING_HIST runs
    on): index AND every histogram bin equal the two-kernel composition rrc_fir -> timing_hist (itself pinned to the
    reference's fixtures) and the oracle; frame counts ragged against the 16-frame workgroups"""
    fs, rs = 19200.0, 2400.0
    m = modem(fs=fs, rs=rs, frame_size=L)
    x, _ = make_frames(F, L, 8, m.taps, fs, noise=0.1, base_seed=L)
    x[0] = 0.0
    x[1] = random_frames(1, L, seed=5)[0]
    x[2] = np.abs(random_frames(1, L, seed=6)[0]) * np.linspace(0.01, 3.0, L)[:, None]     # the running max keeps moving
    if F > 4:
        x[3, L // 2:] = 0.0
    idx, hist = m.timing_scan(x)
    m.sync()
    idx2, hist2 = m.timing_hist(m.rrc_fir(x), want_hist=True)
    m.sync()
    assert np.array_equal(cpu(idx), cpu(idx2)) and np.array_equal(cpu(hist), cpu(hist2))
    taps = oracle.rrc_make(fs, rs, np.float32(.35))
    for f in range(F):
        y = x[f].copy()
        oracle.rrc_fir(taps, np.zeros((127, 2), np.float32), y)
        assert cpu(idx)[f] == oracle.timing_index(y, 8), f
    # the three-kernel path (QPSK_HIST_GENERIC = 1) and the fused one give the same batch results
    mh = modem(fs=fs, rs=rs, frame_size=L, timing_mode=TIMING_HIST)
    a = mh.rx_batch(x, want_costas=True)
    mh.sync()
    mh.tune(hist_generic=1)
    b = mh.rx_batch(x, want_costas=True)
    mh.sync()
    for k in ("sym", "costas", "phase", "freq", "index", "hz"):
        assert bits_equal(cpu(a[k]), cpu(b[k])), k
    assert np.array_equal(cpu(a["index"]), cpu(idx))


@pytest.mark.parametrize("block", [0, 1])
def test_streams_stretches_of_zero_symbols(oracle, block):
    """Symbols that are exactly (+0, +0) -- every stream's first block after qpsk_streams_reset(), a squelched input from its third
    silent block on -- trip the Costas instruction stream's exact-zero test in every group; the lanes concerned are excused from it
    for the stretch of zeros they have ahead (costas_asm.h `ign`, rx_fused.hip costas_wave, streamblock.hip) instead of sending the
    whole workgroup through the C++ step.  Loop states that meet such a stretch: zeros of both signs in phase and frequency (where
    signed zeros decide: those stay with the C++ step), phases in every quadrant at rest, moving loops that wrap, a frequency at its
    limit, denormal frequencies; then squelch and signal again.  Every block of every stream against the oracle's modem; block = 1:
    the one-launch-per-block kernel, 0: the loop kernel of the composition."""
    fs, rs, L, S = 19200.0, 2400.0, 1024, 40
    m = modem(fs=fs, rs=rs, frame_size=L)
    m.tune(stream_block=block)
    m.streams_reset(S, 1500.0)
    nz = np.float32(-0.0)
    states = [(0.0, 0.0), (nz, 0.0), (0.0, nz), (nz, nz), (1.0, 0.0), (2.5, 0.0), (-2.5, nz), (4.0, 0.0), (5.5, nz), (-4.0, 0.0), (-5.5, nz),
              (-1.0, nz), (0.3, 0.7), (-1.0, -0.9), (6.0, 0.99), (0.0, 1e-30), (3.0, 1.0), (-3.0, -1.0), (6.2831855, 0.0), (-6.2831855, nz),
              (1.5707964, 0.0), (3.1415927, nz), (0.0, 1e-42), (nz, -1e-42), (2.0, 0.01)]
    st = np.array([states[i % len(states)] for i in range(S)], np.float32)
    m._check(m.L.qpsk_streams_set_loop_state(m.h, st.ctypes.data_as(C.POINTER(C.c_float))))
    om = [oracle.modem(fs, rs, L, loop_bw=BW) for _ in range(S)]
    for s_, o in enumerate(om):
        o.set_mixer_hz(1500.0)
        o.s.loop.phase = float(st[s_, 0])
        o.s.loop.freq = float(st[s_, 1])
    rng = np.random.default_rng(77)
    squelched = set(range(0, S, 3))
    for k in range(7):
        pcm = (6000 * rng.standard_normal((S, L))).astype(np.int16)
        if 1 <= k <= 4:
            for s_ in squelched:
                pcm[s_] = 0
        o = m.streams_rx_pcm(pcm)
        m.sync()
        for s_ in range(S):
            om[s_].rx_pcm(pcm[s_])
            assert cpu(o["index"])[s_] == om[s_].index, (k, s_)
            assert bits_equal(cpu(o["sym"][s_]), om[s_].symbols), (k, s_)
            assert bits_equal(cpu(o["costas"][s_]), om[s_].costas_frame), (k, s_)
            assert cpu(o["phase"])[s_].tobytes() == om[s_].phase.tobytes() and cpu(o["freq"])[s_].tobytes() == om[s_].freq.tobytes(), (k, s_, st[s_])


def test_streams_leave_the_shared_carrier(oracle):
    """the streams' one carrier (a table of the block's phases, advanced a block ahead by spare waves of stream_scan_kernel + the loop
    kernel, or of stream_block_kernel) while the two kernels that use it alternate, then handed back to the per-stream mixer state when
    the kernels apart take over (mixer_kernel reads every stream's own state), then the one-launch kernel and stream_scan_kernel on
    per-stream state (its mixer wave), a reset, and the table again; every block against the oracle's modems"""
    fs, rs, L, S = 19200.0, 2400.0, 512, 21
    m = modem(fs=fs, rs=rs, frame_size=L)
    m.tune(stream_block=0)
    m.tune(stream_scan=1)
    rng = np.random.default_rng(99)
    for round_ in range(2):
        m.tune(stream_block=0)
        m.tune(stream_scan=1)
        m.streams_reset(S, 1350.0)
        om = [oracle.modem(fs, rs, L, loop_bw=BW) for _ in range(S)]
        for o in om:
            o.set_mixer_hz(1350.0)
        for k, (scan, block) in enumerate([(1, 0), (0, 1), (1, 0), (1, 0), (0, 1), (0, 0), (0, 0), (0, 1), (1, 0), (1, 0)]):
            m.tune(stream_scan=scan)
            m.tune(stream_block=block)
            pcm = (7000 * rng.standard_normal((S, L))).astype(np.int16)
            o = m.streams_rx_pcm(pcm)
            m.sync()
            want_kernel = "stream_block_kernel" if block else ("stream_scan_kernel + costas_pipe_kernel" if scan else "filter, timing, costas_pipe_kernel")
            assert m.last_kernel() == want_kernel
            for s_ in range(S):
                om[s_].rx_pcm(pcm[s_])
                assert cpu(o["index"])[s_] == om[s_].index, (round_, k, s_)
                assert bits_equal(cpu(o["sym"][s_]), om[s_].symbols) and bits_equal(cpu(o["costas"][s_]), om[s_].costas_frame), (round_, k, s_)
                assert cpu(o["phase"])[s_] == om[s_].phase and cpu(o["freq"])[s_] == om[s_].freq, (round_, k, s_)


@pytest.mark.parametrize("kind", ["pcm", "pcm_own_carrier", "cplx"])
@pytest.mark.parametrize("fixed", [None, 2, 6])
@pytest.mark.parametrize("L,S", [(256, 20), (1024, 33)])
def test_stream_scan_kernel_at_four_samples_per_symbol(oracle, L, S, fixed, kind):
    """stream_scan_kernel at CYCLES = 4, the reference's shipped rates (FS 9600 / RS 2400): a lane's 8 outputs are two symbols, the
    filtered block has four planes, the histogram keeps its 8 bins -- so the timing index (and a fixed one: 6) may be CYCLES or more
    and the pick reaches into the next symbol, past the block for the last one (0.0) -- block after block against the
    oracle's modems, PCM (the streams' one carrier / every stream's own) and complex input"""
    fs, rs = 9600.0, 2400.0
    kw = dict(timing_mode=TIMING_FIXED, fixed_index=fixed) if fixed is not None else {}
    m = modem(fs=fs, rs=rs, frame_size=L, **kw)
    m.tune(stream_block=0)
    m.tune(stream_scan=1)
    m.tune(stream_carrier=0 if kind == "pcm_own_carrier" else 1)
    m.streams_reset(S, 1500.0)
    om = [oracle.modem(fs, rs, L, loop_bw=BW, **kw) for _ in range(S)]
    for o in om:
        o.set_mixer_hz(1500.0)
    rng = np.random.default_rng(L + S + (fixed or 0))
    for k in range(5):
        if kind == "cplx":
            blk = rng.standard_normal((S, L, 2)).astype(np.float32)
            if k == 2:
                blk[1] = 0.0
            o = m.streams_rx_cplx(blk)
        else:
            blk = (6000 * rng.standard_normal((S, L))).astype(np.int16)
            if k == 2:
                blk[1] = 0
            o = m.streams_rx_pcm(blk)
        m.sync()
        assert m.last_kernel() == "stream_scan_kernel + costas_pipe_kernel"
        for s_ in range(S):
            (om[s_].rx_cplx if kind == "cplx" else om[s_].rx_pcm)(blk[s_])
            assert cpu(o["index"])[s_] == om[s_].index, (k, s_)
            assert bits_equal(cpu(o["sym"][s_]), om[s_].symbols) and bits_equal(cpu(o["costas"][s_]), om[s_].costas_frame), (k, s_)
            assert cpu(o["phase"])[s_] == om[s_].phase and cpu(o["freq"])[s_] == om[s_].freq, (k, s_)
