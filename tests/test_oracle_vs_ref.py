"""The oracle against the reference ITSELF (oracle/_ref = /root/reference compiled untouched, see
oracle/Makefile).  Only possible in the build container; skipped where oracle/_ref is absent (GPU box),
where tests/test_oracle_golden.py pins the same functions through the committed fixtures instead."""
import numpy as np
import pytest

from oracle.pyoracle import TAU, TIMING_HIST, Reference, ref_available
from sigutil import bits_equal, make_frames, random_frames

pytestmark = pytest.mark.skipif(not ref_available("c1small"), reason="oracle/_ref not built (no /root/reference here)")
BW = np.float32(TAU / 100.0)


def rrc_grid():
    """(fs, rs, alpha) sets for rrc_make(): samples per symbol from 1 up, roll-offs to 1.  Below FS/RS ~ 2.2 (alpha .35) the
    arguments of cosf/sinf (rrc_fir.c:46-49,62-64) pass 120 and glibc takes its large-argument reduction."""
    out = []
    for rs in (2400.0, 1200.0, 300.0):
        for spb in (1.0, 5.0 / 3.0, 2.0, 2.2, 2.5, 3.0, 10.0 / 3.0, 4.0, 5.0, 8.0, 16.0, 20.0, 160.0):
            for alpha in (.05, .1, .2, .25, .35, .5, .6, .75, .8, .9, 1.0):
                out.append((np.float32(rs * spb), np.float32(rs), np.float32(alpha)))
    return out


def test_rrc_make_matches_reference(oracle):
    """rrc_make() (rrc_fir.c:32-76): oracle == reference == the product's host routine, bit for bit, on every set -- round 4's
    oracle answered NaN taps where the reference's cosf/sinf arguments reach 120 (FS/RS <= 2.2 at alpha .35)."""
    import ctypes as C
    import qpsk_amd
    hip = C.CDLL(qpsk_amd.lib_path())          # host symbol only: no GPU call
    hip.qpsk_host_rrc_taps.argtypes = [C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float)]
    hip.qpsk_host_rrc_taps.restype = None
    ref = Reference("shipped")
    large = 0
    for fs, rs, alpha in rrc_grid():
        want = ref.taps(fs, rs, alpha)
        got = oracle.rrc_make(fs, rs, alpha)
        prod = np.zeros(127, np.float32)
        hip.qpsk_host_rrc_taps(fs, rs, alpha, prod.ctypes.data_as(C.POINTER(C.c_float)))
        assert np.isfinite(want).all(), (fs, rs, alpha)   # the reference's taps are finite on the whole grid
        assert bits_equal(got, want), ("oracle", fs, rs, alpha)
        assert bits_equal(prod, want), ("product", fs, rs, alpha)
        large += (1.0 + float(alpha)) * np.pi * 63.0 * float(rs) / float(fs) >= 120.0
    assert large >= 40          # the grid does reach the large-argument path


@pytest.mark.parametrize("name", ["shipped", "c1small", "c5small"])
def test_streaming_pcm_matches_reference(oracle, name):
    ref = Reference(name)
    ref.reset(BW, -1.0, 1.0, .35, 1550.0, 1500.0)
    rng = np.random.default_rng(100)
    L, N = ref.frame_size, ref.nsym
    nblk = 5
    bits = rng.integers(0, 2, size=2 * N * nblk).astype(np.int32)
    pcm = np.concatenate([ref.tx_symbols(bits[2 * k:2 * min(k + 256, N * nblk)]) for k in range(0, N * nblk, 256)])
    # the oracle's transmitter restatement (qpsk.c:225-285) produces the same PCM
    tx = oracle.tx(ref.fs, ref.rs, np.float32(.35), 1550.0)
    pcm_o = np.concatenate([tx.symbols(bits[2 * k:2 * min(k + 256, N * nblk)]) for k in range(0, N * nblk, 256)])
    assert bits_equal(pcm, pcm_o)
    m = oracle.modem(ref.fs, ref.rs, L, loop_bw=BW)
    m.set_mixer(ref.mixer)
    for k in range(nblk):
        blk = pcm[k * L:(k + 1) * L]
        ref.rx_pcm(blk)
        m.rx_pcm(blk)
        assert bits_equal(ref.input_frame, m.input_frame), "filtered block %d" % k
        last = 2 * N if ref.cycles >= 8 else 2 * N - 1  # Q5: the last pick is undefined in the reference at CYCLES<8
        assert bits_equal(ref.decimated[:last], m.decimated[:last])
        assert bits_equal(ref.costas_frame, m.costas_frame) and bits_equal(ref.symbols, m.symbols)
        assert ref.phase == m.phase and ref.freq == m.freq and ref.offset_hz == m.offset_hz
        assert bits_equal(ref.rx_filter, m.rx_filter) and bits_equal(ref.mixer, m.mixer)
        if ref.cycles < 8:
            d = ref.decimated
            d[2 * N - 1] = m.decimated[2 * N - 1]  # defined-as-zero value
            ref.set_decimated(d)


@pytest.mark.parametrize("name,kind", [("c1small", "modem"), ("c1small", "noise"), ("c5small", "modem"), ("c1", "modem")])
def test_independent_frames_match_reference(oracle, name, kind):
    ref = Reference(name)
    ref.reset()
    L = ref.frame_size
    F = 2 if name == "c1" else 6
    if kind == "modem":
        x, _ = make_frames(F, L, ref.cycles, ref.taps(), ref.fs, offset_hz=40.0, base_seed=7, noise=0.05)
    else:
        x = random_frames(F, L, seed=3, scale=0.7)
    o = oracle.rx_batch(x, ref.fs, ref.rs, loop_bw=BW, timing_mode=TIMING_HIST, want_costas=True)
    for f in range(F):
        r = ref.independent_frame(x[f], loop_bw=BW)
        assert bits_equal(r["sym"], o["sym"][f]) and bits_equal(r["costas"], o["costas"][f])
        assert r["phase"] == o["phase"][f] and r["freq"] == o["freq"][f] and r["hz"] == o["hz"][f]


def _same_with_nans(a, b):
    """bit-equal where finite; NaN where the other is NaN (a NaN's sign and payload depend on operand order, which the
    oracle's component-wise complex products and the compiler's __mulsc3 calls need not share)"""
    a, b = np.asarray(a, np.float32).ravel(), np.asarray(b, np.float32).ravel()
    na, nb = np.isnan(a), np.isnan(b)
    return a.shape == b.shape and np.array_equal(na, nb) and bits_equal(a[~na], b[~nb])


@pytest.mark.parametrize("where", [0, 700, 2047])
def test_nan_input_matches_reference(oracle, where):
    """A NaN sample: the reference's complex products go through __mulsc3 (qpsk.c:197, 114-118), whose
    recovery branch only acts on INFINITE operands, so for NaN inputs the oracle's component-wise arithmetic gives the
    reference's values: the same symbols, the same finite values before the NaN reaches the loop, NaN in the same places
    after.  (An infinite sample is not compared: the reference's phase_wrap() never returns from an infinite phase,
    costas_loop.c:61-67 -- the library fences both, tests/test_gpu_parity.py::test_nonfinite_input_is_an_error.)"""
    ref = Reference("c1small")
    ref.reset()
    L = ref.frame_size
    x, _ = make_frames(3, L, ref.cycles, ref.taps(), ref.fs, offset_hz=40.0, base_seed=11, noise=0.05)
    x[1, min(where, L - 1), 0] = np.float32("nan")
    o = oracle.rx_batch(x, ref.fs, ref.rs, loop_bw=BW, timing_mode=TIMING_HIST, want_costas=True)
    for f in range(3):
        r = ref.independent_frame(x[f], loop_bw=BW)
        assert bits_equal(r["sym"], o["sym"][f])
        assert _same_with_nans(r["costas"], o["costas"][f])
        assert _same_with_nans([r["phase"], r["freq"], r["hz"]], [o["phase"][f], o["freq"][f], o["hz"][f]])
    # the frame's last sample only reaches a decimated symbol when the timing offset is CYCLES - 1: no NaN anywhere otherwise
    assert (np.isnan(o["phase"][1]) or where >= L - 1) and not np.isnan(o["phase"][0]) and not np.isnan(o["phase"][2])


def test_loop_bandwidth_sweep_matches_reference(oracle):
    ref = Reference("c5small")
    ref.reset()
    x, _ = make_frames(2, ref.frame_size, ref.cycles, ref.taps(), ref.fs, offset_hz=20.0, base_seed=5)
    bws = [np.float32(TAU / d) for d in (100.0, 150.0, 200.0)]
    o = oracle.rx_batch_bw(x, ref.fs, ref.rs, bws, timing_mode=TIMING_HIST)
    for f in range(2):
        for b, bw in enumerate(bws):
            r = ref.independent_frame(x[f], loop_bw=bw)
            assert bits_equal(r["sym"], o["sym"][f, b]) and r["phase"] == o["phase"][f, b] and r["freq"] == o["freq"][f, b]


def test_costas_scalar_api_matches_reference(oracle):
    from oracle.pyoracle import Costas
    import ctypes as C
    ref = Reference("shipped")
    ref.reset()
    rng = np.random.default_rng(8)
    for bw, lo, hi in [(BW, -1.0, 1.0), (np.float32(TAU / 200), -0.2, 0.3), (np.float32(0.5), -5.0, 5.0)]:
        ref.lib.create_control_loop(bw, lo, hi)
        c = Costas()
        oracle.lib.qo_costas_create(C.byref(c), bw, lo, hi)
        for _ in range(200):
            e = np.float32(rng.standard_normal() * 3)
            ref.lib.advance_loop(e); ref.lib.phase_wrap(); ref.lib.frequency_limit()
            oracle.lib.qo_advance_loop(C.byref(c), e); oracle.lib.qo_phase_wrap(C.byref(c)); oracle.lib.qo_frequency_limit(C.byref(c))
            st = ref.costas_state
            mine = np.array([c.phase, c.freq, c.max_freq, c.min_freq, c.damping, c.loop_bw, c.alpha, c.beta], np.float32)
            assert bits_equal(st, mine)
        # setters, including the ones whose range checks are dead code (costas_loop.c:79-115)
        for name, v in [("set_alpha", 7.0), ("set_beta", -3.0), ("set_loop_bandwidth", -0.1), ("set_damping_factor", -1.0),
                        ("set_frequency", 99.0), ("set_frequency", -99.0), ("set_phase", 20.0), ("set_phase", -20.0)]:
            getattr(ref.lib, name)(v)
            getattr(oracle.lib, "qo_" + name)(C.byref(c), v)
            mine = np.array([c.phase, c.freq, c.max_freq, c.min_freq, c.damping, c.loop_bw, c.alpha, c.beta], np.float32)
            assert bits_equal(ref.costas_state, mine), name


def test_fft_matches_reference(oracle):
    ref = Reference("shipped")
    rng = np.random.default_rng(4)
    for n in (1, 2, 4, 16, 512, 1024):
        x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        assert bits_equal(ref.fftn(x), oracle.fftn(x)), n
        assert bits_equal(ref.ifftn(x), oracle.ifftn(x)), n


def test_bit_stages_match_reference(oracle):
    ref = Reference("shipped")
    rng = np.random.default_rng(6)
    for n in (1, 2, 8, 22, 40, 43):
        d = rng.integers(0, 256, size=n).astype(np.uint8)
        assert bits_equal(ref.interleave(d, 0), oracle.interleave(d, 0)), n
        assert bits_equal(ref.interleave(d, 1), oracle.interleave(d, 1)), n
        assert ref.crc16(d.tobytes()) == oracle.crc16(d.tobytes())
    s = rng.integers(0, 4, size=300).astype(np.uint8)
    assert bits_equal(ref.scramble_stream(s, 0), oracle.scramble_stream(s))


@pytest.mark.parametrize("name", ["shipped", "c1small"])
def test_zero_symbols_and_signed_zero_states_match_reference(oracle, name):
    """Exactly-zero symbols (the first block after start-up, silent PCM) meeting loop states where the sign of a zero decides
    (costas_loop.c:44-59: sgn(0) = -1, x + -0): the oracle against the reference itself, state loaded through the reference's own
    set_phase() / set_frequency().  The HIP path's handling of such stretches (costas_asm.h `ign`, costas_step_t's exact-zero branch) is
    tested against the oracle on the GPU (test_streams_stretches_of_zero_symbols); this pins the oracle there."""
    nz = np.float32(-0.0)
    states = [(0.0, 0.0), (nz, 0.0), (0.0, nz), (nz, nz), (1.0, 0.0), (2.5, nz), (-2.5, nz), (4.0, 0.0), (5.5, nz), (-4.0, 0.0), (-5.5, nz),
              (0.3, 0.7), (-1.0, -0.9), (6.0, 0.99), (0.0, 1e-30), (6.2831855, 0.0), (-6.2831855, nz), (1.5707964, 0.0), (3.1415927, nz),
              (nz, -1e-42)]
    rng = np.random.default_rng(5)
    for ph, fr in states:
        ref = Reference(name)
        ref.reset(BW, -1.0, 1.0, .35, 1550.0, 1500.0)
        ref.lib.set_phase(float(ph))
        ref.lib.set_frequency(float(fr))
        L = ref.frame_size
        m = oracle.modem(ref.fs, ref.rs, L, loop_bw=BW)
        m.set_mixer(ref.mixer)
        m.s.loop.phase = float(ph)
        m.s.loop.freq = float(fr)
        for k in range(6):
            blk = np.zeros(L, np.int16) if k in (0, 2, 3, 4) else (6000 * rng.standard_normal(L)).astype(np.int16)
            ref.rx_pcm(blk)
            m.rx_pcm(blk)
            assert bits_equal(ref.costas_frame, m.costas_frame) and bits_equal(ref.symbols, m.symbols), (ph, fr, k)
            assert ref.phase.tobytes() == m.phase.tobytes() and ref.freq.tobytes() == m.freq.tobytes(), (ph, fr, k)
            if ref.cycles < 8:      # SURVEY Q5, as in test_streaming_pcm_matches_reference: the reference's last pick may lie past its array
                d = ref.decimated
                d[2 * ref.nsym - 1] = m.decimated[2 * ref.nsym - 1]
                ref.set_decimated(d)
