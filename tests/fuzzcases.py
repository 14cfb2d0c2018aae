"""Differential fuzzing of the HIP path against the CPU oracle: randomly drawn configurations and inputs through the library's OWN
dispatch (no tuning keys set, so whichever kernel api.cpp picks for the shape is the one compared), every output bit for bit.

Used by tests/test_gpu_fuzz.py (a fixed handful of seeds in the suite) and tools/fuzz.py (long runs on the GPU box, log kept under
profiles/).  The oracle is the checker here, as everywhere under tests/ (oracle/ header: test infrastructure, never a product path).

A case is (kind, seed): everything about it is drawn from numpy's PCG64 seeded with the seed, so a failure is reproduced by
`python tools/fuzz.py --kind batch --seed N --cases 1`.
"""
import numpy as np

from oracle.pyoracle import TAU, TIMING_FFT, TIMING_FIXED, TIMING_HIST
from sigutil import bits_equal, make_frames

RATES = [(9600.0, 2400.0), (19200.0, 2400.0), (19200.0, 2400.0), (19200.0, 2400.0), (9600.0, 1200.0), (12000.0, 2400.0),
         (20000.0, 2400.0), (38400.0, 2400.0), (7200.0, 2400.0), (14400.0, 2400.0),
         (4800.0, 2400.0)]   # two samples per symbol: rrc_make()'s cosf/sinf arguments pass 120 (glibc's large-argument reduction)


def _pick(rng, seq):
    return seq[int(rng.integers(0, len(seq)))]


def _nsym(rng):
    k = rng.integers(0, 10)
    if k == 0:
        return int(rng.integers(1, 20))
    if k <= 4:
        return int(rng.integers(20, 300))
    if k <= 7:
        return int(_pick(rng, [64, 128, 256, 512, 1024, 2048]))
    return int(rng.integers(300, 2100))


def _stimulus(rng, F, L, cycles, taps, fs, seed):
    """modem frames at a random offset / level / noise, with a few odd frames mixed in"""
    level = float(10.0 ** rng.uniform(-3, 3)) if rng.integers(0, 3) == 0 else 1.0
    if rng.integers(0, 12) == 0:
        level = float(10.0 ** rng.uniform(-42, -36))        # subnormal products: no flush to zero anywhere on the path
    x, _ = make_frames(F, L, cycles, taps, fs, offset_hz=float(rng.uniform(-150, 150)), base_seed=seed, amplitude=level,
                       noise=float(rng.choice([0.0, 0.02, 0.3])) * level)
    for _ in range(int(rng.integers(0, 3))):
        f = int(rng.integers(0, F))
        k = rng.integers(0, 4)
        if k == 0:
            x[f] = 0.0
        elif k == 1:
            x[f] = (level * rng.standard_normal((L, 2))).astype(np.float32)
        elif k == 2:
            x[f, :, 1] = 0.0                                   # real-only: exact zeros into the detector
        else:
            x[f, int(rng.integers(0, L)):] = 0.0               # a frame that ends early
    return x


def batch_case(oracle, modem_factory, seed, max_samples=6_000_000):
    """one qpsk_rx_batch / _pitched / _bw call on a random configuration -> (description, list of mismatching keys)"""
    import torch
    rng = np.random.Generator(np.random.PCG64(seed))
    big = rng.integers(0, 8) == 0    # more than 16 frames per CU: the 32-frame workgroups (rx_lean_kernel / rx_pipe2_kernel), short frames
    fs, rs = _pick(rng, RATES)
    if big and rng.integers(0, 4):
        fs, rs = 19200.0, 2400.0
    cycles = int(fs / rs)
    alpha = float(np.float32(rng.uniform(0.2, 0.6))) if rng.integers(0, 3) == 0 else 0.35
    taps = oracle.rrc_make(fs, rs, np.float32(alpha))
    # the reference's rrc_make() is finite at every rate and roll-off drawn here (tests/test_oracle_vs_ref.py's grid); an oracle that
    # says otherwise is a broken checker, not a case to skip (round 4 skipped, and hid the oracle's hole above |x| = 120)
    assert np.all(np.isfinite(taps)), "oracle rrc_make(%g, %g, %g) is not finite" % (fs, rs, alpha)
    nsym = int(_pick(rng, [128, 128, 192, 256, 100, 37])) if big else _nsym(rng)
    L = nsym * cycles
    F = int(rng.integers(4097, 9000)) if big else int(_pick(rng, [rng.integers(1, 6), rng.integers(6, 80), rng.integers(80, 700)]))
    if not big and rng.integers(0, 7) == 0:
        # 4..16 frames per CU in whole workgroups: the shapes rx_lean_kernel takes in one launch since round 5 (16-frame workgroups,
        # LDS-DMA staging with a window per unit when every timing offset of a wave is even, registers otherwise)
        fs, rs, cycles = 19200.0, 2400.0, 8
        taps = oracle.rrc_make(fs, rs, np.float32(alpha))
        nsym = int(_pick(rng, [128, 128, 192, 256]))
        L = nsym * cycles
        G = 2 * int(rng.integers(2, 9))
        F = G * int(rng.integers(256 - 256 // G + 1, 257))
    F = max(1, min(F, (2 * max_samples if big else max_samples) // L))
    mode = int(_pick(rng, [TIMING_FIXED, TIMING_FIXED, TIMING_HIST, TIMING_HIST, TIMING_FFT]))
    if mode == TIMING_FFT and (L < 128 + 512 + 126 or cycles not in (2, 4, 8)):   # the library's FFT estimate: CYCLES 2, 4, 8
        mode = TIMING_HIST
    fixed = int(rng.integers(0, min(cycles, 8)))                                    # the reference's histograms have 8 bins (qpsk.c:130)
    bw = np.float32(TAU / 100.0) if rng.integers(0, 2) else np.float32(TAU / 100.0 * 10.0 ** rng.uniform(-1, 0.7))
    lim = float(_pick(rng, [1.0, 1.0, 0.05, 3.0]))
    call = _pick(rng, ["plain", "plain", "costas", "pitched", "bw"])
    if big and rng.integers(0, 2):
        call = "plain"
    x = _stimulus(rng, F, L, cycles, taps, fs, seed)
    kw = dict(rrc_alpha=alpha, min_freq=-lim, max_freq=lim, timing_mode=mode, fixed_index=fixed)
    m = modem_factory(fs=fs, rs=rs, frame_size=L, loop_bw=bw, **kw)
    desc = "seed %d: fs %g rs %g alpha %.3f L %d F %d mode %d index %d bw %.5f lim %g call %s" % (seed, fs, rs, alpha, L, F, mode, fixed, bw, lim, call)
    bad = []
    if call == "bw":
        bws = [np.float32(TAU / 100.0 * 10.0 ** rng.uniform(-1, 0.5)) for _ in range(int(rng.integers(2, 5)))]
        want = oracle.rx_batch_bw(x, fs, rs, bws, **kw)
        got = m.rx_batch_bw(x, bws)
        keys = ("sym", "freq", "phase", "index")
    elif call == "pitched":
        pitch = L + int(rng.integers(1, 40))
        xp = torch.zeros((F, pitch, 2), dtype=torch.float32, device=m.dev)
        xp[:, :L] = torch.from_numpy(x).to(m.dev)
        got = dict(sym=m.empty((F, m.nsym), torch.uint8), freq=m.empty((F,), torch.float32), phase=m.empty((F,), torch.float32),
                   index=m.empty((F,), torch.int32))
        m.rx_batch_raw(xp, F, got["sym"], got["freq"], got["phase"], pitch=pitch, index=got["index"])
        want = oracle.rx_batch(x, fs, rs, loop_bw=bw, **kw)
        keys = ("sym", "freq", "phase", "index")
    else:
        want = oracle.rx_batch(x, fs, rs, loop_bw=bw, want_costas=call == "costas", **kw)
        got = m.rx_batch(x, want_costas=call == "costas")
        keys = ("sym", "freq", "phase", "index", "hz") + (("costas",) if call == "costas" else ())
    m.sync()
    desc += " kernel " + m.last_kernel()
    for k in keys:
        g = got[k].cpu().numpy()
        if not bits_equal(g, want[k].astype(g.dtype)):
            bad.append(k)
    if mode == TIMING_HIST and call == "plain" and not bad:
        # round 6: a histogram-mode context's SECOND call may take the one-pass route (rx_hist_kernel on the first batch's majority index,
        # a fall-back pass over the frames it misses); forced here whatever the batch's mix of indices, where the shape allows it
        m.tune(hist_onepass=1)
        got = m.rx_batch(x)
        m.sync()
        desc += " | again: " + m.last_kernel().split(" (")[0]
        for k in keys:
            g = got[k].cpu().numpy()
            if not bits_equal(g, want[k].astype(g.dtype)):
                bad.append(k + " (second call)")
    m.close()
    return desc, bad


def streams_case(oracle, modem_factory, seed):
    """a few consecutive blocks of a random number of streams with carried state (qpsk_streams_rx_pcm / _rx_cplx) against one oracle
    modem per checked stream"""
    rng = np.random.Generator(np.random.PCG64(seed))
    fs, rs = _pick(rng, RATES)
    cycles = int(fs / rs)
    taps = oracle.rrc_make(fs, rs, np.float32(0.35))
    assert np.all(np.isfinite(taps)), "oracle rrc_make(%g, %g, .35) is not finite" % (fs, rs)
    nsym = _nsym(rng)
    many = rng.integers(0, 6) == 0      # thousands of streams, whole 256-sample tiles at CYCLES = 8, histogram timing: stream_scan_kernel
    if many:
        fs, rs, cycles = 19200.0, 2400.0, 8
        taps = oracle.rrc_make(fs, rs, np.float32(0.35))
        nsym = 32 * int(rng.integers(1, 4))
    L = nsym * cycles
    S = int(_pick(rng, [1, rng.integers(2, 9), rng.integers(9, 100), rng.integers(100, 2000), rng.integers(2560, 4200)]))
    if many:
        S = int(rng.integers(2560, 4300))
    S = max(1, min(S, 3_000_000 // L))
    pcm_in = bool(rng.integers(0, 2))
    mode = int(_pick(rng, [TIMING_HIST, TIMING_HIST, TIMING_FIXED, TIMING_FFT]))
    if many:
        mode = TIMING_HIST
    if mode == TIMING_FFT and (L < 128 + 512 + 126 or cycles not in (2, 4, 8)):
        mode = TIMING_HIST
    fixed = int(rng.integers(0, min(cycles, 8)))
    bw = np.float32(TAU / 100.0)
    hz = float(_pick(rng, [1500.0, 1500.0, 1100.0, 0.0]))
    nblocks = int(rng.integers(2, 5))
    m = modem_factory(fs=fs, rs=rs, frame_size=L, loop_bw=bw, timing_mode=mode, fixed_index=fixed)
    m.streams_reset(S, hz)
    check = sorted(set([0, S - 1, S // 2] + [int(v) for v in rng.integers(0, S, size=3)]))
    om = {s: oracle.modem(fs, rs, L, loop_bw=bw, timing_mode=mode, fixed_index=fixed) for s in check}
    for o in om.values():
        o.set_mixer_hz(hz)
    desc = "seed %d: fs %g rs %g L %d streams %d %s mode %d index %d mixer %g Hz blocks %d" % (
        seed, fs, rs, L, S, "pcm" if pcm_in else "cplx", mode, fixed, hz, nblocks)
    bad = []
    for k in range(nblocks):
        if pcm_in:
            n = np.arange(k * L, (k + 1) * L)
            base = 9000.0 * np.cos(2 * np.pi * (hz + 40.0) * n / fs)[None] * np.sign(rng.standard_normal((S, 1)))
            blk = (base + 3000.0 * rng.standard_normal((S, L))).astype(np.int16)
            if k == 1 and S > 1:
                blk[S // 2] = 0
            o = m.streams_rx_pcm(blk)
        else:
            blk = (rng.standard_normal((S, L, 2)) * float(_pick(rng, [1.0, 1e-3, 50.0]))).astype(np.float32)
            if k == 1 and S > 1:
                blk[S // 2] = 0.0
            o = m.streams_rx_cplx(blk)
        m.sync()
        for s, orc in om.items():
            (orc.rx_pcm if pcm_in else orc.rx_cplx)(blk[s])
            ok = (int(o["index"][s].item()) == orc.index and bits_equal(o["sym"][s].cpu().numpy(), orc.symbols)
                  and bits_equal(o["costas"][s].cpu().numpy(), orc.costas_frame)
                  and np.float32(o["phase"][s].item()).tobytes() == orc.phase.tobytes()
                  and np.float32(o["freq"][s].item()).tobytes() == orc.freq.tobytes())
            if not ok:
                bad.append("block %d stream %d" % (k, s))
    desc += " kernel " + m.last_kernel()
    m.close()
    return desc, bad


def _dibits(bits):
    b = np.asarray(bits).reshape(bits.shape[0], -1, 2)
    return ((b[:, :, 0] << 1) | b[:, :, 1]).astype(np.uint8)      # qpsk.c:270


def stages_case(oracle, modem_factory, seed):
    """one of the stage entry points on random shapes: rrc_fir with and without delay lines, the histogram estimate on its own, the
    Costas loop over given symbols from given states, the radix-2 FFT, the transmitter over a few ragged calls, the bit stages"""
    import ctypes as C
    import torch
    from oracle.pyoracle import Costas
    rng = np.random.Generator(np.random.PCG64(seed))
    fs, rs = _pick(rng, RATES)
    cycles = int(fs / rs)
    taps = oracle.rrc_make(fs, rs, np.float32(0.35))
    assert np.all(np.isfinite(taps)), "oracle rrc_make(%g, %g, .35) is not finite" % (fs, rs)
    what = _pick(rng, ["fir", "fir", "hist", "costas", "fft", "tx", "bits"])
    m = modem_factory(fs=fs, rs=rs, frame_size=cycles * 64, loop_bw=np.float32(TAU / 100.0))
    desc = "seed %d: fs %g rs %g stage %s" % (seed, fs, rs, what)
    bad = []
    if what == "fir":
        F, n = int(_pick(rng, [1, rng.integers(2, 40), rng.integers(40, 600)])), int(_pick(rng, [rng.integers(1, 127), rng.integers(127, 700), rng.integers(700, 9000)]))
        F = max(1, min(F, 2_000_000 // n))
        x = (rng.standard_normal((F, n, 2)) * float(_pick(rng, [1.0, 1e-4, 300.0]))).astype(np.float32)
        mem = (rng.standard_normal((F, 127, 2))).astype(np.float32) if rng.integers(0, 2) else None
        desc += " F %d n %d memory %s" % (F, n, mem is not None)
        dmem = torch.from_numpy(mem.copy()).to(m.dev) if mem is not None else None
        y = m.rrc_fir(x, dmem).cpu().numpy()
        m.sync()
        for f in sorted(set([0, F - 1, F // 2] + [int(v) for v in rng.integers(0, F, size=4)])):
            om = mem[f].copy() if mem is not None else np.zeros((127, 2), np.float32)
            s = x[f].copy()
            oracle.rrc_fir(taps, om, s)
            if not bits_equal(y[f], s):
                bad.append("y[%d]" % f)
            if mem is not None and not bits_equal(dmem[f].cpu().numpy(), om):
                bad.append("memory[%d]" % f)
    elif what == "hist":
        F, n = int(rng.integers(1, 300)), cycles * int(rng.integers(1, 600))
        y = (rng.standard_normal((F, n, 2)) * float(_pick(rng, [1.0, 1e-3]))).astype(np.float32)
        if rng.integers(0, 2):
            y[0] = 0.0
        desc += " F %d n %d" % (F, n)
        m2 = modem_factory(fs=fs, rs=rs, frame_size=n, loop_bw=np.float32(TAU / 100.0))
        idx, hist = m2.timing_hist(y, want_hist=True)
        m2.sync()
        idx, hist = idx.cpu().numpy(), hist.cpu().numpy()
        for f in range(F):
            wi, wh = oracle.timing_hist(y[f], cycles)
            if idx[f] != wi or not np.array_equal(hist[f], wh):
                bad.append("frame %d" % f)
        m2.close()
    elif what == "costas":
        F, N = int(rng.integers(1, 200)), int(_pick(rng, [rng.integers(1, 70), rng.integers(70, 700)]))
        d = (rng.standard_normal((F, N, 2)) * float(_pick(rng, [1.0, 1e-3, 40.0]))).astype(np.float32)
        for _ in range(3):      # stretches of exact zeros, one-component zeros
            f, a = int(rng.integers(0, F)), int(rng.integers(0, N))
            d[f, a:a + int(rng.integers(1, 200))] = 0.0
            d[int(rng.integers(0, F)), :, int(rng.integers(0, 2))] = 0.0
        st0 = np.stack([rng.uniform(-6.2, 6.2, F), rng.uniform(-1, 1, F)], -1).astype(np.float32)
        st0[0] = [_pick(rng, [0.0, -0.0, 3.0]), _pick(rng, [0.0, -0.0])]
        desc += " F %d N %d" % (F, N)
        st = torch.from_numpy(st0.copy()).to(m.dev)
        sym, z = m.costas(d, st)
        m.sync()
        sym, z, st = sym.cpu().numpy(), z.cpu().numpy(), st.cpu().numpy()
        bw = np.float32(TAU / 100.0)
        for f in sorted(set([0, F - 1] + [int(v) for v in rng.integers(0, F, size=6)])):
            c = Costas()
            oracle.lib.qo_costas_create(C.byref(c), bw, -1.0, 1.0)
            c.phase, c.freq = float(st0[f, 0]), float(st0[f, 1])
            zr, zi = C.c_float(), C.c_float()
            ws, wz = np.zeros(N, np.uint8), np.zeros((N, 2), np.float32)
            for i in range(N):
                ws[i] = oracle.lib.qo_costas_step(C.byref(c), float(d[f, i, 0]), float(d[f, i, 1]), C.byref(zr), C.byref(zi))
                wz[i] = (zr.value, zi.value)
            if not (bits_equal(sym[f], ws) and bits_equal(z[f], wz) and bits_equal(st[f], np.array([c.phase, c.freq], np.float32))):
                bad.append("frame %d" % f)
    elif what == "fft":
        n, B = 1 << int(rng.integers(0, 14)), int(rng.integers(1, 9))
        x = rng.standard_normal((B, n)) + 1j * rng.standard_normal((B, n))
        inv = bool(rng.integers(0, 2))
        desc += " n %d batch %d inverse %s" % (n, B, inv)
        got = m.fft(x, inverse=inv).cpu().numpy()
        m.sync()
        for b in range(B):
            if not bits_equal(got[b], (oracle.ifftn if inv else oracle.fftn)(x[b])):
                bad.append("row %d" % b)
    elif what == "tx":
        S = int(rng.integers(1, 60))
        hz = float(_pick(rng, [1550.0, 1500.0, 900.0]))
        m.tx_reset(S, hz)
        otx = [oracle.tx(fs, rs, np.float32(.35), hz) for _ in range(S)]
        desc += " transmitters %d at %g Hz" % (S, hz)
        for _ in range(int(rng.integers(1, 4))):
            nsym = int(_pick(rng, [1, rng.integers(2, 70), rng.integers(70, 1500)]))
            bits = rng.integers(0, 2, size=(S, 2 * nsym)).astype(np.int32)
            got = m.tx_symbols(_dibits(bits))["pcm"].cpu().numpy()
            m.sync()
            for s in range(S):
                if not np.array_equal(got[s], otx[s].symbols(bits[s])):
                    bad.append("call of %d symbols, transmitter %d" % (nsym, s))
    else:
        P, nb = int(rng.integers(1, 80)), int(rng.integers(1, 256))
        pk = rng.integers(0, 256, size=(P, nb)).astype(np.uint8)
        desc += " packets %d of %d bytes" % (P, nb)
        crc = m.crc16(pk)
        fwd, back = m.interleave(pk, 0).cpu().numpy(), m.interleave(pk, 1).cpu().numpy()
        syms = rng.integers(0, 4, size=(min(P, 4), int(rng.integers(1, 3000)))).astype(np.uint8)
        sc = m.scramble(syms).cpu().numpy()
        m.sync()
        for p in range(P):
            if int(crc[p]) != oracle.crc16(pk[p].tobytes()):
                bad.append("crc %d" % p)
            if not (bits_equal(fwd[p], oracle.interleave(pk[p], 0)) and bits_equal(back[p], oracle.interleave(pk[p], 1))):
                bad.append("interleave %d" % p)
        for p in range(syms.shape[0]):
            if not bits_equal(sc[p], oracle.scramble_stream(syms[p])):
                bad.append("scramble %d" % p)
    m.close()
    return desc + " kernel -", bad
