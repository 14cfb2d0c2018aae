#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE ITSELF (oracle/_ref, i.e. /root/reference compiled
untouched by oracle/Makefile).  Run in the build container only:

    make -C oracle ref && python tools/make_golden.py

The fixtures are data (inputs and the reference's outputs); they let the oracle be pinned on the GPU
box, where the reference does not exist.  Inputs are produced with the reference's own transmitter
(qpsk.c:225-285) where a waveform is needed, or with numpy PRNGs of fixed seed.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.pyoracle import TAU, Reference  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
LOOP_BW = np.float32(TAU / 100.0)


def infer_index(ref):
    """index is a local of rx_frame (qpsk.c:105): recover it from decimated_frame[N] == input_frame[index]."""
    filt, dec, N = ref.input_frame, ref.decimated, ref.nsym
    cands = [k for k in range(8) if all(
        (i * ref.cycles + k < ref.frame_size) and np.array_equal(dec[N + i].view(np.uint32), filt[i * ref.cycles + k].view(np.uint32))
        for i in range(min(N - 1, 16)))]
    assert len(cands) == 1, cands
    return cands[0]


def tx_pcm(ref, rng, nblocks):
    nsym_total = ref.nsym * nblocks
    bits = rng.integers(0, 2, size=2 * nsym_total).astype(np.int32)
    chunk = 256
    parts = [ref.tx_symbols(bits[2 * k:2 * min(k + chunk, nsym_total)]) for k in range(0, nsym_total, chunk)]
    return np.concatenate(parts), bits


def tx_complex(ref, rng, nblocks, offset_hz):
    """complex baseband at +offset_hz built from the reference's modulator (zero-stuff + RRC)."""
    nsym_total = ref.nsym * nblocks
    bits = rng.integers(0, 2, size=2 * nsym_total).astype(np.int32)
    chunk = 256
    bb = np.concatenate([ref.tx_baseband(bits[2 * k:2 * min(k + chunk, nsym_total)]) for k in range(0, nsym_total, chunk)])
    z = bb[:, 0].astype(np.float64) + 1j * bb[:, 1].astype(np.float64)
    z = z * np.exp(2j * np.pi * offset_hz * np.arange(z.size) / ref.fs)
    return np.stack([z.real, z.imag], -1).astype(np.float32)


def gen_taps():
    ref = Reference("shipped")
    cases = [(9600, 2400, .35), (19200, 2400, .35), (9600, 1200, .35), (19200, 2400, .5), (19200, 2400, .25),
             (9600, 2400, .5), (9600, 1200, 1.0), (48000, 300, .2), (8000, 2400, .35),
             # low samples per symbol: (1 + alpha) pi 63 RS/FS >= 120, glibc's large-argument cosf/sinf (rrc_fir.c:46-49,62-64)
             (4800, 2400, .35), (4800, 2400, 1.0), (4000, 2400, .35), (2400, 2400, .35), (2400, 2400, .9), (5280, 2400, .35),
             (7200, 2400, .9)]
    arr = np.stack([ref.taps(np.float32(a), np.float32(b), np.float32(c)) for a, b, c in cases])
    np.savez(os.path.join(OUT, "taps.npz"), cases=np.array(cases, np.float64), taps=arr)


def gen_stream_pcm(name, nblocks, seed):
    ref = Reference(name)
    ref.reset(LOOP_BW, -1.0, 1.0, .35, 1550.0, 1500.0)
    rng = np.random.default_rng(seed)
    pcm, _ = tx_pcm(ref, rng, nblocks)
    L, N = ref.frame_size, ref.nsym
    rec = dict(pcm=pcm, mixer0=ref.mixer, sym=[], costas=[], phase=[], freq=[], hz=[], index=[], mixer=[],
               rx_filter=[])
    for k in range(nblocks):
        ref.rx_pcm(pcm[k * L:(k + 1) * L])
        idx = infer_index(ref)
        if ref.cycles < 8 and (N - 1) * ref.cycles + idx >= L:
            # the reference reads past input_frame[] here (undefined); the build defines that
            # sample as 0.  Put the defined value where the reference's next call will pick it up.
            d = ref.decimated
            d[2 * N - 1] = 0.0
            ref.set_decimated(d)
        rec["sym"].append(ref.symbols); rec["costas"].append(ref.costas_frame)
        rec["phase"].append(ref.phase); rec["freq"].append(ref.freq); rec["hz"].append(ref.offset_hz)
        rec["index"].append(idx); rec["mixer"].append(ref.mixer); rec["rx_filter"].append(ref.rx_filter)
    np.savez_compressed(os.path.join(OUT, "stream_pcm_%s.npz" % name), fs=ref.fs, rs=ref.rs, frame_size=L,
                        loop_bw=LOOP_BW, **{k: np.array(v) for k, v in rec.items()})


def gen_stream_cplx(name, nblocks, seed, offset_hz=50.0):
    ref = Reference(name)
    ref.reset(LOOP_BW, -1.0, 1.0, .35, 1550.0, 1500.0)
    rng = np.random.default_rng(seed)
    x = tx_complex(ref, rng, nblocks, offset_hz)
    L = ref.frame_size
    rec = dict(x=x, sym=[], costas=[], phase=[], freq=[], hz=[], index=[], filtered_head=[])
    for k in range(nblocks):
        ref.rx_cplx(x[k * L:(k + 1) * L])
        rec["sym"].append(ref.symbols); rec["costas"].append(ref.costas_frame)
        rec["phase"].append(ref.phase); rec["freq"].append(ref.freq); rec["hz"].append(ref.offset_hz)
        rec["index"].append(infer_index(ref)); rec["filtered_head"].append(ref.input_frame[:256])
    np.savez_compressed(os.path.join(OUT, "stream_cplx_%s.npz" % name), fs=ref.fs, rs=ref.rs, frame_size=L,
                        loop_bw=LOOP_BW, **{k: np.array(v) for k, v in rec.items()})


def gen_independent(name, nframes, seed, loop_bw=LOOP_BW, tag=""):
    """SURVEY 8(c) independent-frame pin: fresh state; rx_frame(frame); rx_frame(zeros)."""
    ref = Reference(name)
    rng = np.random.default_rng(seed)
    L = ref.frame_size
    rec = dict(x=[], sym=[], costas=[], phase=[], freq=[], hz=[], index=[])
    for f in range(nframes):
        ref.reset(loop_bw, -1.0, 1.0, .35, 1550.0, 1500.0)
        x = tx_complex(ref, rng, 1, 50.0 + 7.0 * f)
        if f == nframes - 1:
            x = (0.5 * rng.standard_normal((L, 2))).astype(np.float32)  # not a modem signal at all
        ref.reset(loop_bw, -1.0, 1.0, .35, 1550.0, 1500.0)
        ref.rx_cplx(x)
        idx = infer_index(ref)
        ref.rx_cplx(np.zeros((L, 2), np.float32))
        rec["x"].append(x); rec["sym"].append(ref.symbols); rec["costas"].append(ref.costas_frame)
        rec["phase"].append(ref.phase); rec["freq"].append(ref.freq); rec["hz"].append(ref.offset_hz)
        rec["index"].append(idx)
    np.savez_compressed(os.path.join(OUT, "independent_%s%s.npz" % (name, tag)), fs=ref.fs, rs=ref.rs,
                        frame_size=L, loop_bw=np.float32(loop_bw), **{k: np.array(v) for k, v in rec.items()})


def gen_fir():
    ref = Reference("shipped")
    ref.reset()
    rng = np.random.default_rng(5)
    mem = rng.standard_normal((127, 2)).astype(np.float32)
    rec = dict(taps=ref.taps(), mem0=mem.copy(), x=[], y=[], mem=[])
    for n in (1, 5, 126, 127, 128, 1000):
        x = rng.standard_normal((n, 2)).astype(np.float32)
        y = x.copy()
        ref.rrc_fir(mem, y)
        rec["x"].append(x); rec["y"].append(y); rec["mem"].append(mem.copy())
    np.savez_compressed(os.path.join(OUT, "fir.npz"), taps=rec["taps"], mem0=rec["mem0"],
                        **{"x%d" % i: v for i, v in enumerate(rec["x"])},
                        **{"y%d" % i: v for i, v in enumerate(rec["y"])},
                        **{"m%d" % i: v for i, v in enumerate(rec["mem"])})


def gen_tx(name, nblocks, nsym, seed):
    """The reference's transmitter (qpsk_packet_mod -> tx_frame, qpsk.c:225-285) over consecutive blocks with
    its carried state (tx_filter, fbb_tx_phase); the shaped baseband of the same bits from a second run."""
    ref = Reference(name)
    rng = np.random.default_rng(seed)
    bits = rng.integers(0, 2, size=(nblocks, 2 * nsym)).astype(np.int32)
    ref.reset(LOOP_BW, -1.0, 1.0, .35, 1550.0, 1500.0)
    pcm = np.stack([ref.tx_symbols(b) for b in bits])
    ref.reset(LOOP_BW, -1.0, 1.0, .35, 1550.0, 1500.0)
    bb = np.stack([ref.tx_baseband(b) for b in bits])
    np.savez_compressed(os.path.join(OUT, "tx_%s.npz" % name), fs=ref.fs, rs=ref.rs, tx_hz=1550.0, bits=bits.astype(np.uint8),
                        pcm=pcm, baseband=bb)


def gen_fft_bits():
    ref = Reference("shipped")
    rng = np.random.default_rng(9)
    d = {}
    for n in (2, 8, 64, 512, 2048):
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex128)
        d["x%d" % n] = x
        d["fft%d" % n] = ref.fftn(x)
        d["ifft%d" % n] = ref.ifftn(x)
    d["ramp8"] = ref.fftn(np.arange(1, 9).astype(np.complex128))
    delta = np.zeros(512, np.complex128); delta[0] = 1
    d["delta512"] = ref.fftn(delta)
    msgs = [b"123456789", b"", b"\x00", bytes(range(256))]
    d["crc"] = np.array([ref.crc16(m) for m in msgs], np.uint16)
    data = np.array([0b10101010] * 4 + [0] * 4, np.uint8)
    d["il_in"] = data
    d["il_out"] = ref.interleave(data, 0)
    d["il_back"] = ref.interleave(d["il_out"], 1)
    r22 = rng.integers(0, 256, size=22).astype(np.uint8)
    d["il22_in"] = r22
    d["il22_out"] = ref.interleave(r22, 0)
    syms = rng.integers(0, 4, size=512).astype(np.uint8)
    d["scr_in"] = syms
    d["scr_out"] = ref.scramble_stream(syms, 0)
    # slicer and detector spot values
    pts = rng.standard_normal((256, 2)).astype(np.float32)
    pts[:8] = [[1, 0], [0, 1], [-1, 0], [0, -1], [0, 0], [1, 1], [-1, 1], [1e-30, -1e-30]]
    d["pts"] = pts
    d["demod"] = np.array([ref.demod(float(a), float(b)) for a, b in pts], np.uint8)
    d["detector"] = np.array([ref.phase_detector(float(a), float(b)) for a, b in pts], np.float32)
    np.savez_compressed(os.path.join(OUT, "fft_bits.npz"), **d)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    only = set(sys.argv[1:])   # e.g. "tx" regenerates the transmitter fixtures alone; no argument = everything

    def want(group):
        return not only or group in only

    if want("taps"):
        gen_taps()
    if want("fir"):
        gen_fir()
    if want("stream"):
        gen_stream_pcm("shipped", 8, 11)
        gen_stream_pcm("c1small", 6, 12)
        gen_stream_cplx("c1small", 6, 13)
        gen_stream_cplx("c5small", 3, 14)
    if want("independent"):
        gen_independent("c1small", 8, 21)
        gen_independent("c1", 2, 22)
        gen_independent("c5small", 3, 23, loop_bw=np.float32(TAU / 200.0), tag="_bw200")
    if want("fft_bits"):
        gen_fft_bits()
    if want("tx"):
        gen_tx("shipped", 6, 100, 31)
        gen_tx("c1small", 4, 333, 32)
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("golden fixtures written to", OUT, "total bytes", tot)
