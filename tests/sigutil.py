"""Synthetic QPSK frames of the shape SURVEY.md 8(d) names (numpy, CPU).

Per frame: dibits from a fixed integer PRNG seeded base_seed ^ frame -> Gray map (reference
qpsk.c:58-63,270,278-279) -> zero-stuff x CYCLES (qpsk.c:232-238) -> transmit RRC with the receive
taps (qpsk.c:243,308) -> rotation by +offset_hz (the reference tests +50 Hz: qpsk.c:320 vs 342)
-> complex float32.  This is stimulus, not a parity object: the same bytes go to the GPU path and to
the oracle, so it does not need to match the reference's transmitter bit for bit.
"""
import numpy as np

CONSTELLATION = np.array([1 + 0j, 0 + 1j, 0 - 1j, -1 + 0j], np.complex64)  # qpsk.c:58-63


def splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


def frame_symbols(nsym, seed):
    """nsym symbol indices 0..3, deterministic in (seed)."""
    rng = np.random.Generator(np.random.PCG64(splitmix64(int(seed))))
    return rng.integers(0, 4, size=nsym, dtype=np.uint8)


def make_frames(nframes, frame_size, cycles, taps, fs, offset_hz=50.0, base_seed=1234, amplitude=1.0,
                noise=0.0, first_frame=0):
    """-> (nframes, frame_size, 2) float32 and the transmitted symbol indices (nframes, nsym)."""
    nsym = frame_size // cycles
    taps64 = np.asarray(taps, np.float64)
    n = np.arange(frame_size, dtype=np.float64)
    rot = np.exp(2j * np.pi * offset_hz * n / fs)
    out = np.zeros((nframes, frame_size, 2), np.float32)
    tx = np.zeros((nframes, nsym), np.uint8)
    for f in range(nframes):
        s = frame_symbols(nsym, base_seed ^ (first_frame + f))
        tx[f] = s
        up = np.zeros(frame_size, np.complex128)
        up[::cycles] = CONSTELLATION[s]
        y = np.convolve(up, taps64)[:frame_size] * 1.85  # rrc_fir's second GAIN (rrc_fir.c:28)
        y = y * rot * amplitude
        if noise > 0.0:
            rng = np.random.Generator(np.random.PCG64(splitmix64((base_seed ^ (first_frame + f)) + 77)))
            y = y + noise * (rng.standard_normal(frame_size) + 1j * rng.standard_normal(frame_size))
        out[f, :, 0] = y.real.astype(np.float32)
        out[f, :, 1] = y.imag.astype(np.float32)
    return out, tx


def random_frames(nframes, frame_size, seed=0, scale=1.0):
    rng = np.random.default_rng(seed)
    return (scale * rng.standard_normal((nframes, frame_size, 2))).astype(np.float32)


def bits_equal(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8))
