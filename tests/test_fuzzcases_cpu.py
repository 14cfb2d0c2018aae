"""The fuzzer's case generator (tests/fuzzcases.py) exercised on the CPU: the oracle stands in for the library behind the Modem
interface the generator drives, so a case compares the oracle with itself -- what is tested is the generator (every call shape it can
draw: plain, costas_frame[], pitched, several loop bandwidths; argument domains the library accepts), not the path.  The GPU suite runs
the same generator against libqpsk_hip (tests/test_gpu_fuzz.py)."""
import numpy as np
import torch

import fuzzcases


class _OracleAsModem:
    def __init__(self, orc, fs, rs, frame_size, loop_bw, rrc_alpha=0.35, min_freq=-1.0, max_freq=1.0, timing_mode=0, fixed_index=0):
        self.o = orc
        self.kw = dict(rrc_alpha=rrc_alpha, min_freq=min_freq, max_freq=max_freq, timing_mode=timing_mode, fixed_index=fixed_index)
        self.fs, self.rs, self.L, self.bw = fs, rs, frame_size, loop_bw
        self.dev = torch.device("cpu")
        self.nsym = frame_size // int(fs / rs)
        assert 0 <= fixed_index < 8 and frame_size % int(fs / rs) == 0      # what qpsk_ctx_create() accepts

    def empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype)

    def tune(self, **kw):      # kernel-geometry keys mean nothing to the oracle
        pass

    def sync(self):
        pass

    def close(self):
        pass

    def last_kernel(self):
        return "oracle"

    def rx_batch(self, x, want_costas=False):
        o = self.o.rx_batch(x, self.fs, self.rs, loop_bw=self.bw, want_costas=want_costas, **self.kw)
        return {k: torch.from_numpy(v) for k, v in o.items()}

    def rx_batch_bw(self, x, bws):
        o = self.o.rx_batch_bw(x, self.fs, self.rs, bws, **self.kw)
        return {k: torch.from_numpy(v) for k, v in o.items()}

    def rx_batch_raw(self, xp, F, sym, freq, phase, pitch=0, index=None):
        o = self.o.rx_batch(xp[:, :self.L].contiguous().numpy(), self.fs, self.rs, loop_bw=self.bw, **self.kw)
        sym.copy_(torch.from_numpy(o["sym"]))
        freq.copy_(torch.from_numpy(o["freq"]))
        phase.copy_(torch.from_numpy(o["phase"]))
        index.copy_(torch.from_numpy(o["index"]))


def test_batch_generator_draws_every_call_shape(oracle):
    calls, modes = set(), set()
    for seed in range(500, 560):
        desc, bad = fuzzcases.batch_case(oracle, lambda **kw: _OracleAsModem(oracle, **kw), seed, max_samples=400_000)
        if bad is None:
            continue
        assert bad == [], desc
        calls.add(desc.split(" call ")[1].split()[0])
        modes.add(int(desc.split(" mode ")[1].split()[0]))
    assert calls == {"plain", "costas", "pitched", "bw"} and modes == {0, 1, 2}
