#!/usr/bin/env python3
"""Measurement helper (GPU box): time qpsk_rx_batch on the bench workload under several tuning settings in ONE
process: one context per setting (qpsk_ctx_create reads the QPSK_* variables once), launches interleaved.

    python tools/sweep.py [--frames 4096] "QPSK_PIPE_V=1" "QPSK_PIPE_V=2 QPSK_PIPE_NF=6" ...
An empty string is the default configuration.  Interleaved rounds, median of per-launch HIP-event times.
The result-changing ablation bits of QPSK_PIPE_DBG (1, 2) only exist in the measurement build:
    QPSK_HIP_LIB=qpsk_amd/libqpsk_hip_prof.so python tools/sweep.py "QPSK_PIPE_DBG=1" ...
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=4096)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--per-round", type=int, default=4)
    ap.add_argument("--timing", choices=["fixed", "hist", "fft"], default="fixed")
    ap.add_argument("--offset-hz", type=float, default=50.0, help="carrier offset of the synthetic frames (sets the 2 pi wrap rate)")
    ap.add_argument("configs", nargs="*", default=[""])
    args = ap.parse_args()
    import torch
    import qpsk_amd
    dev = torch.device("cuda", 0)
    F = args.frames
    mode = {"fixed": qpsk_amd.TIMING_FIXED, "hist": qpsk_amd.TIMING_HIST, "fft": qpsk_amd.TIMING_FFT}[args.timing]
    keys = set()
    for c in args.configs:
        for kv in c.split():
            keys.add(kv.split("=")[0])
    modems = {}
    for c in args.configs:
        for k in keys:
            os.environ.pop(k, None)
        for kv in c.split():
            k, v = kv.split("=")
            os.environ[k] = str(int(v, 0))      # the library reads decimal integers
        modems[c] = qpsk_amd.Modem(fs=bench.FS, rs=bench.RS, frame_size=bench.L, timing_mode=mode, fixed_index=bench.FIXED_INDEX)
    for k in keys:
        os.environ.pop(k, None)
    m = modems[args.configs[0]]
    x = bench.synth_frames_gpu(torch, dev, F, m.taps, seed=1000, offset_hz=args.offset_hz)
    sym = torch.empty((F, m.nsym), dtype=torch.uint8, device=dev)
    freq = torch.empty((F,), dtype=torch.float32, device=dev)
    phase = torch.empty((F,), dtype=torch.float32, device=dev)
    times = {c: [] for c in args.configs}
    for r in range(args.rounds + 1):
        for c in args.configs:
            m = modems[c]
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.per_round)]
            for a, b in evs:
                a.record()
                m.rx_batch_raw(x, F, sym, freq, phase)
                b.record()
            torch.cuda.synchronize()
            if r > 0:
                times[c] += [a.elapsed_time(b) for a, b in evs]
    nbytes = 8.0 * F * bench.L
    ref = None
    for c in args.configs:
        t = np.array(times[c])
        modems[c].rx_batch_raw(x, F, sym, freq, phase)      # every setting must give the same bits (layouts, not results)
        torch.cuda.synchronize()
        got = (sym.clone(), freq.clone(), phase.clone())
        if ref is None:
            ref = got
        same = all(torch.equal(a.view(torch.uint8), b.view(torch.uint8)) for a, b in zip(got, ref))
        print("%-40s median %.4f ms  min %.4f ms  -> %.0f GB/s (%.1f%% of 8 TB/s)  %s" % (
            c or "(default)", np.median(t), t.min(), nbytes / np.median(t) / 1e6, nbytes / np.median(t) / 1e6 / 80.0,
            "same bits as the first setting" if same else "RESULT DIFFERS FROM THE FIRST SETTING"))


if __name__ == "__main__":
    main()
