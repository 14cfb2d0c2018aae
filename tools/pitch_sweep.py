#!/usr/bin/env python3
"""Measurement helper (GPU box): qpsk_rx_batch_pitched on the bench signal at several frame pitches, interleaved in one process.

    python tools/pitch_sweep.py [--frames 8192] [--dbg N] [extra_samples ...]      (default 0 32 128 512 1024 2080)
With the measurement build (QPSK_HIP_LIB=...prof.so) --dbg 49153 times the kernel's memory-side floor."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
import qpsk_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=8192)
ap.add_argument("--dbg", type=lambda v: int(v, 0), default=0)
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("extra", nargs="*", type=int, default=[0, 32, 128, 512, 1024, 2080])
args = ap.parse_args()
dev = torch.device("cuda", 0)
F, L = args.frames, bench.L
m = qpsk_amd.Modem(fs=bench.FS, rs=bench.RS, frame_size=L, timing_mode=qpsk_amd.TIMING_FIXED, fixed_index=bench.FIXED_INDEX)
if args.dbg:
    m.tune(pipe_dbg=args.dbg)
x = bench.synth_frames_gpu(torch, dev, F, m.taps, seed=1000)
bufs = {}
for ex in args.extra:
    b = torch.zeros((F, L + ex, 2), dtype=torch.float32, device=dev)
    b[:, :L] = x
    bufs[ex] = b
sym = torch.empty((F, m.nsym), dtype=torch.uint8, device=dev)
fr = torch.empty((F,), dtype=torch.float32, device=dev)
ph = torch.empty_like(fr)
times = {ex: [] for ex in args.extra}
for r in range(args.rounds + 1):
    for ex in args.extra:
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for a, b in evs:
            a.record()
            m.rx_batch_raw(bufs[ex], F, sym, fr, ph, pitch=L + ex)
            b.record()
        torch.cuda.synchronize()
        if r:
            times[ex] += [a.elapsed_time(b) for a, b in evs]
ref = None
for ex in args.extra:
    t = np.array(times[ex])
    m.rx_batch_raw(bufs[ex], F, sym, fr, ph, pitch=L + ex)
    torch.cuda.synchronize()
    got = (sym.clone(), fr.clone(), ph.clone())
    ref = ref or got
    same = all(torch.equal(a.view(torch.uint8), b.view(torch.uint8)) for a, b in zip(got, ref))
    print("%d frames, pitch %d samples (%d B), %s: median %.4f ms  min %.4f  -> %.0f GB/s (%.1f %% of 8 TB/s)  %s" % (
        F, L + ex, 8 * (L + ex), m.last_kernel(), np.median(t), t.min(), 8.0 * F * L / np.median(t) / 1e6,
        8.0 * F * L / np.median(t) / 1e6 / 80.0, "same bits" if same else "RESULT DIFFERS"), flush=True)
