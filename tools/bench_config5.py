#!/usr/bin/env python3
"""Measurement helper (GPU box): BASELINE config 5 -- 1200 baud, 8x oversample, 1,048,576 samples per frame,
loop bandwidths TAU/100 .. TAU/200 (11 values) as independent Costas chains sharing one FIR pass.
Frame count: 384 by default (384 x 11 = 4224 chains >= 4096, 3 GiB of input; SURVEY 8(d) leaves the count open)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=384)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import torch
    import qpsk_amd
    dev = torch.device("cuda", 0)
    fs, rs, L, F = 9600.0, 1200.0, 1 << 20, args.frames
    bws = [np.float32(bench.TAU / d) if hasattr(bench, "TAU") else np.float32(2 * np.pi / d) for d in range(100, 201, 10)]
    m = qpsk_amd.Modem(fs=fs, rs=rs, frame_size=L, timing_mode=qpsk_amd.TIMING_FIXED, fixed_index=6)
    # the generator works on config-2 sized rows: build the long frames from 64 rows each (a test signal, not a modem
    # frame boundary: rows are concatenated in time)
    rows = bench.synth_frames_gpu(torch, dev, F * (L // bench.L), m.taps, seed=55)
    x = rows.reshape(F, L, 2).contiguous()
    del rows
    ts = []
    for r in range(args.reps + 1):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = m.rx_batch_bw(x, bws)
        e.record()
        torch.cuda.synchronize()
        if r:
            ts.append(a.elapsed_time(e))
    t = float(np.median(ts))
    hz = out["freq"].float().cpu().numpy().astype(np.float64) * rs / (2 * np.pi)
    print("config 5: %d frames x %d samples x %d loop bandwidths: %.2f ms -> %.0f Msamples/s of input, %.0f M loop steps/s, "
          "%.1f GB/s (%.2f%% of 8 TB/s); loop offsets %.2f..%.2f Hz" % (
              F, L, len(bws), t, F * L / t / 1e3, F * (L // 8) * len(bws) / t / 1e3, 8.0 * F * L / t / 1e6,
              8.0 * F * L / t / 1e6 / 80.0, hz.min(), hz.max()))


if __name__ == "__main__":
    main()
