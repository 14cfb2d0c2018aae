#!/usr/bin/env python3
"""Measurement helper (GPU box): per-launch times of the first launches after a synchronisation, as bench.py's timed region sees them (a settle
loop, W warmup launches, synchronize, then K launches): is the 20-step region slower than steady state because its first launches are?
    python tools/launch_ramp.py [frames] [K]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import qpsk_amd  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 25
dev = torch.device("cuda", 0)
m = qpsk_amd.Modem(fs=bench.FS, rs=bench.RS, frame_size=bench.L, timing_mode=qpsk_amd.TIMING_FIXED, fixed_index=bench.FIXED_INDEX)
x = bench.tx_frames_gpu(torch, dev, qpsk_amd, F, seed=1000)
sym = torch.empty((F, m.nsym), dtype=torch.uint8, device=dev)
fr = torch.empty((F,), dtype=torch.float32, device=dev)
ph = torch.empty_like(fr)
rows = []
for rep in range(6):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.25:
        for _ in range(10):
            m.rx_batch_raw(x, F, sym, fr, ph)
        torch.cuda.synchronize()
    for _ in range(5):
        m.rx_batch_raw(x, F, sym, fr, ph)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    ev[0].record()
    for k in range(K):
        m.rx_batch_raw(x, F, sym, fr, ph)
        ev[k + 1].record()
    torch.cuda.synchronize()
    rows.append([ev[k].elapsed_time(ev[k + 1]) for k in range(K)])
r = np.array(rows)
print("%d frames: per-launch ms behind a synchronisation, median of %d repetitions (an event between every two launches)" % (F, len(rows)))
print(" ".join("%.4f" % v for v in np.median(r, axis=0)))
print("mean of the first 20: %.4f   of launches 10..%d: %.4f" % (np.median(r, axis=0)[:20].mean(), K - 1, np.median(r, axis=0)[10:].mean()))
