#!/usr/bin/env python3
"""Measurement helper (GPU box): does the receive kernel run slower right behind the VALU-saturated timing_scan_kernel
(as in QPSK_TIMING_HIST batches, where rocprofv3 reads ~190 us for it against ~162 us in fixed-index batches)?
Single launches between HIP events, with and without a scan kernel launched just before."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, qpsk_amd, numpy as np
dev = torch.device("cuda", 0)
mf = qpsk_amd.Modem(fs=bench.FS, rs=bench.RS, frame_size=bench.L, timing_mode=qpsk_amd.TIMING_FIXED, fixed_index=6)
F = 4096
x = bench.synth_frames_gpu(torch, dev, F, mf.taps, seed=1)
sym = torch.empty((F, mf.nsym), dtype=torch.uint8, device=dev)
fr = torch.empty((F,), dtype=torch.float32, device=dev); ph = torch.empty_like(fr)
idx = torch.empty((F,), dtype=torch.int32, device=dev); hist = torch.empty((F, 8), dtype=torch.int32, device=dev)
def timed(pre_scan, n=30):
    ts = []
    for _ in range(n):
        if pre_scan:
            mf.timing_scan(x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); mf.rx_batch_raw(x, F, sym, fr, ph); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return np.median(ts)
for _ in range(200): mf.rx_batch_raw(x, F, sym, fr, ph)
torch.cuda.synchronize()
print("fixed-index receive kernel alone:            %.4f ms" % timed(False))
print("the same right behind a timing_scan_kernel:  %.4f ms" % timed(True))
print("alone again:                                 %.4f ms" % timed(False))
