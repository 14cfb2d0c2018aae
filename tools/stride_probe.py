#!/usr/bin/env python3
"""Measurement helper (GPU box): does the 128 KB frame stride (16384 complex samples) cost HBM bandwidth?  Times rx_lean_kernel
on 8192 frames of several frame sizes -- the product stream and, with the measurement build, the stream without filter
arithmetic, window reads and flush (QPSK_PIPE_DBG 49153: the kernel's memory-side floor) -- and prints GB/s.

    QPSK_HIP_LIB=qpsk_amd/libqpsk_hip_prof.so python tools/stride_probe.py [frame_size ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
import qpsk_amd  # noqa: E402

dev = torch.device("cuda", 0)
frames = 8192
sizes = [int(a) for a in sys.argv[1:]] or [16384, 16896, 15872, 16384]
prof = "prof" in os.environ.get("QPSK_HIP_LIB", "")
for L in sizes:
    m = qpsk_amd.Modem(fs=bench.FS, rs=bench.RS, frame_size=L, timing_mode=qpsk_amd.TIMING_FIXED, fixed_index=6)
    g = torch.Generator(device=dev)
    g.manual_seed(L)
    x = (torch.rand((frames, L, 2), generator=g, device=dev, dtype=torch.float32) - 0.5)
    sym = torch.empty((frames, m.nsym), dtype=torch.uint8, device=dev)
    fr = torch.empty((frames,), dtype=torch.float32, device=dev)
    ph = torch.empty_like(fr)
    for dbg in ([0, 49153] if prof else [0]):
        m.tune(pipe_dbg=dbg)
        for _ in range(300):
            m.rx_batch_raw(x, frames, sym, fr, ph)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100):
                m.rx_batch_raw(x, frames, sym, fr, ph)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 100)
        t = float(np.median(ts))
        print("frame_size %6d (stride %7d B) dbg %5d %s: %.4f ms  %.0f GB/s" % (L, 8 * L, dbg, m.last_kernel(), t, 8.0 * frames * L / t / 1e6), flush=True)
    del x
