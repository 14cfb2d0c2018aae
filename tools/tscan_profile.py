#!/usr/bin/env python3
"""Measurement helper (GPU box): shader cycles per 256-sample tile that the FIR waves of timing_scan_kernel's
workgroup 0 spend staging, filtering, waiting for the scan waves and handing over.  Needs the measurement build
(make -C qpsk_amd/csrc profile); the kernel prints when the batch has an odd number of frames."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["QPSK_HIP_LIB"] = os.path.join(ROOT, "qpsk_amd", "libqpsk_hip_prof.so")
import torch  # noqa: E402
import bench  # noqa: E402
import qpsk_amd  # noqa: E402

dev = torch.device("cuda", 0)
m = qpsk_amd.Modem(fs=bench.FS, rs=bench.RS, frame_size=bench.L, timing_mode=qpsk_amd.TIMING_HIST)
F = 4095
x = bench.synth_frames_gpu(torch, dev, F, m.taps, seed=1)
sym = torch.empty((F, m.nsym), dtype=torch.uint8, device=dev)
fr = torch.empty((F,), dtype=torch.float32, device=dev)
ph = torch.empty_like(fr)
for _ in range(2):
    m.rx_batch_raw(x, F, sym, fr, ph)
    torch.cuda.synchronize()
    print("----", flush=True)
