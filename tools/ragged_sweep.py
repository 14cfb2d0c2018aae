#!/usr/bin/env python3
"""Measurement helper (GPU box): qpsk_rx_batch on batches that are NOT whole workgroups against the next whole-workgroup size, one
process, launches interleaved (round 5 sent a ragged batch's remainder to a second, serialized launch).

    python tools/ragged_sweep.py [frames ...]        default: 4097 5001 8191 8193 8200
For each size F: G = the library's frames per workgroup (ceil(F / CUs), even, at most 32), W = ceil(F / G) * G the next whole-workgroup
size; median ms per launch of F and of W frames, their ratio, the kernel that served F."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch
    import qpsk_amd
    sizes = [int(a) for a in sys.argv[1:]] or [4097, 5001, 8191, 8193, 8200]
    dev = torch.device("cuda", 0)
    m = qpsk_amd.Modem(fs=bench.FS, rs=bench.RS, frame_size=bench.L, timing_mode=qpsk_amd.TIMING_FIXED, fixed_index=bench.FIXED_INDEX)
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    fmax = 0
    pairs = []
    for F in sizes:
        G = min(32, -(-F // ncu))
        G += G & 1
        W = -(-F // G) * G
        pairs.append((F, W, G))
        fmax = max(fmax, W)
    x = bench.synth_frames_gpu(torch, dev, fmax, m.taps, seed=1000)
    sym = torch.empty((fmax, m.nsym), dtype=torch.uint8, device=dev)
    freq = torch.empty((fmax,), dtype=torch.float32, device=dev)
    phase = torch.empty((fmax,), dtype=torch.float32, device=dev)

    def t(F, n=6):
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for a, b in evs:
            a.record()
            m.rx_batch_raw(x, F, sym, freq, phase)
            b.record()
        torch.cuda.synchronize()
        return [a.elapsed_time(b) for a, b in evs]

    print("%8s %8s %4s %12s %12s %8s  %s" % ("frames", "whole", "G", "ms", "ms (whole)", "ratio", "kernel"))
    for F, W, G in pairs:
        tf, tw = [], []
        for r in range(9):
            a, b = t(F), t(W)
            if r:
                tf += a
                tw += b
        m.rx_batch_raw(x, F, sym, freq, phase)
        torch.cuda.synchronize()
        k = m.last_kernel()
        print("%8d %8d %4d %12.4f %12.4f %8.3f  %s" % (F, W, G, np.median(tf), np.median(tw), np.median(tf) / np.median(tw), k))


if __name__ == "__main__":
    main()
