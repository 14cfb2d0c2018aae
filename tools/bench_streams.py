#!/usr/bin/env python3
"""Measurement helper (GPU box): throughput of the streaming mode (consecutive rx_frame() calls with carried
state, qpsk.c:344-354) for N streams, complex and PCM input.  Reports ms per block of all streams."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=4096)
    ap.add_argument("--blocks", type=int, default=6)
    ap.add_argument("--timing", choices=["fixed", "hist"], default="hist")
    args = ap.parse_args()
    import torch
    import qpsk_amd
    dev = torch.device("cuda", 0)
    n = args.streams
    mode = qpsk_amd.TIMING_FIXED if args.timing == "fixed" else qpsk_amd.TIMING_HIST
    m = qpsk_amd.Modem(fs=bench.FS, rs=bench.RS, frame_size=bench.L, timing_mode=mode, fixed_index=bench.FIXED_INDEX)
    x = bench.synth_frames_gpu(torch, dev, n, m.taps, seed=5)
    pcm = (x[:, :, 0] * 8000.0).clamp(-32767, 32767).to(torch.int16).contiguous()
    for name, fn, data in (("cplx", m.streams_rx_cplx, x), ("pcm", m.streams_rx_pcm, pcm)):
        m.streams_reset(n, 1500.0)
        ts = []
        for b in range(args.blocks):
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn(data, want_costas=False)
            e.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(e))
        t = float(np.median(ts[1:]))
        print("streams %s (%s timing): %d x %d samples per block: %.3f ms -> %.0f Msamples/s" % (
            name, args.timing, n, bench.L, t, n * bench.L / t / 1e3))
    # transmit side: the same number of transmitters, one block of nsym symbols each (qpsk.c:273-285)
    m.tx_reset(n, 1550.0)
    sym = torch.randint(0, 4, (n, m.nsym), dtype=torch.uint8, device=dev)
    ts = []
    for b in range(args.blocks):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        m.tx_symbols(sym)
        e.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(e))
    t = float(np.median(ts[1:]))
    print("transmitters: %d x %d samples per block: %.3f ms -> %.0f Msamples/s" % (n, bench.L, t, n * bench.L / t / 1e3))


if __name__ == "__main__":
    main()
