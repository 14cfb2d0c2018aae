#!/usr/bin/env python3
"""Measurement helper (GPU box): BASELINE config 3 (config 2's batch with the FFT timing estimate in front) against config 2
in ONE process, the way bench.py times a step: K back-to-back qpsk_rx_batch calls between two events, after a clock settle.

    python tools/bench_config3.py [--frames 4096] [--steps 200] [--rounds 3]
Prints per round: fixed-index step, FFT-timing step, the estimator alone (qpsk_timing_fft_bin_batch), the histogram-mode step.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--hist", action="store_true", help="time the histogram mode too")
    args = ap.parse_args()
    import torch
    import qpsk_amd
    dev = torch.device("cuda", 0)
    F = args.frames
    mk = lambda mode: qpsk_amd.Modem(fs=bench.FS, rs=bench.RS, frame_size=bench.L, timing_mode=mode, fixed_index=bench.FIXED_INDEX)
    mfix, mfft = mk(qpsk_amd.TIMING_FIXED), mk(qpsk_amd.TIMING_FFT)
    mh = mk(qpsk_amd.TIMING_HIST) if args.hist else None
    x = bench.synth_frames_gpu(torch, dev, F, mfix.taps, seed=1000)
    sym = torch.empty((F, mfix.nsym), dtype=torch.uint8, device=dev)
    freq = torch.empty((F,), dtype=torch.float32, device=dev)
    phase = torch.empty((F,), dtype=torch.float32, device=dev)
    idx = torch.empty((F,), dtype=torch.int32, device=dev)

    def region(fn, steps):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.25:
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(steps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / steps

    runs = [("config 2 (fixed index)", lambda: mfix.rx_batch_raw(x, F, sym, freq, phase)),
            ("config 3 (FFT timing)", lambda: mfft.rx_batch_raw(x, F, sym, freq, phase)),
            ("FFT estimator alone", lambda: mfft._check(mfft.L.qpsk_timing_fft_bin_batch(mfft.h, x.data_ptr(), F, idx.data_ptr(), None, None)))]
    if mh:
        runs.append(("histogram mode", lambda: mh.rx_batch_raw(x, F, sym, freq, phase)))
    nbytes = 8.0 * F * bench.L
    for r in range(args.rounds):
        for name, fn in runs:
            ms = region(fn, args.steps)
            print("round %d  %-26s %.4f ms per step  (%.1f %% of 8 TB/s on the batch's bytes)" % (r, name, ms, nbytes / ms / 1e6 / 80.0), flush=True)
    mfft.rx_batch_raw(x, F, sym, freq, phase)
    s3 = sym.clone()
    mfix.rx_batch_raw(x, F, sym, freq, phase)
    torch.cuda.synchronize()
    print("config 3 symbols == config 2 symbols:", bool(torch.equal(s3, sym)), " last kernels:", mfft.last_kernel(), mfix.last_kernel())


if __name__ == "__main__":
    main()
