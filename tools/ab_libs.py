#!/usr/bin/env python3
"""Measurement helper (GPU box): interleaved A/B timing of library builds in ONE process on the bench workload.

    python tools/ab_libs.py [--frames 4096] [--rounds 40] [--dbg N] qpsk_amd/libqpsk_hip.so qpsk_amd/libqpsk_hip_old.so

Run-to-run noise of a single bench.py line is +-3 %; here the builds alternate call by call on the same input and the
median of each is reported, which resolves differences of ~0.5 %."""
import argparse
import importlib.util
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402


def load_build(path, tag):
    """qpsk_amd.lib bound to one shared object (the module keeps its handle in a global)"""
    os.environ["QPSK_HIP_LIB"] = os.path.abspath(path)
    import qpsk_amd  # noqa: F401  (package import, once)
    spec = importlib.util.spec_from_file_location("qpsk_amd.lib_" + tag, os.path.join(ROOT, "qpsk_amd", "lib.py"))
    mod = importlib.util.module_from_spec(spec)
    mod.__package__ = "qpsk_amd"
    spec.loader.exec_module(mod)
    mod.load()
    return mod


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--frames", type=int, default=4096)
    ap.add_argument("--rounds", type=int, default=40)
    ap.add_argument("--dbg", default=None)
    ap.add_argument("--timing", choices=["fixed", "hist", "fft"], default="fixed")
    args = ap.parse_args()
    if args.dbg is not None:
        os.environ["QPSK_PIPE_DBG"] = args.dbg
    dev = torch.device("cuda", 0)
    mods = [load_build(p, str(i)) for i, p in enumerate(args.libs)]
    modems = [m.Modem(fs=bench.FS, rs=bench.RS, frame_size=bench.L, timing_mode={"fixed": m.TIMING_FIXED, "hist": m.TIMING_HIST, "fft": m.TIMING_FFT}[args.timing], fixed_index=6) for m in mods]
    x = bench.synth_frames_gpu(torch, dev, args.frames, modems[0].taps, seed=1)
    sym = torch.empty((args.frames, modems[0].nsym), dtype=torch.uint8, device=dev)
    fr = torch.empty((args.frames,), dtype=torch.float32, device=dev)
    ph = torch.empty_like(fr)
    times = [[] for _ in modems]
    for r in range(args.rounds + 3):
        for i, m in enumerate(modems):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            m.rx_batch_raw(x, args.frames, sym, fr, ph)
            e1.record()
            torch.cuda.synchronize()
            if r >= 3:
                times[i].append(e0.elapsed_time(e1))
    ref = None
    for p, t, m in zip(args.libs, times, modems):
        t.sort()
        m.rx_batch_raw(x, args.frames, sym, fr, ph)       # builds differ in layout and schedule, never in results
        torch.cuda.synchronize()
        got = (sym.clone(), fr.clone(), ph.clone())
        ref = ref or got
        same = all(torch.equal(a.view(torch.uint8), b.view(torch.uint8)) for a, b in zip(got, ref))
        print("%-40s median %.4f ms   min %.4f   p90 %.4f   %s" % (os.path.basename(p), statistics.median(t), t[0], t[int(len(t) * .9)],
                                                                  "same bits as the first" if same else "RESULT DIFFERS FROM THE FIRST"))


if __name__ == "__main__":
    main()
