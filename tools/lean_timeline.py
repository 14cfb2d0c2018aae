#!/usr/bin/env python3
"""Measurement helper (GPU box): the wall-clock timeline of ONE rx_lean_kernel launch across the chip -- every workgroup's entry, its
serial wave's first and last step and the end of each of its waves on the 100 MHz constant clock (measurement build, dbg bit 23) --
against the launch's duration between two events: what of a launch is not the 2048 steps.

    make -C qpsk_amd/csrc profile
    python tools/lean_timeline.py [frames [KEY=VAL ...]]
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

sys.path.insert(0, ROOT)
PROF = os.path.join(ROOT, "qpsk_amd", "libqpsk_hip_prof.so")
if not os.path.exists(PROF):
    raise SystemExit("build the measurement library first: make -C qpsk_amd/csrc profile")
os.environ["QPSK_HIP_LIB"] = PROF

import torch  # noqa: E402
import bench  # noqa: E402
import qpsk_amd  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
st = dict(kv.split("=") for kv in sys.argv[2:])
dev = torch.device("cuda", 0)
m = qpsk_amd.Modem(fs=bench.FS, rs=bench.RS, frame_size=bench.L, timing_mode=qpsk_amd.TIMING_FIXED, fixed_index=6)
x = bench.synth_frames_gpu(torch, dev, frames, m.taps, seed=1)
sym = torch.empty((frames, m.nsym), dtype=torch.uint8, device=dev)
fr = torch.empty((frames,), dtype=torch.float32, device=dev)
ph = torch.empty_like(fr)
m.tune(**{k: int(str(v), 0) for k, v in st.items()})
for _ in range(200):
    m.rx_batch_raw(x, frames, sym, fr, ph)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    m.rx_batch_raw(x, frames, sym, fr, ph)
e1.record()
torch.cuda.synchronize()
base = e0.elapsed_time(e1) / 20
m.tune(pipe_dbg=4096 | 8388608)
for _ in range(20):
    m.rx_batch_raw(x, frames, sym, fr, ph)
e0.record()
m.rx_batch_raw(x, frames, sym, fr, ph)
e1.record()
torch.cuda.synchronize()
print("==== %d frames, %s, %s: %.4f ms per launch without stamps (20 back to back); the stamped launch %.4f ms between its events"
      % (frames, st, m.last_kernel(), base, e0.elapsed_time(e1)), flush=True)
G = frames // 256 if frames % 256 == 0 else 16
nwg = (frames + G - 1) // G
buf = (ctypes.c_ulonglong * (48 * 1024))()
lib = ctypes.CDLL(PROF)
lib.qpsk_prof_timeline.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = lib.qpsk_prof_timeline(buf, 48 * 1024)
if rc:
    raise SystemExit("qpsk_prof_timeline: %d" % rc)
wgs = []
for b in range(1024):
    r = buf[48 * b:48 * b + 48]
    if r[0] and r[3]:
        wgs.append({"entry": r[0], "start": r[1], "first": r[2], "last": r[3], "end": r[4], "fir": [v for v in r[5:16] if v],
                    "fir_barrier": [v for v in r[16:32] if v], "fir_stream": [v for v in r[32:48] if v]})
t0 = min(w["entry"] for w in wgs)
t1 = max(max([w["end"]] + w["fir"]) for w in wgs)
us = lambda v: 0.01 * v


def stats(name, vals):
    vals = sorted(vals)
    print("%-62s min %7.2f  median %7.2f  max %7.2f us" % (name, us(vals[0]), us(vals[len(vals) // 2]), us(vals[-1])))


print("%d workgroups stamped; first entry -> last wave's end: %.2f us" % (len(wgs), us(t1 - t0)))
stats("entry after the first workgroup's", [w["entry"] - t0 for w in wgs])
stats("entry -> the serial wave is in costas_wave", [w["start"] - w["entry"] for w in wgs])
stats("entry -> the last FIR wave is behind the barrier", [max(w["fir_barrier"]) - w["entry"] for w in wgs if w["fir_barrier"]])
stats("entry -> the last FIR wave enters its stream", [max(w["fir_stream"]) - w["entry"] for w in wgs if w["fir_stream"]])
stats("... -> its first step (first chunk of every unit is there)", [w["first"] - w["start"] for w in wgs])
stats("first -> last step", [w["last"] - w["first"] for w in wgs])
stats("last step -> the serial wave's end", [w["end"] - w["last"] for w in wgs])
stats("last step -> the workgroup's last FIR wave has flushed", [max(w["fir"]) - w["last"] for w in wgs if w["fir"]])
stats("the workgroup's end before the launch's last", [t1 - max([w["end"]] + w["fir"]) for w in wgs])
