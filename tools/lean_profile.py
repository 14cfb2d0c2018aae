#!/usr/bin/env python3
"""Measurement helper (GPU box): where the waves of rx_lean_kernel (and the serial wave of either pipeline kernel) spend their
shader cycles -- the measurement build's stamped streams (fir_lean_prof_asm.h, costas_wave's ring timing).

    make -C qpsk_amd/csrc profile                      # build container: libqpsk_hip_prof.so
    python tools/lean_profile.py [frames [KEY=VAL ...]]   # e.g. 7680 pipe_g=30 pipe_layout_lo=0x20222 pipe_layout_hi=0x111022
Prints one line per wave of workgroups 0 and 77."""
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

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
st = dict(kv.split("=") for kv in sys.argv[2:])
dev = torch.device("cuda", 0)
m = qpsk_amd.Modem(fs=bench.FS, rs=bench.RS, frame_size=bench.L, timing_mode=qpsk_amd.TIMING_FIXED, fixed_index=6)
x = bench.synth_frames_gpu(torch, dev, frames, m.taps, seed=1)
sym = torch.empty((frames, m.nsym), dtype=torch.uint8, device=dev)
fr = torch.empty((frames,), dtype=torch.float32, device=dev)
ph = torch.empty_like(fr)
extra = int(str(st.pop("pipe_dbg", 0)), 0)
m.tune(**{k: int(str(v), 0) for k, v in st.items()})
for _ in range(200):          # clocks settled
    m.rx_batch_raw(x, frames, sym, fr, ph)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    m.rx_batch_raw(x, frames, sym, fr, ph)
e1.record()
torch.cuda.synchronize()
print("==== %d frames, %s, %s: %.4f ms per launch without stamps" % (frames, st, m.last_kernel(), e0.elapsed_time(e1) / 20), flush=True)
m.tune(pipe_dbg=32 | 4096 | extra)
e0.record()
m.rx_batch_raw(x, frames, sym, fr, ph)
e1.record()
torch.cuda.synchronize()
print("     stamped launch: %.4f ms" % e0.elapsed_time(e1), flush=True)
