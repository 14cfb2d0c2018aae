#!/usr/bin/env python3
"""Measurement helper (GPU box): the FIR waves' cycle accounting of workgroup 0 (tools/fir_wave_profile.py) for any
QPSK_PIPE_DBG combination on config 2, e.g. the two-lane-mapping workgroup:

    make -C qpsk_amd/csrc profile            # build container: libqpsk_hip_prof.so
    python tools/prof_dbg.py 160             # 32 (print the accounting) + 128 (mixed lane mappings)
"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROF = os.path.join(ROOT, "qpsk_amd", "libqpsk_hip_prof.so")
if not os.path.exists(PROF):
    raise SystemExit("build the measurement library first: make -C qpsk_amd/csrc profile")
os.environ["QPSK_HIP_LIB"] = PROF

import torch  # noqa: E402
import bench  # noqa: E402
import qpsk_amd  # noqa: E402

dev = torch.device("cuda", 0)
frames = 4096
m = qpsk_amd.Modem(fs=bench.FS, rs=bench.RS, frame_size=bench.L, timing_mode=qpsk_amd.TIMING_FIXED, fixed_index=6)
x = bench.synth_frames_gpu(torch, dev, frames, m.taps, seed=1)
sym = torch.empty((frames, m.nsym), dtype=torch.uint8, device=dev)
fr = torch.empty((frames,), dtype=torch.float32, device=dev)
ph = torch.empty_like(fr)
for dbg in sys.argv[1:] or ["32"]:
    os.environ["QPSK_PIPE_DBG"] = dbg
    print("==== QPSK_PIPE_DBG=%s" % dbg, flush=True)
    m.rx_batch_raw(x, frames, sym, fr, ph)
    torch.cuda.synchronize()
