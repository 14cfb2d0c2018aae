#!/usr/bin/env python3
"""Measurement helper (GPU box): shader cycles per chunk that the FIR waves of workgroup 0 spend waiting for the
Costas loop, flushing, staging the window, filtering and handing over -- both pipeline geometries.

    make -C qpsk_amd/csrc profile          # in the build container: libqpsk_hip_prof.so (-DQPSK_PIPE_PROFILE)
    python tools/fir_wave_profile.py       # on the GPU box

The product library has no counters and no device printf; this script refuses to run against it."""
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

dev = torch.device("cuda", 0)
settings = [dict(pipe_v=2), dict(pipe_v=1)] if len(sys.argv) < 2 else [dict(kv.split("=") for kv in a.split()) for a in sys.argv[1:]]
for frames in (4096, 8192):
    m = qpsk_amd.Modem(fs=bench.FS, rs=bench.RS, frame_size=bench.L, timing_mode=qpsk_amd.TIMING_FIXED, fixed_index=6)
    x = bench.synth_frames_gpu(torch, dev, frames, m.taps, seed=1)
    sym = torch.empty((frames, m.nsym), dtype=torch.uint8, device=dev)
    fr = torch.empty((frames,), dtype=torch.float32, device=dev)
    ph = torch.empty_like(fr)
    for st in settings:
        m.tune(pipe_layout_lo=None, pipe_layout_hi=None)
        m.tune(**{k: int(str(v), 0) for k, v in st.items()})
        try:
            for _ in range(3):
                m.rx_batch_raw(x, frames, sym, fr, ph)
        except qpsk_amd.QpskError as e:      # a layout made for another batch size
            print("==== %d frames, %s: skipped (%s)" % (frames, st, e), flush=True)
            continue
        torch.cuda.synchronize()
        m.tune(pipe_dbg=32 | int(str(st.get("pipe_dbg", 0)), 0))      # a setting's own bits (e.g. the ablations 1, 2) stay on
        print("==== %d frames, %s" % (frames, st), flush=True)
        m.rx_batch_raw(x, frames, sym, fr, ph)
        torch.cuda.synchronize()
        m.tune(pipe_dbg=0)
        m.tune(pipe_dbg=None)
    del x
