#!/usr/bin/env python3
"""Measurement helper (GPU box): consecutive batches on ONE stream against the same batches dealt to TWO contexts on two streams
(each with its own result buffers, as a host that double-buffers its results would).  A launch's workgroups hold a CU each for the whole
launch (LDS), so two launches never share a CU -- but with two streams the next launch's workgroups take over CU by CU as the previous
launch's finish, instead of behind its slowest workgroup and the dispatch gap.

    python tools/two_streams.py [frames [steps [rounds]]]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import bench  # noqa: E402
import qpsk_amd  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda", 0)
streams = [torch.cuda.Stream(dev) for _ in range(3)]
ms = []
outs = []
for s in streams:
    with torch.cuda.stream(s):
        m = qpsk_amd.Modem(fs=bench.FS, rs=bench.RS, frame_size=bench.L, timing_mode=qpsk_amd.TIMING_FIXED, fixed_index=6)
        ms.append(m)
        outs.append((torch.empty((frames, m.nsym), dtype=torch.uint8, device=dev), torch.empty((frames,), dtype=torch.float32, device=dev),
                     torch.empty((frames,), dtype=torch.float32, device=dev)))
x = bench.synth_frames_gpu(torch, dev, frames, ms[0].taps, seed=1)
torch.cuda.synchronize()


def run(nstreams, k):
    for i in range(k):
        j = i % nstreams
        ms[j].rx_batch_raw(x, frames, *outs[j])


def timed(nstreams, k):
    run(nstreams, 10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(nstreams, k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / k


t_end = time.perf_counter() + 0.5
while time.perf_counter() < t_end:      # clocks
    run(1, 20)
    torch.cuda.synchronize()
for r in range(rounds):
    print("%d frames, %d steps: one stream %.4f ms per step, two streams %.4f, three %.4f" %
          (frames, steps, timed(1, steps), timed(2, steps), timed(3, steps)), flush=True)
ref = [o.clone() for o in outs[0]]
for j in (1, 2):
    for a_, b_ in zip(ref, outs[j]):
        assert torch.equal(a_.view(torch.uint8), b_.view(torch.uint8)), "the streams' results differ"
print("results of the three contexts equal bit for bit; kernel:", ms[0].last_kernel())
