#!/usr/bin/env python3
"""Measurement (GPU box): what stretches of exactly-zero symbols cost (costas_asm.h `ign`; before that every group of such a lane's
workgroup went through the C++ step).  4096 PCM streams x 16384-sample blocks: (a) the first block after qpsk_streams_reset() (every
symbol zero), (b) steady state, (c) steady state with one stream in 16 squelched (all-zero PCM: one such lane in every workgroup of
the loop kernel); and config 2's batch with one all-zero frame in it.  Events around single calls; ms per call."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import qpsk_amd  # noqa: E402

S, L = 4096, 16384
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(5)
pcm = (6000 * torch.randn((S, L), generator=g, device=dev)).to(torch.int16)
sq = pcm.clone()
sq[::16] = 0


def block(m, x, out):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    m._check(m.L.qpsk_streams_rx_pcm(m.h, x.data_ptr(), out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), None, out[3].data_ptr()))
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


for _once in (1,):
    m = qpsk_amd.Modem(fs=19200.0, rs=2400.0, frame_size=L, timing_mode=qpsk_amd.TIMING_HIST)
    out = (torch.empty((S, m.nsym), dtype=torch.uint8, device=dev), torch.empty((S,), dtype=torch.float32, device=dev),
           torch.empty((S,), dtype=torch.float32, device=dev), torch.empty((S,), dtype=torch.int32, device=dev))
    m.streams_reset(S, 1500.0)
    for _ in range(3):          # warm: buffers, code objects
        block(m, pcm, out)
    first = []
    for _ in range(5):
        m.streams_reset(S, 1500.0)
        first.append(block(m, pcm, out))
        block(m, pcm, out)
    steady = sorted(block(m, pcm, out) for _ in range(20))
    for _ in range(4):          # the squelched streams' rows are all zero from their third silent block on
        block(m, sq, out)
    squelched = sorted(block(m, sq, out) for _ in range(20))
    m.sync()
    print("streams: first block after reset %.3f ms (min of 5), steady %.3f ms (median of 20), with 256 of 4096 streams squelched "
          "%.3f ms (median of 20)" % (min(first), steady[10], squelched[10]), flush=True)
    m.close()

import bench  # noqa: E402
m = qpsk_amd.Modem(fs=bench.FS, rs=bench.RS, frame_size=bench.L, timing_mode=qpsk_amd.TIMING_FIXED, fixed_index=bench.FIXED_INDEX)
x = bench.synth_frames_gpu(torch, dev, 4096, m.taps, seed=1)
sym = torch.empty((4096, m.nsym), dtype=torch.uint8, device=dev)
fr = torch.empty((4096,), dtype=torch.float32, device=dev)
ph = torch.empty_like(fr)


def batch(xx):
    ts = []
    for _ in range(30):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        m.rx_batch_raw(xx, 4096, sym, fr, ph)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[15]


t_all = batch(x)
x[777] = 0.0
t_one = batch(x)
x[777, 8000:] = 1.0
x[777, :8000] = 0.0
t_half = batch(x)
m.sync()
print("config 2 (4096 x 16384, %s): %.4f ms per call; with frame 777 all zero %.4f ms; with its first 8000 samples zero %.4f ms" % (
    m.last_kernel(), t_all, t_one, t_half), flush=True)
