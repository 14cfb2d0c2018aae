#!/usr/bin/env python3
"""Measurement (GPU box): many streams at the reference's SHIPPED configuration (FS 9600, RS 2400, FRAME_SIZE 512: CYCLES = 4,
qpsk.h:16-23), device buffers, back to back: the one-launch-per-block kernel against the kernels apart, per stream count.
    python tools/bench_streams_shipped.py [streams ...]          (QPSK_BENCH_L=<frame size> for other block lengths; QPSK_BENCH_FS=19200 for CYCLES = 8, where
stream_scan_kernel is the third candidate)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import qpsk_amd  # noqa: E402

dev = torch.device("cuda", 0)
L = int(os.environ.get("QPSK_BENCH_L", "512"))
FS = float(os.environ.get("QPSK_BENCH_FS", "9600"))
CPLX = os.environ.get("QPSK_BENCH_CPLX", "0") == "1"      # complex blocks at the rrc_fir() call (qpsk_streams_rx_cplx) instead of PCM
for n in [int(a) for a in sys.argv[1:]] or [256, 1024, 2048, 4096, 16384]:
    g = torch.Generator(device=dev)
    g.manual_seed(n)
    pcm = (6000 * torch.randn((n, L), generator=g, device=dev)).to(torch.int16)
    xc = torch.randn((n, L, 2), generator=g, device=dev, dtype=torch.float32) if CPLX else None
    res = []
    for block, scan in ((1, 0), (0, 0)) + (((0, 1),) if L % 256 == 0 else ()) + ((None, None),):      # last: the library's own choice
        m = qpsk_amd.Modem(fs=FS, rs=2400.0, frame_size=L, timing_mode=qpsk_amd.TIMING_HIST)
        m.tune(stream_block=block)
        m.tune(stream_scan=scan)
        m.streams_reset(n, 1500.0)
        sym = torch.empty((n, m.nsym), dtype=torch.uint8, device=dev)
        fr = torch.empty((n,), dtype=torch.float32, device=dev)
        ph = torch.empty_like(fr)
        idx = torch.empty((n,), dtype=torch.int32, device=dev)
        if CPLX:
            fn = lambda: m._check(m.L.qpsk_streams_rx_cplx(m.h, xc.data_ptr(), sym.data_ptr(), fr.data_ptr(), ph.data_ptr(), None, idx.data_ptr()))
        else:
            fn = lambda: m._check(m.L.qpsk_streams_rx_pcm(m.h, pcm.data_ptr(), sym.data_ptr(), fr.data_ptr(), ph.data_ptr(), None, idx.data_ptr()))
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t0 = time.time()
        last = 0.0
        while time.time() - t0 < 1.5:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                fn()
            e1.record()
            torch.cuda.synchronize()
            last = e0.elapsed_time(e1) / 200
        res.append((("the library's choice: " if block is None else "") + m.last_kernel(), last))
        m.close()
    print("%6d streams x %d samples at FS %g%s: " % (n, L, FS, ", complex input" if CPLX else "") + "; ".join("%s %.1f us per block (%.0f Msamples/s)" % (k, t * 1e3, n * L / t / 1e3) for k, t in res), flush=True)
