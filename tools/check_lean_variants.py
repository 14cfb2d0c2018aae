#!/usr/bin/env python3
"""Measurement build (GPU box; QPSK_HIP_LIB=qpsk_amd/libqpsk_hip_prof.so): the variants of rx_lean_kernel's stream that claim the RIGHT result --
LDS-DMA window staging (QPSK_PIPE_DBG 131072) and the frame-alternating load order (1048576) -- against the oracle, bit for bit, before
anything is read from their timings.  Even decimation offsets (the DMA's 16-byte granules)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import qpsk_amd  # noqa: E402
from oracle.pyoracle import Oracle, TIMING_FIXED  # noqa: E402
from sigutil import make_frames  # noqa: E402

fs, rs = 19200.0, 2400.0
orc = Oracle()
bad = 0
for L, F, ix in ((1024, 8192, 6), (2048, 4608, 0), (1536, 8192, 2), (1024, 5120, 4)):
    m = qpsk_amd.Modem(fs=fs, rs=rs, frame_size=L, timing_mode=qpsk_amd.TIMING_FIXED, fixed_index=ix)
    x, _ = make_frames(F, L, 8, m.taps, fs, offset_hz=35.0, base_seed=L + ix, noise=0.04)
    want = orc.rx_batch(x, fs, rs, timing_mode=TIMING_FIXED, fixed_index=ix, threads=16)
    xd = torch.from_numpy(x).cuda()
    # (round 5's first build selected LDS-DMA staging with QPSK_PIPE_DBG bit 131072; it is the product default behind QPSK_LEAN_DMA since
    # -- 0: through registers, 1: DMA with a window per unit where the LDS allows, 2: DMA with one window per FIR wave -- and the bit is gone:
    # ADVICE r5.  Round 6 adds the serial wave's lane pairing, QPSK_LEAN_PAIR 0 / 2.  dbg 1048576 = frame-alternating loads, measurement build only.)
    for dbg, dma, pair in ((0, 1, 0), (0, 0, 0), (0, 2, 0), (0, 1, 2), (0, 0, 2), (1048576, 1, 0)):
        m.tune(pipe_dbg=dbg if dbg else None, lean_dma=dma, lean_pair=pair)
        got = m.rx_batch(xd)
        m.sync()
        ok = all(np.array_equal(got[k].cpu().numpy().view(np.uint8), want[k].view(np.uint8)) for k in ("sym", "freq", "phase"))
        print("L %5d F %5d index %d dbg %8d lean_dma %d lean_pair %d %-16s %s" % (L, F, ix, dbg, dma, pair, m.last_kernel(), "bit-exact" if ok else "DIFFERS"), flush=True)
        bad += not ok
        if not ok:
            sys.exit(1)         # one wrong variant: stop, nothing further touches the GPU
    m.close()
sys.exit(1 if bad else 0)
