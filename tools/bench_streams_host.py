#!/usr/bin/env python3
"""Measurement helper (GPU box): qpsk_streams_rx_pcm_host -- one rx_frame() block per call, host buffers, the reference's call
pattern (qpsk.c:344-354) -- for S streams of the shipped configuration (FS 9600, RS 2400, 512-sample blocks), with the
one-launch-per-block kernel (streamblock.hip) and with the five-kernel composition.

    python tools/bench_streams_host.py [streams] [blocks]
With QPSK_HIP_LIB=qpsk_amd/libqpsk_hip_sbprof.so (make VARIANT=sbprof EXTRA=-DQPSK_SBLK_PROF) stream 0's two waves print their phases.
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import qpsk_amd  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
B = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
fs, rs, L = 9600.0, 2400.0, 512
rng = np.random.default_rng(1)
pcm = (3000 * rng.standard_normal((B, S, L))).astype(np.int16)
for block, poll in ((1, 1), (1, 0), (0, 0)):
    m = qpsk_amd.Modem(fs=fs, rs=rs, frame_size=L)
    m.tune(stream_block=block)
    m.tune(stream_poll=poll)
    m.streams_reset(S, 1500.0)
    N = m.nsym
    st = np.zeros((S, 2), np.float32)
    sym = np.zeros((S, N), np.uint8)
    cos = np.zeros((S, N, 2), np.float32)
    idx = np.zeros(S, np.int32)
    fn = m.L.qpsk_streams_rx_pcm_host
    args = (m.h, None, C.c_void_p(st.ctypes.data), C.c_void_p(sym.ctypes.data), C.c_void_p(cos.ctypes.data), C.c_void_p(idx.ctypes.data))
    ptrs = [C.c_void_p(pcm[k].ctypes.data) for k in range(B)]
    for k in range(min(B, 50)):
        assert fn(args[0], ptrs[k], *args[2:]) == 0, m.L.qpsk_last_error()
    t0 = time.perf_counter()
    for k in range(B):
        fn(args[0], ptrs[k], *args[2:])
    dt = time.perf_counter() - t0
    print("%-60s %d stream(s) x %d samples: %.1f us per block = %.2f Msamples/s (ctypes call included)" % (
        ("one launch per block, completion watched in pinned memory" if poll else "one launch per block, hipStreamSynchronize") if block else "five-kernel composition + copies", S, L, dt / B * 1e6, S * L * B / dt / 1e6), flush=True)
    m.close()
