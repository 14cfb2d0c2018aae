#!/usr/bin/env python3
"""Measurement helper (GPU box): the drop-in rx_frame() (examples/dropin_main.c: one 512-sample int16 block per call,
host buffers, the reference's own call pattern qpsk.c:344-354) next to the reference's rx_frame() compiled here
(oracle/_ref, Makefile flags, one core) on the same box.  Prints both in Msamples/s.

    python tools/bench_dropin.py [blocks]
"""
import os
import re
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
exe = "/tmp/dropin_main"
subprocess.check_call(["gcc", "-std=c11", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "dropin_main.c"),
                       "-L", os.path.join(ROOT, "qpsk_amd"), "-lqpsk_hip", "-Wl,-rpath," + os.path.join(ROOT, "qpsk_amd"), "-lm", "-o", exe])
out = subprocess.run([exe, str(blocks)], capture_output=True, text=True)
print(out.stdout.strip().splitlines()[-2:])
m = re.search(r"= ([0-9.]+) Msamples/s", out.stdout)
gpu = float(m.group(1)) if m else float("nan")
from oracle.pyoracle import Reference, ref_available
if ref_available("shipped"):
    ref = Reference("shipped")
    ref.reset()
    pcm = (3000 * np.random.default_rng(1).standard_normal((blocks, 512))).astype(np.int16)
    t0 = time.perf_counter()
    for k in range(blocks):
        ref.rx_pcm(pcm[k])
    dt = time.perf_counter() - t0
    cpu = 512 * blocks / dt / 1e6
    print("reference rx_frame() on one host core (oracle/_ref, -O0 as its Makefile): %.3f Msamples/s (incl. the ctypes call per block)" % cpu)
else:
    cpu = float("nan")
    print("oracle/_ref did not travel to this box")
print("drop-in rx_frame() through libqpsk_hip: %.3f Msamples/s" % gpu)
