"""Decimated symbols (what the Costas loop consumes) of synthetic frames through the oracle, for tools/spec_sim.c:
    python tools/spec_sim_gen.py <frames> <noise> <offset_hz> <out.bin>"""
import sys, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle.pyoracle import Oracle, TAU
from sigutil import make_frames
orc = Oracle()
fs, rs, L = 19200.0, 2400.0, 16384
taps = orc.rrc_make(fs, rs, np.float32(.35))
F = int(sys.argv[1]); noise=float(sys.argv[2]); off=float(sys.argv[3])
x,_ = make_frames(F, L, 8, taps, fs, offset_hz=off, base_seed=4242, noise=noise)
out = np.zeros((F, L//8, 2), np.float32)
from oracle.pyoracle import TIMING_FIXED
for f in range(F):
    m = orc.modem(fs, rs, L, timing_mode=TIMING_FIXED, fixed_index=6)
    m.rx_cplx(x[f])
    d = m.decimated
    out[f] = d[L//8:]
out.tofile(sys.argv[4])
print(out.shape, out[0,:3], np.abs(out[...,0]+1j*out[...,1]).mean())
