import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, qpsk_amd
from sigutil import make_frames, random_frames
from oracle.pyoracle import Oracle, TIMING_HIST
fs, rs, L, F = 19200.0, 2400.0, 2048, 64
m = qpsk_amd.Modem(fs=fs, rs=rs, frame_size=L, timing_mode=TIMING_HIST)
orc = Oracle()
x, _ = make_frames(F, L, 8, m.taps, fs, offset_hz=40.0, base_seed=10, noise=0.03)
want = orc.rx_batch(x, fs, rs, timing_mode=TIMING_HIST)
xd = torch.from_numpy(x).cuda()
m.tune(hist_onepass=1)
for call, tune in enumerate([{}, {}]):
    m.tune(**tune)
    got = m.rx_batch(xd); m.sync()
    print("tune", tune)
    import ctypes as C
    st = (C.c_int32 * 5)(); m.L.qpsk_test_hist_state(m.h, st)
    print("call", call, m.last_kernel(), "hist state", list(st))
    s = got["sym"].cpu().numpy()
    bad = (s != want["sym"])
    print(" index want", want["index"][:16], "got", got["index"].cpu().numpy()[:16])
    print(" mismatches per frame", bad.sum(axis=1)[:32])
    if bad.any():
        f = int(np.argmax(bad.sum(axis=1)))
        print(" frame", f, "bad positions", np.nonzero(bad[f])[0][:64])
        print(" per chunk", bad[f].reshape(-1, 64).sum(axis=1), "frame 1:", bad[1].reshape(-1, 64).sum(axis=1))
        print(" got ", s[f][:24]); print(" want", want["sym"][f][:24])
        print(" got ", s[1][:24]); print(" want", want["sym"][1][:24])
        print(" freq eq", np.array_equal(got["freq"].cpu().numpy().view(np.uint32), want["freq"].view(np.uint32)))
