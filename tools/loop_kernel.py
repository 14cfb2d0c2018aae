#!/usr/bin/env python3
"""Measurement helper (GPU box): one library call back to back for some seconds, steady-state ms per call -- the child of
`tools/power_probe.py --cmd "python3 tools/loop_kernel.py <what> [secs] [frames] [launches per loop]"` (board power and shader clock beside it).

    what: fir (qpsk_rrc_fir_batch, the stream kernel)   fir_generic (the compiler-scheduled rrc_fir_kernel)
          fft_est (qpsk_timing_fft_bin_batch)           scan (qpsk_timing_scan_batch)
          config2 / config3 / hist (qpsk_rx_batch in the three timing modes)
          streams (qpsk_streams_rx_pcm: one 16384-sample PCM block of `frames` running streams per call, histogram timing)
          streams_fixed (the same with the fixed timing offset)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import qpsk_amd  # noqa: E402

what = sys.argv[1]
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
F = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
per = int(sys.argv[4]) if len(sys.argv) > 4 else 200      # launches between two synchronisations (counter passes: a handful, with secs = 0)
dev = torch.device("cuda", 0)
mode = {"config3": qpsk_amd.TIMING_FFT, "fft_est": qpsk_amd.TIMING_FFT, "hist": qpsk_amd.TIMING_HIST, "streams": qpsk_amd.TIMING_HIST}.get(what, qpsk_amd.TIMING_FIXED)
if what == "streams_fixed":
    what = "streams"
m = qpsk_amd.Modem(fs=bench.FS, rs=bench.RS, frame_size=bench.L, timing_mode=mode, fixed_index=bench.FIXED_INDEX)
if what == "fir_generic":
    m.tune(fir_generic=1)
x = bench.synth_frames_gpu(torch, dev, F, m.taps, seed=1)
sym = torch.empty((F, m.nsym), dtype=torch.uint8, device=dev)
fr = torch.empty((F,), dtype=torch.float32, device=dev)
ph = torch.empty_like(fr)
idx = torch.empty((F,), dtype=torch.int32, device=dev)
y = torch.empty_like(x) if what.startswith("fir") else None
if what.startswith("fir"):
    fn = lambda: m._check(m.L.qpsk_rrc_fir_batch(m.h, None, x.data_ptr(), y.data_ptr(), F, bench.L))
elif what == "fft_est":
    fn = lambda: m._check(m.L.qpsk_timing_fft_bin_batch(m.h, x.data_ptr(), F, idx.data_ptr(), None, None))
elif what == "scan":
    fn = lambda: m._check(m.L.qpsk_timing_scan_batch(m.h, x.data_ptr(), F, idx.data_ptr(), None))
elif what == "streams":
    m.streams_reset(F, 1500.0)
    pcm = (x[:, :, 0] * 8000.0).clamp(-32767, 32767).to(torch.int16).contiguous()
    fn = lambda: m._check(m.L.qpsk_streams_rx_pcm(m.h, pcm.data_ptr(), sym.data_ptr(), fr.data_ptr(), ph.data_ptr(), None, idx.data_ptr()))
else:
    fn = lambda: m.rx_batch_raw(x, F, sym, fr, ph)
t0 = time.time()
n = 0
last = 0.0
while n == 0 or time.time() - t0 < secs:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(per):
        fn()
    e1.record()
    torch.cuda.synchronize()
    n += per
    last = e0.elapsed_time(e1) / per
m.sync()
print("child: %s, %d frames x %d: %.4f ms per call (last %d of %d)" % (what, F, bench.L, last, per, n), flush=True)
