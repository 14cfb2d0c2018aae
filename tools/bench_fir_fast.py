#!/usr/bin/env python3
"""Measurement helper (GPU box): qpsk_rrc_fir_batch (exact, rrc_fir_kernel) against qpsk_rrc_fir_batch_fast (overlap-save,
512-point fp32 FFTs) on the config-2 block, 4096 x 16384 samples; the error of the fast one in the same run."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
import qpsk_amd  # noqa: E402

dev = torch.device("cuda", 0)
F = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m = qpsk_amd.Modem(fs=bench.FS, rs=bench.RS, frame_size=bench.L, timing_mode=qpsk_amd.TIMING_FIXED, fixed_index=6)
x = bench.synth_frames_gpu(torch, dev, F, m.taps, seed=3)
y0, y1 = torch.empty_like(x), torch.empty_like(x)
mg = qpsk_amd.Modem(fs=bench.FS, rs=bench.RS, frame_size=bench.L, timing_mode=qpsk_amd.TIMING_FIXED, fixed_index=6)
mg.tune(fir_generic=1)
y2 = torch.empty_like(x)
for name, fn, y, m_ in (("qpsk_rrc_fir_batch (exact, rrc_fir_stream_kernel)", m.L.qpsk_rrc_fir_batch, y0, m),
                        ("qpsk_rrc_fir_batch (exact, compiler-scheduled rrc_fir_kernel)", m.L.qpsk_rrc_fir_batch, y2, mg),
                        ("qpsk_rrc_fir_batch_fast (overlap-save)", m.L.qpsk_rrc_fir_batch_fast, y1, m)):
    for _ in range(5):
        fn(m_.h, None, x.data_ptr(), y.data_ptr(), F, bench.L)
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn(m_.h, None, x.data_ptr(), y.data_ptr(), F, bench.L)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    t = float(np.median(ts))
    print("%-62s %d x %d samples: %.3f ms -> %.0f Msamples/s, %.0f GB/s read + written, %.1f unfused TFLOP/s at 508 per sample" % (
        name, F, bench.L, t, F * bench.L / t / 1e3, 16.0 * F * bench.L / t / 1e6, 508.0 * F * bench.L / t / 1e9))
print("the two exact kernels agree bit for bit:", bool(torch.equal(y0.view(torch.int32), y2.view(torch.int32))))
err = (y1 - y0).abs().amax(dim=(1, 2)) / y0.abs().amax(dim=(1, 2))
print("fast against exact: max error %.2e of the frame's peak (worst of %d frames)" % (float(err.max()), F))
