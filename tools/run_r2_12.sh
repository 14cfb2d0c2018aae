cd $GRAFT_REPO_ROOT
python - <<'PY' > gpurun_out/r2_prof8.log 2>&1
import os, sys
sys.path.insert(0, ".")
for lib in ("qpsk_amd/libqpsk_hip_prof.so", "qpsk_amd/libqpsk_hip_prof_a2f6.so"):
    import subprocess
    code = r'''
import os, sys
sys.path.insert(0, ".")
os.environ["QPSK_HIP_LIB"] = %r
import torch, bench, qpsk_amd
dev = torch.device("cuda", 0)
frames = 6 * 256
m = qpsk_amd.Modem(fs=bench.FS, rs=bench.RS, frame_size=bench.L, timing_mode=qpsk_amd.TIMING_FIXED, fixed_index=6)
x = bench.synth_frames_gpu(torch, dev, frames, m.taps, seed=1)
sym = torch.empty((frames, m.nsym), dtype=torch.uint8, device=dev); fr = torch.empty((frames,), dtype=torch.float32, device=dev); ph = torch.empty_like(fr)
for g, lo in ((6, 0x00111), (2, 0x00001), (4, 0x00002)):
    m.tune(pipe_v=2, pipe_g=g, pipe_layout_lo=lo, pipe_layout_hi=0)
    for _ in range(3): m.rx_batch_raw(x, frames, sym, fr, ph)
    torch.cuda.synchronize()
    m.tune(pipe_dbg=32)
    print("==== %%s G=%%d layout %%#x" %% (%r, g, lo), flush=True)
    m.rx_batch_raw(x, frames, sym, fr, ph)
    torch.cuda.synchronize()
    m.tune(pipe_dbg=0)
''' % (lib, lib)
    print(subprocess.run([sys.executable, "-c", code], capture_output=True, text=True).stdout, flush=True)
PY
grep -v amdgpu gpurun_out/r2_prof8.log
