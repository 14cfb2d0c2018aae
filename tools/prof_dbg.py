import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
os.environ["QPSK_HIP_LIB"] = os.path.join(ROOT, "qpsk_amd", "libqpsk_hip_prof.so")
import torch, bench, qpsk_amd
dev = torch.device("cuda", 0)
frames = 4096
m = qpsk_amd.Modem(fs=bench.FS, rs=bench.RS, frame_size=bench.L, timing_mode=qpsk_amd.TIMING_FIXED, fixed_index=6)
x = bench.synth_frames_gpu(torch, dev, frames, m.taps, seed=1)
sym = torch.empty((frames, m.nsym), dtype=torch.uint8, device=dev)
fr = torch.empty((frames,), dtype=torch.float32, device=dev)
ph = torch.empty_like(fr)
for dbg in sys.argv[1:]:
    os.environ["QPSK_PIPE_DBG"] = dbg
    print("==== dbg", dbg, flush=True)
    m.rx_batch_raw(x, frames, sym, fr, ph)
    torch.cuda.synchronize()
