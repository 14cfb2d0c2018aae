import os, sys
sys.path.insert(0, "/root/repo")
import torch, bench, qpsk_amd
dev = torch.device("cuda", 0)
for frames, tag in ((4096, "narrow"), (8192, "wide")):
    m = qpsk_amd.Modem(fs=bench.FS, rs=bench.RS, frame_size=bench.L, timing_mode=qpsk_amd.TIMING_FIXED, fixed_index=6)
    x = bench.synth_frames_gpu(torch, dev, frames, m.taps, seed=1)
    sym = torch.empty((frames, m.nsym), dtype=torch.uint8, device=dev)
    fr = torch.empty((frames,), dtype=torch.float32, device=dev); ph = torch.empty_like(fr)
    for dbg in (32,):
        os.environ["QPSK_PIPE_DBG"] = str(dbg)
        print("==== %s, QPSK_PIPE_DBG=%d" % (tag, dbg), flush=True)
        m.rx_batch_raw(x, frames, sym, fr, ph)
        torch.cuda.synchronize()
    os.environ.pop("QPSK_PIPE_DBG")
    del x
