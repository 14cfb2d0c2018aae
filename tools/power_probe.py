#!/usr/bin/env python3
"""Measurement helper (GPU box): board power and shader clock (rocm-smi) while one receive-kernel shape runs back to back.

    python tools/power_probe.py [frames[:pipe_dbg[:key=val,key=val...]] ...]      (default: 4096 8192; keys as Modem.tune)
    python tools/power_probe.py --cmd "<program and arguments>" ...                 (any GPU program running for ~6 s)
A child process launches the kernel in a loop for ~6 s per shape; the parent samples rocm-smi once a second.
With QPSK_HIP_LIB=qpsk_amd/libqpsk_hip_prof.so (measurement build) the pipe_dbg bits 1 / 16384 / 32768 select rx_lean_kernel's
streams without the filter arithmetic / the window reads / the flush arithmetic: their times at the power limit price the parts."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time
sys.path.insert(0, %r)
import torch, bench, qpsk_amd
frames = int(sys.argv[1]); secs = float(sys.argv[2]); dbg = int(sys.argv[3]); tune = sys.argv[4]
dev = torch.device("cuda", 0)
m = qpsk_amd.Modem(fs=bench.FS, rs=bench.RS, frame_size=bench.L, timing_mode=qpsk_amd.TIMING_FIXED, fixed_index=6)
x = bench.synth_frames_gpu(torch, dev, frames, m.taps, seed=1)
sym = torch.empty((frames, m.nsym), dtype=torch.uint8, device=dev)
fr = torch.empty((frames,), dtype=torch.float32, device=dev); ph = torch.empty_like(fr)
if dbg: m.tune(pipe_dbg=dbg)
kv = {k.split("=")[0]: int(k.split("=")[1], 0) for k in tune.split(",")} if tune else {}
pitch = kv.pop("pitch", 0)       # pitch=16448: frames 16384 + 64 samples apart (qpsk_rx_batch_pitched) instead of packed
if pitch:
    xp = torch.zeros((frames, pitch, 2), dtype=torch.float32, device=dev)
    xp[:, :bench.L] = x
    x = xp
if kv: m.tune(**kv)
print("ready", flush=True)
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(500):
        m.rx_batch_raw(x, frames, sym, fr, ph, pitch=pitch)
    e1.record(); torch.cuda.synchronize(); n += 500
    last = e0.elapsed_time(e1) / 500
print("child: %%d frames, dbg %%d %%s, %%s, %%.4f ms per launch (last 500), %%d launches" %% (frames, dbg, tune, m.last_kernel(), last, n), flush=True)
''' % ROOT


def amd_smi():
    """socket power, per-XCD clocks and the power-tracking (PPT) throttle activity from amd-smi, where it exists"""
    try:
        r = subprocess.run(["amd-smi", "metric", "-g", "0"], capture_output=True, text=True, timeout=30).stdout
    except Exception:      # noqa: BLE001
        return ""
    import re
    pw = re.search(r"SOCKET_POWER: (\d+) W", r)
    ppt = re.search(r"PPT_VIOLATION_ACTIVITY: (\S+ ?%?)", r)
    st = re.search(r"PPT_VIOLATION_STATUS: (\S+ ?\S*)", r)
    clk = [int(x) for x in re.findall(r"GFX_\d:\s+CLK: (\d+) MHz", r)]
    umc = re.search(r"UMC_ACTIVITY: (\d+) %", r)
    if not pw:
        return ""
    return " | amd-smi: %s W, XCD clocks %d-%d MHz (mean %d), PPT violation %s (%s), UMC activity %s %%" % (
        pw.group(1), min(clk) if clk else 0, max(clk) if clk else 0, sum(clk) // max(1, len(clk)), st.group(1).strip() if st else "?",
        ppt.group(1).strip() if ppt else "?", umc.group(1) if umc else "?")


def smi():
    out = []
    for args in (["--showpower"], ["--showclocks"]):
        try:
            r = subprocess.run(["rocm-smi"] + args, capture_output=True, text=True, timeout=20)
            for ln in r.stdout.splitlines():
                if any(k in ln for k in ("Package Power", "sclk")):
                    out.append(ln.split(":", 1)[1].strip() if ":" in ln else ln.strip())
        except Exception as e:      # noqa: BLE001
            out.append("rocm-smi %s: %s" % (args, e))
    return " | ".join(out)


print("idle:", smi(), flush=True)
args = sys.argv[1:]
if args and args[0] == "--cmd":      # any program that prints a line when it has finished: python tools/power_probe.py --cmd "/tmp/ubench_valu 6 2" ...
    for cmd in args[1:]:
        p = subprocess.Popen(cmd.split(), stdout=subprocess.PIPE, text=True)
        time.sleep(2.5)
        for i in range(2):
            print("%s, t+%ds: %s%s" % (cmd, i + 2, smi(), amd_smi() if i == 1 else ""), flush=True)
            time.sleep(0.5)
        print(p.stdout.read().strip(), flush=True)
        p.wait()
    sys.exit(0)
for arg in args or ["4096", "8192"]:
    parts = arg.split(":")
    frames, dbg, tune = int(parts[0]), int(parts[1] or "0", 0) if len(parts) > 1 else 0, parts[2] if len(parts) > 2 else ""
    p = subprocess.Popen([sys.executable, "-c", CHILD, str(frames), "7", str(dbg), tune], stdout=subprocess.PIPE, text=True)
    p.stdout.readline()
    time.sleep(1.5)
    for i in range(2):
        print("%d frames, dbg %d, t+%ds: %s%s" % (frames, dbg, i + 1, smi(), amd_smi() if i == 1 else ""), flush=True)
        time.sleep(0.5)
    print(p.stdout.read().strip(), flush=True)
    p.wait()
