#!/usr/bin/env python3
"""bench.py -- throughput of the QPSK receive hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames F]

One "step" = one pass of the fused RRC-FIR + Costas + slicer kernel (qpsk_rx_batch, fixed timing
offset) over one batch of synthetic frames that is already resident in HBM.
  N = 1: BASELINE.json configs[1], "Batch 4096 frames x 16384 complex samples, RRC FIR + Costas on 1 MI355X".
  N > 1: BASELINE.json configs[3], "65536 frames x 16384 samples sharded across 8 MI355X": 8192 frames per
         GPU, one process per GPU, every rank demodulating its own contiguous shard of the job's N x 8192
         independent frames (no data-path collective); the only communication is the barrier
         and the max-over-ranks of the elapsed time.
Ranks: under torch.distributed.run (RANK/LOCAL_RANK/WORLD_SIZE in the environment) this process IS one rank.
Started plainly with --gpus N > 1, it starts the N ranks itself as child processes -- before anything in
this process has touched the GPU -- relays rank 0's JSON line and exits non-zero if any rank failed.

Prints ONE JSON line on rank 0 (see the keys at the bottom).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FS, RS, L = 19200.0, 2400.0, 16384        # 2400 baud, 8x oversample, 16384 complex samples per frame
CYCLES = 8
FIXED_INDEX = 6                            # TX RRC (63) + RX RRC (63) group delay = 126 = 15*8 + 6
FRAMES_1GPU = 4096                         # BASELINE configs[1]
FRAMES_PER_GPU_SHARDED = 8192              # BASELINE configs[3]: 65536 frames over 8 GPUs
HBM_PEAK_GBS = 8000.0                      # MI355X HBM3E peak (MI355X_MICROARCH.md)
BYTES_PER_SAMPLE = 8                       # one complex float read per input sample


def synth_frames_gpu(torch, dev, nframes, taps, seed, offset_hz=50.0):
    """Synthetic QPSK frames built on the GPU (same recipe as tests/sigutil.make_frames: random dibits ->
    Gray map -> zero-stuff x8 -> TX RRC with the RX taps -> +50 Hz rotation), float32 (F, L, 2)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    nsym = L // CYCLES
    out = torch.empty((nframes, L, 2), dtype=torch.float32, device=dev)
    k =torch.from_numpy(np.asarray(taps, np.float32)[::-1].copy()).to(dev).view(1, 1, -1) * 1.85
    n = torch.arange(L, device=dev, dtype=torch.float64)
    ang = 2.0 * np.pi * offset_hz * n / FS
    cr, ci = torch.cos(ang).float(), torch.sin(ang).float()
    chunk = 512
    for f0 in range(0, nframes, chunk):
        f1 = min(nframes, f0 + chunk)
        s = torch.randint(0, 4, (f1 - f0, nsym), generator=g, device=dev)
        up = torch.zeros((f1 - f0, 2, L), dtype=torch.float32, device=dev)
        # Gray-coded constellation 0:(1,0) 1:(0,1) 2:(0,-1) 3:(-1,0)  (qpsk.c:58-63)
        up[:, 0, ::CYCLES] = (s == 0).float() - (s == 3).float()
        up[:, 1, ::CYCLES] = (s == 1).float() - (s == 2).float()
        y = torch.nn.functional.conv1d(torch.nn.functional.pad(up.reshape(-1, 1, L), (126, 0)), k).reshape(f1 - f0, 2, L)
        out[f0:f1, :, 0] = y[:, 0] * cr - y[:, 1] * ci
        out[f0:f1, :, 1] = y[:, 0] * ci + y[:, 1] * cr
    return out


def cpu_baseline(x_host, taps):
    """The reference's CPU path timed on this box's host cores, on a bounded sample of the same frames.
    kind "reference": oracle/_ref (the untouched reference compiled with its own Makefile flags, one core,
    consecutive rx_frame() calls as in its main loop, qpsk.c:344-354) when that library travelled here;
    otherwise kind "port": the oracle restatement (-O2), one core."""
    from oracle.pyoracle import Oracle, Reference, TIMING_HIST, ref_available
    nsamp = x_host.shape[0] * x_host.shape[1]
    out = {}
    if ref_available("c1"):
        ref = Reference("c1")
        ref.reset()
        t0 = time.perf_counter()
        for f in range(x_host.shape[0]):
            ref.rx_cplx(x_host[f])
        dt = time.perf_counter() - t0
        out = dict(value=nsamp / dt / 1e6, unit="Msamples/s", cores=1, kind="reference",
                   sample="%d frames x %d samples, consecutive rx_frame() calls on oracle/_ref (gcc -std=c11, no -O, as the reference Makefile)" % x_host.shape[:2])
    orc = Oracle()
    t0 = time.perf_counter()
    orc.rx_batch(x_host, FS, RS, timing_mode=TIMING_HIST, threads=1)
    dt1 = time.perf_counter() - t0
    try:
        ncores = len(os.sched_getaffinity(0))   # the cores this process may use (the box's CPU share)
    except AttributeError:
        ncores = os.cpu_count() or 1
    t0 = time.perf_counter()
    orc.rx_batch(x_host, FS, RS, timing_mode=TIMING_HIST, threads=ncores)
    dtn = time.perf_counter() - t0
    port = dict(port_1core_msps=nsamp / dt1 / 1e6, port_allcores_msps=nsamp / dtn / 1e6, port_cores=ncores,
                port_flags="gcc -O2 -ffp-contract=off, full-rate FIR + histogram timing + Costas (the reference's work)")
    if not out:
        out = dict(value=port["port_1core_msps"], unit="Msamples/s", cores=1, kind="port",
                   sample="%d frames x %d samples, oracle restatement" % x_host.shape[:2])
    out.update(port)
    return out


def launch_ranks(n, argv):
    """Start n ranks of this script as child processes (plain subprocess, never an exec of a process that has
    initialised the GPU), relay rank 0's single JSON line; returns the exit status (non-zero if any rank failed)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        print("bench.py: ranks failed: %s" % bad, file=sys.stderr)
        return 1
    return 0


def note(msg):
    if os.environ.get("QPSK_BENCH_VERBOSE"):
        print("[bench] " + msg, file=sys.stderr, flush=True)


def main():
    global L
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs of the job (default: WORLD_SIZE, else 1)")
    # a step is 0.2 ms: 20 steps are over before the clocks have settled (0.210 ms/step against 0.198 over 200)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--frames", type=int, default=None,
                    help="frames per GPU per step (default: 4096 = config 2 on one GPU, 8192 = config 4's per-GPU share on several)")
    ap.add_argument("--cpu-frames", type=int, default=2048,
                    help="frames of the CPU baseline sample (0 = skip); 2048 = half a batch, ~12 s of the reference on one core")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--settle", type=float, default=0.25, help="seconds of untimed launches before the warmup steps (GPU clocks)")
    ap.add_argument("--frame-size", type=int, default=L, help="complex samples per frame (config 2: 16384)")
    args = ap.parse_args()
    L = args.frame_size

    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus is None:
        args.gpus = int(env_world) if env_world else 1
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if env_world is None and args.gpus > 1:
        # nothing has touched the GPU yet (no torch, no libqpsk_hip): start the ranks as children
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    if env_world is not None and int(env_world) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%s" % (args.gpus, env_world))
    if args.frames is None:
        args.frames = FRAMES_1GPU if args.gpus == 1 else FRAMES_PER_GPU_SHARDED
    # ONE line on stdout: whatever the libraries of a rank print there (gloo announces its connections) goes to stderr
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import qpsk_amd

    from qpsk_amd.shard import (env_rank_world, init_distributed, local_device, max_over_ranks, shard_range,
                                sum_over_ranks)
    rank, local, world = env_rank_world()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libqpsk_hip has no CPU path")
    ndev = torch.cuda.device_count()
    local = local_device(local, ndev)             # one GPU per rank; ranks share only when the box has fewer GPUs
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # nccl == RCCL on ROCm (one GPU per rank); ranks that share a GPU (rehearsal on a smaller box) meet over gloo
    shared = world > ndev
    dist = init_distributed("gloo" if shared else "nccl", None if shared else dev)   # None when WORLD_SIZE == 1
    rdev = None if shared else dev
    if not os.path.exists(qpsk_amd.lib_path()):
        if rank == 0:
            qpsk_amd.build()
        if dist:
            dist.barrier()

    # the job is world * args.frames independent frames; this rank demodulates its contiguous shard of them
    lo, hi = shard_range(world * args.frames, rank, world)
    F = hi - lo
    m = qpsk_amd.Modem(fs=FS, rs=RS, frame_size=L, timing_mode=qpsk_amd.TIMING_FIXED, fixed_index=FIXED_INDEX,
                       device=local)
    note("context ready, generating %d frames" % F)
    x = synth_frames_gpu(torch, dev, F, m.taps, seed=1000 + rank)
    torch.cuda.synchronize()
    note("frames ready")
    sym = torch.empty((F, m.nsym), dtype=torch.uint8, device=dev)
    freq = torch.empty((F,), dtype=torch.float32, device=dev)
    phase = torch.empty((F,), dtype=torch.float32, device=dev)

    def barrier():
        if dist:
            dist.barrier()

    # clock settle, before the W warmup steps and outside every count: a step is 0.2-0.3 ms, so W + K = 25 steps are
    # over in 5 ms, before the GPU has left its idle clocks (the same 20 steps read 6-8 % longer cold than after
    # 0.1 s of work).  The timed region below is still exactly K steps after exactly W warmup steps.
    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < args.settle:
        for _ in range(10):
            m.rx_batch_raw(x, F, sym, freq, phase)
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        m.rx_batch_raw(x, F, sym, freq, phase)
    torch.cuda.synchronize()
    note("warmup done")
    barrier()
    torch.cuda.synchronize()
    # a step is ONE launch of the dominant kernel, so its average duration is the span of the timed region on the
    # launch stream (= torch's current stream) over K: HIP events around the K back-to-back launches
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        m.rx_batch_raw(x, F, sym, freq, phase)
    ev1.record()
    torch.cuda.synchronize()
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0, dist, rdev)
    kernel_ms = ev0.elapsed_time(ev1) / args.steps
    m.sync()      # raises if a kernel's in-LDS pipeline gave up (bounded spins): such a run has no valid timing
    joined = int(round(sum_over_ranks(1, dist, rdev)))      # ranks that really took part

    # cross-check, outside the timed region: one event pair per launch (each pair adds its own ~2 us)
    nev = min(args.steps, 50)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nev)]
    for a, b in evs:
        a.record()
        m.rx_batch_raw(x, F, sym, freq, phase)
        b.record()
    torch.cuda.synchronize()
    kernel_ms_pairs = float(np.mean([a.elapsed_time(b) for a, b in evs]))

    if rank != 0:
        if dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    samples_per_step = world * F * L
    value = samples_per_step * args.steps / elapsed / 1e6
    achieved = BYTES_PER_SAMPLE * F * L / (kernel_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):   # PMC bytes per launch of this shape's kernel (tools/collect_profiles.py), if it was profiled
        try:
            tj = json.load(open(tpath))
            tj = tj.get("shapes", {}).get("%dx%d" % (F, L), tj)
            if tj.get("frames") == F and tj.get("frame_size") == L:
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    kernel_name = "rx_pipe2_kernel" if F > 16 * torch.cuda.get_device_properties(dev).multi_processor_count else "rx_fused_pipe_kernel"
    res = {
        "metric": "complex Msamples/s demodulated + % HBM roofline, 2400-baud RRC+Costas path",
        "value": value, "unit": "Msamples/s", "n_gpus": joined, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic", "clock_settle_s": args.settle,
        "config": {"workload": "batch %d frames x %d complex samples per GPU, 2400 baud, 8x oversample, fused RRC FIR + Costas + slicer, fixed timing offset %d (%s)" % (
                       F, L, FIXED_INDEX, "BASELINE configs[1]" if (F, L) == (4096, 16384) else
                       "BASELINE configs[3] per-GPU share" if (F, L) == (8192, 16384) else "non-BASELINE shape"),
                   "frames_per_gpu": F, "frames_total": world * F, "gpus_visible_per_rank_box": ndev, "frame_size": L, "fs": FS, "rs": RS, "loop_bw": "TAU/100", "sharding": "independent frames per GPU, no collective"},
        "roofline": {"bound": "hbm", "kernel": kernel_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel_ms": kernel_ms,
                     "kernel_ms_event_pair_per_launch": kernel_ms_pairs,
                     "algorithmic_bytes_per_launch": BYTES_PER_SAMPLE * F * L},
    }
    if world == 1:          # the CPU baseline is reported at N = 1 only
        ncpu = min(args.cpu_frames, F)
        if ncpu > 0:
            xh = x[:ncpu].cpu().numpy()
            res["cpu_baseline"] = cpu_baseline(xh, m.taps)
    if True:                # parity gate of the same run (rank 0's shard)
        if not args.no_parity:
            from oracle.pyoracle import Oracle, TIMING_FIXED
            npar = min(256, F)
            xh = x[:npar].cpu().numpy()
            want = Oracle().rx_batch(xh, FS, RS, timing_mode=TIMING_FIXED, fixed_index=FIXED_INDEX)
            gs, gf, gp = sym[:npar].cpu().numpy(), freq[:npar].cpu().numpy(), phase[:npar].cpu().numpy()
            res["parity"] = {"frames_checked": npar, "symbol_mismatches": int(np.sum(gs != want["sym"])),
                             "freq_bit_mismatches": int(np.sum(gf.view(np.uint32) != want["freq"].view(np.uint32))),
                             "phase_bit_mismatches": int(np.sum(gp.view(np.uint32) != want["phase"].view(np.uint32))),
                             "mean_freq_hz": float(np.mean(gf.astype(np.float64) * RS / (2 * np.pi)))}
    print(json.dumps(res), file=real_stdout, flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
