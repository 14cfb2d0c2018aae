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


def tx_frames_gpu(torch, dev, qpsk_amd, nframes, seed, offset_hz=50.0, local=0, fs=None, rs=None, frame_size=None):
    """The same stimulus from the library's own transmit chain (qpsk.c:225-285): one transmitter per frame,
    random dibits -> qpsk_tx_symbols() = Gray map, zero-stuffing and TX RRC shaping in tx_shape_kernel -> its complex baseband
    output, rotated by the +50 Hz carrier offset the reference tests (qpsk.c:320 against 342).  N2 at the scale it was written
    for: 1 GiB of frames per GPU built in place from 16 MB of symbols, nothing staged from the host.  Returns (F, L, 2) float32."""
    fs, rs, fl = fs or FS, rs or RS, frame_size or L
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    nsym = fl // int(fs / rs)
    mt = qpsk_amd.Modem(fs=fs, rs=rs, frame_size=fl, timing_mode=qpsk_amd.TIMING_FIXED, fixed_index=FIXED_INDEX, device=local)
    mt.tx_reset(nframes, 0.0)
    sym = torch.randint(0, 4, (nframes, nsym), generator=g, device=dev, dtype=torch.uint8)
    out = mt.tx_symbols(sym, want_pcm=False, want_baseband=True)["baseband"]
    mt.sync()
    mt.close()
    n = torch.arange(fl, device=dev, dtype=torch.float64)
    ang = 2.0 * np.pi * offset_hz * n / fs
    cr, ci = torch.cos(ang).float(), torch.sin(ang).float()
    chunk = max(1, (512 * 16384) // fl)
    for f0 in range(0, nframes, chunk):
        f1 = min(nframes, f0 + chunk)
        re, im = out[f0:f1, :, 0].clone(), out[f0:f1, :, 1].clone()
        out[f0:f1, :, 0] = re * cr - im * ci
        out[f0:f1, :, 1] = re * ci + im * cr
    return out


def cpu_baseline(x_host, taps):
    """The reference's CPU path timed on this box's host cores, on a bounded sample of the same frames.
    `value` = the build's own C restatement of the path (oracle/, proven bit-equal to the reference), gcc -O2 -ffp-contract=off,
    ONE core -- what travels with the repository and is directly comparable to the single-threaded reference; kind "port".
    Beside it: the same on all host cores (one independent frame per thread, core count stated), and -- when oracle/_ref travelled
    here (a git-ignored build product of the build container) -- the untouched reference itself at its Makefile's flags (no -O), one
    core, consecutive rx_frame() calls as in its main loop (qpsk.c:344-354), as `reference_makefile_flags_msps`."""
    from oracle.pyoracle import Oracle, Reference, TIMING_HIST, ref_available
    nsamp = x_host.shape[0] * x_host.shape[1]
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
    out = dict(value=nsamp / dt1 / 1e6, unit="Msamples/s", cores=1, kind="port",
               sample="%d frames x %d samples of the timed batch; oracle restatement: full-rate rrc_fir() + histogram timing + Costas + slicer "
                      "(the reference's work per rx_frame() call)" % x_host.shape[:2],
               flags="gcc -O2 -ffp-contract=off (results are optimisation-level independent once contraction is off: tests/test_oracle_vs_ref.py)",
               port_1core_msps=nsamp / dt1 / 1e6, port_allcores_msps=nsamp / dtn / 1e6, port_cores=ncores)
    have_ref = bool(ref_available("c1"))
    if have_ref:
        nref = min(x_host.shape[0], 512)        # -O0: 2.9 Msamples/s, 512 frames = 3 s
        ref = Reference("c1")
        ref.reset()
        t0 = time.perf_counter()
        for f in range(nref):
            ref.rx_cplx(x_host[f])
        dt = time.perf_counter() - t0
        out["reference_makefile_flags_msps"] = nref * x_host.shape[1] / dt / 1e6
        out["reference_sample"] = "%d frames, oracle/_ref (the untouched reference, gcc -std=c11, no -O, as its Makefile), 1 core" % nref
    out["ref_so"] = "present" if have_ref else "absent (oracle/_ref is built only where /root/reference exists)"
    return out


def launch_ranks(n, argv):
    """Start n ranks of this script as child processes (plain subprocess, never an exec of a process that has
    initialised the GPU), relay rank 0's single JSON line; returns the exit status (non-zero if any rank failed).
    All children are polled: the first one to fail ends the others at once (they would otherwise sit in the
    rendezvous until its timeout), and a bind that lost the race for its port is retried once."""
    import socket
    import subprocess
    import tempfile
    for attempt in range(2):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        procs, outs = [], []
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port))
            out = tempfile.TemporaryFile() if r == 0 else subprocess.DEVNULL
            err = tempfile.TemporaryFile()
            outs.append((out, err))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                          stdout=out, stderr=err))
        rcs = [None] * n
        while any(rc is None for rc in rcs):
            for r, p_ in enumerate(procs):
                if rcs[r] is None:
                    rcs[r] = p_.poll()
            if any(rc not in (None, 0) for rc in rcs):
                # a rank failed: the others get a moment to fail by themselves (the same cause, usually), then go
                t_grace = time.time() + 3.0
                for r, p_ in enumerate(procs):       # our own children, by handle
                    if rcs[r] is None:
                        try:
                            rcs[r] = p_.wait(timeout=max(0.0, t_grace - time.time()))
                        except subprocess.TimeoutExpired:
                            p_.kill()
                            p_.wait()
                            rcs[r] = -9
                break
            time.sleep(0.05)
        errs = []
        for r, (out, err) in enumerate(outs):
            err.seek(0)
            errs.append(err.read().decode(errors="replace"))
            err.close()
        sys.stderr.write("".join(errs))
        outs[0][0].seek(0)
        line = outs[0][0].read().decode()
        outs[0][0].close()
        if attempt == 0 and any(rc != 0 for rc in rcs) and any("EADDRINUSE" in e or "Address already in use" in e for e in errs):
            continue
        sys.stdout.write(line)
        sys.stdout.flush()
        bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (0, -9)] or [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
        if bad:
            print("bench.py: ranks failed: %s" % bad, file=sys.stderr)
            return 1
        return 0
    return 1


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
    ap.add_argument("--cpu-frames", type=int, default=4096,
                    help="frames of the CPU baseline sample (0 = skip); 4096 = the whole config-2 batch: ~4 s of the -O2 port on one core, "
                         "then all cores, then 512 frames of the -O0 reference build when it is present")
    ap.add_argument("--no-config5", action="store_true",
                    help="N = 1, config 2 only: skip BASELINE configs[4] (1200 baud, 2^20 samples per frame, 11 loop bandwidths; key config5)")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-shard", action="store_true",
                    help="N = 1, config 2 only: skip the extra measurement of the 8192-frame per-GPU share (key shard_8192)")
    ap.add_argument("--no-gather", action="store_true",
                    help="N = 1: skip the result-gather measurement (key gather: 8192 frames through the MULTI host layer, copy-back serial / overlapped)")
    ap.add_argument("--no-sustained", action="store_true",
                    help="N = 1: skip the steady-state figures (sustained_ms_per_step: >= 2 s back to back, then 200 launches; 3 s per shape)")
    ap.add_argument("--no-streams", action="store_true",
                    help="N = 1, config 2 only: skip the extra measurements of the streaming mode (4096 running PCM streams per block; one rx_frame() block per call)")
    ap.add_argument("--no-timing-modes", action="store_true",
                    help="N = 1, config 2 only: skip the extra measurements of the same batch with the FFT timing estimate in front "
                         "(BASELINE configs[2], key config3) and with the reference's histogram estimate (key hist)")
    ap.add_argument("--stimulus", choices=["tx", "synth"], default="tx",
                    help="tx: frames from the library's own transmit chain (qpsk_tx_symbols, N2); synth: torch conv1d (rounds 1-3)")
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

    from qpsk_amd.shard import (device_identity, distinct_devices, env_rank_world, gather_identities, init_distributed,
                                local_device, max_over_ranks, shard_range, sum_over_ranks)
    rank, local, world = env_rank_world()
    # The control plane (barrier, two scalar reductions, the ranks' device identities) is host data over gloo whatever
    # the number of GPUs: the data path has no collective (north_star), so no rank creates an RCCL communicator and
    # the N = 8 run executes the control code that the two-rank rehearsal on a one-GPU box and the CPU tests execute.
    dist = init_distributed("gloo")               # None when WORLD_SIZE == 1; before anything touches the GPU
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libqpsk_hip has no CPU path")
    ndev = torch.cuda.device_count()
    local = local_device(local, ndev)             # one GPU per rank; ranks share only when the box has fewer GPUs
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    idents = gather_identities(device_identity(torch, local), dist)
    if not os.path.exists(qpsk_amd.lib_path()):
        if rank == 0:
            qpsk_amd.build()
        if dist:
            dist.barrier()

    def barrier():
        if dist:
            dist.barrier()

    def make_batch(F, seed, mode=qpsk_amd.TIMING_FIXED):
        m_ = qpsk_amd.Modem(fs=FS, rs=RS, frame_size=L, timing_mode=mode, fixed_index=FIXED_INDEX,
                            device=local)
        if args.stimulus == "tx":
            x_ = tx_frames_gpu(torch, dev, qpsk_amd, F, seed=seed, local=local)
        else:
            x_ = synth_frames_gpu(torch, dev, F, m_.taps, seed=seed)
        torch.cuda.synchronize()
        return m_, x_, (torch.empty((F, m_.nsym), dtype=torch.uint8, device=dev),
                        torch.empty((F,), dtype=torch.float32, device=dev),
                        torch.empty((F,), dtype=torch.float32, device=dev))

    def timed_region(m_, x_, F, outs, steps, warmup, settle):
        """W untimed warmup steps, then EXACTLY K steps between barrier + synchronize on both sides.  Returns (wall
        seconds of the K steps on this rank: from leaving the leading barrier + synchronize to the return of the
        synchronize behind the last step; kernel ms per launch from HIP events on the launch stream)."""
        sym_, freq_, phase_ = outs
        # clock settle, before the W warmup steps and outside every count: a step is 0.2-0.3 ms, so W + K = 25 steps
        # are over in 5 ms, before the GPU has left its idle clocks
        t_settle = time.perf_counter()
        while time.perf_counter() - t_settle < settle:
            for _ in range(10):
                m_.rx_batch_raw(x_, F, sym_, freq_, phase_)
            torch.cuda.synchronize()
        for _ in range(warmup):
            m_.rx_batch_raw(x_, F, sym_, freq_, phase_)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        # a step is ONE launch of the dominant kernel, so its average duration is the span of the timed region on the
        # launch stream (= torch's current stream, which the context enqueues on) over K
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(steps):
            m_.rx_batch_raw(x_, F, sym_, freq_, phase_)
        ev1.record()
        torch.cuda.synchronize()
        # this rank's clock stops HERE, when its K steps are done; the job's time is the MAX over ranks of these (the slowest
        # rank defines the job).  The trailing barrier still brackets the region, but it is not timed: a gloo rendezvous costs
        # 0.4-0.8 ms (profiles/r04_bench_gpus2_shared.json: 0.76 ms per region), i.e. 13 % of a 20-step region of 0.26 ms steps --
        # a tax the N = 1 line (no barrier) does not pay, which would have made a perfect 8x read 6.9x.
        dt = time.perf_counter() - t0
        barrier()
        m_.sync()      # raises if a kernel's in-LDS pipeline gave up (bounded spins): such a run has no valid timing
        return dt, ev0.elapsed_time(ev1) / steps

    def amd_smi_now():
        """socket power, PPT (power tracking) throttle activity and the XCD clocks from amd-smi while the GPU is busy; {} where it is not
        there or not readable (an ordinary user on the GPU box can read it)"""
        import re
        import subprocess
        try:
            r = subprocess.run(["amd-smi", "metric", "-g", str(local)], capture_output=True, text=True, timeout=30).stdout
        except Exception:      # noqa: BLE001
            return {}
        pw = re.search(r"SOCKET_POWER: (\d+) W", r)
        ppt = re.search(r"PPT_VIOLATION_ACTIVITY: (\d+)", r)
        clk = [int(v) for v in re.findall(r"GFX_\d:\s+CLK: (\d+) MHz", r)]
        out = {}
        if pw:
            out["socket_power_w"] = int(pw.group(1))
        if ppt:
            out["ppt_active_pct"] = int(ppt.group(1))
        if clk:
            out["mean_clock_mhz"] = float(np.mean(clk))
        return out

    def sustained_region(m_, x_, F, outs, seconds=2.0, last=200):
        """The same step back to back for at least `seconds` (the board reaches its power limit after tens of milliseconds; the 20-step
        region the driver times is over in 3-5 ms: a BURST figure), then the last `last` launches between two HIP events; the power
        state is read while a further queue of launches keeps the GPU busy.  -> dict(ms_per_step, launches, seconds, socket_power_w, ...)"""
        sym_, freq_, phase_ = outs
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < seconds:
            for _ in range(200):
                m_.rx_batch_raw(x_, F, sym_, freq_, phase_)
            torch.cuda.synchronize()
            n += 200
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(last):
            m_.rx_batch_raw(x_, F, sym_, freq_, phase_)
        ev1.record()
        for _ in range(3000):          # 0.5-0.8 s of queued work: the power state below is read while the GPU runs it
            m_.rx_batch_raw(x_, F, sym_, freq_, phase_)
        smi = amd_smi_now()
        torch.cuda.synchronize()
        m_.sync()
        d = {"ms_per_step": ev0.elapsed_time(ev1) / last, "timed_launches": last, "launches_before": n,
             "seconds_before": time.perf_counter() - t0,
             "what": "back to back for >= %.0f s, then %d launches between two HIP events: the steady-state step at the board's power limit; "
                     "ms_per_step above is the %d-step region of the contract" % (seconds, last, args.steps)}
        d.update(smi)
        return d

    def hz_check(freq_):
        """every frame of the batch must have locked on the +50 Hz carrier of the stimulus (qpsk.c:217)"""
        hz = freq_.double() * RS / (2 * np.pi)
        return int(freq_.numel()), int(((hz - 50.0).abs() >= 2.0).sum().item()), float(hz.mean().item())

    # the job is world * args.frames independent frames; this rank demodulates its contiguous shard of them
    lo, hi = shard_range(world * args.frames, rank, world)
    F = hi - lo
    note("generating %d frames" % F)
    m, x, outs = make_batch(F, seed=1000 + rank)
    sym, freq, phase = outs
    note("frames ready")
    dt, kernel_ms = timed_region(m, x, F, outs, args.steps, args.warmup, args.settle)
    elapsed = max_over_ranks(dt, dist)
    joined = int(round(sum_over_ranks(1, dist)))      # ranks that really took part
    kernel_name = m.last_kernel()                     # reported by the library, not inferred from the shape
    hz_n, hz_bad, hz_mean = hz_check(freq)
    hz_bad_all = int(round(sum_over_ranks(hz_bad, dist)))
    hz_n_all = int(round(sum_over_ranks(hz_n, dist)))

    # (round 5's "kernel_ms_event_pair_per_launch" -- one event pair per launch -- read 24 % above the span figure: it measured the
    # events' own cost, not the kernel; dropped.  The steady-state step below is the figure worth having beside the timed region.)
    sustained = sustained_region(m, x, F, outs) if world == 1 and not args.no_sustained else None

    if rank != 0:
        if dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    import hashlib
    lib_file = qpsk_amd.lib_path()
    lib_sha256 = hashlib.sha256(open(lib_file, "rb").read()).hexdigest()

    def traffic_of(F_):
        """PMC bytes per launch of this shape's kernel from an EARLIER rocprofv3 --pmc session of this command
        (profiles/traffic.json, tools/collect_profiles.py) -- a constant of the code version profiled, not a counter
        of this run; the source says which file and when."""
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        try:
            tj = json.load(open(tpath))
            tj = tj.get("shapes", {}).get("%dx%d" % (F_, L), tj)
            if tj.get("frames") == F_ and tj.get("frame_size") == L:
                sha_prof = tj.get("library_sha256")
                return tj.get("hbm_bytes_per_launch"), {
                    "file": "profiles/traffic.json", "from": tj.get("source"),
                    # the counters belong to ONE build of the library: the one whose hash the profiling session stored
                    "library_sha256_profiled": sha_prof,
                    "traffic_matches_library": (bool(sha_prof) and sha_prof == lib_sha256),
                    "file_mtime": time.strftime("%Y-%m-%d", time.gmtime(os.path.getmtime(tpath))),
                    "kernel_profiled": tj.get("kernel"),
                    "note": "PMC FETCH_SIZE x 2 + WRITE_SIZE of an earlier profiling session, not a counter of this run"}
        except Exception:
            pass
        return None, None

    def roofline_of(F_, kms, kname):
        ach = BYTES_PER_SAMPLE * F_ * L / (kms * 1e-3) / 1e9
        tr, src = traffic_of(F_)
        return {"bound": "hbm", "kernel": kname, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "traffic": tr, "traffic_source": src, "kernel_ms": kms,
                "algorithmic_bytes_per_launch": BYTES_PER_SAMPLE * F_ * L,
                # context, not a claim of this run: what a synthetic kernel with nothing but this path's operation mix (63.5
                # unfused flops per 8 input bytes on random data + the window reads) sustains on this part before it throttles
                "operation_mix_ceiling": {"input_gbs": [4500.0, 4800.0], "source": "profiles/r03_power_ceiling.txt "
                                          "(tools/ubench_valu_power.hip under tools/power_probe.py, an earlier session)"}}

    ndistinct = distinct_devices(idents)
    samples_per_step = world * F * L
    value = samples_per_step * args.steps / elapsed / 1e6
    rl = roofline_of(F, kernel_ms, kernel_name)
    res = {
        "metric": "complex Msamples/s demodulated + % HBM roofline, 2400-baud RRC+Costas path",
        "value": value, "unit": "Msamples/s", "n_gpus": ndistinct, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic", "clock_settle_s": args.settle,
        **({"sustained_ms_per_step": sustained["ms_per_step"],
            "sustained": dict(sustained, frac=BYTES_PER_SAMPLE * F * L / (sustained["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS)} if sustained else {}),
        "clock": "per rank: barrier + synchronize, K launches, synchronize -> stop; MAX over ranks; the trailing barrier is outside the clock",
        "stimulus": ("library transmit chain: random dibits -> qpsk_tx_symbols (tx_shape_kernel: Gray map, zero-stuffing, TX RRC) -> +50 Hz rotation"
                     if args.stimulus == "tx" else "torch: random dibits -> Gray map -> zero-stuffing -> conv1d with the RX taps -> +50 Hz rotation"),
        # n_gpus counts DISTINCT physical devices (host + PCI address of every rank, gathered); ranks that share a GPU
        # (a rehearsal of the launch path on a smaller box) show up as ranks > n_gpus and gpu_shared
        "ranks": joined, "gpu_shared": ndistinct < joined, "devices": idents,
        "control_plane": "gloo (host scalars only: barrier, max of elapsed, device identities); data path: no collective",
        "config": {"workload": "batch %d frames x %d complex samples per GPU, 2400 baud, 8x oversample, fused RRC FIR + Costas + slicer, fixed timing offset %d (%s)" % (
                       F, L, FIXED_INDEX, "BASELINE configs[1]" if (F, L) == (4096, 16384) else
                       "BASELINE configs[3] per-GPU share" if (F, L) == (8192, 16384) else "non-BASELINE shape"),
                   "frames_per_gpu": F, "frames_total": world * F, "gpus_visible_per_rank_box": ndev, "frame_size": L, "fs": FS, "rs": RS, "loop_bw": "TAU/100", "sharding": "independent frames per GPU, no collective",
                   # N = 1 is quoted on BASELINE configs[1] (4096 frames), N > 1 on configs[3]'s per-GPU share (8192 frames each): the single-GPU
                   # rate at THAT shape is the N = 1 line's shard_8192 key, the like-for-like base of a scaling curve
                   "single_gpu_rate_at_this_shape": ("this line's value" if world == 1 else "the N = 1 line's shard_8192.msamples_per_s")},
        "roofline": rl,
        "library": {"path": os.path.relpath(lib_file, ROOT) if lib_file.startswith(ROOT) else lib_file,
                    "sha256": lib_sha256,
                    "override_QPSK_HIP_LIB": bool(os.environ.get("QPSK_HIP_LIB")), "version": qpsk_amd.version()},
    }
    parity = {"hz_frames_checked": hz_n_all, "hz_out_of_range": hz_bad_all, "mean_freq_hz": hz_mean}
    if not args.no_parity:       # parity gate of the same run (rank 0's shard): bits against the oracle
        from oracle.pyoracle import Oracle, TIMING_FIXED
        # EVERY frame of the timed batch (round 4 checked 256 of 4096): the oracle's fixed-offset path is the decimating filter + the
        # loop, 56 Msamples/s on one core -- 16 threads take config 2's 67 M samples in a fraction of a second
        npar = F
        xh = x[:npar].cpu().numpy()
        try:
            nthr = min(16, len(os.sched_getaffinity(0)))
        except AttributeError:
            nthr = min(16, os.cpu_count() or 1)
        want = Oracle().rx_batch(xh, FS, RS, timing_mode=TIMING_FIXED, fixed_index=FIXED_INDEX, threads=nthr)
        del xh
        gs, gf, gp = sym[:npar].cpu().numpy(), freq[:npar].cpu().numpy(), phase[:npar].cpu().numpy()
        parity.update({"frames_checked": npar, "symbol_mismatches": int(np.sum(gs != want["sym"])),
                       "freq_bit_mismatches": int(np.sum(gf.view(np.uint32) != want["freq"].view(np.uint32))),
                       "phase_bit_mismatches": int(np.sum(gp.view(np.uint32) != want["phase"].view(np.uint32)))})
    res["parity"] = parity
    if world == 1:          # the CPU baseline is reported at N = 1 only
        ncpu = min(args.cpu_frames, F)
        if ncpu > 0:
            xh = x[:ncpu].cpu().numpy()
            res["cpu_baseline"] = cpu_baseline(xh, m.taps)
    if world == 1 and (F, L) == (FRAMES_1GPU, 16384) and not args.no_timing_modes:
        # The same resident batch with a timing ESTIMATE in front of the receive kernel, by the same timed-region code, after
        # config 2's region: key config3 = BASELINE configs[2] (the FFT timing estimate: timing_fft_kernel + receive kernel per
        # step), key hist = the reference's own histogram estimate (qpsk.c:127-180: timing_scan_kernel + receive kernel).  A step
        # is one qpsk_rx_batch call = two launches; frac is the batch's algorithmic bytes (8 B per sample) over the step.
        # (their own step counts, named in each key: a 20-step region of a 0.17 ms step is over in 3.5 ms, before the power controller has
        # settled -- +4 % on the same box; the contract's EXACTLY K steps is the headline's)
        for key, mode, steps_ in (("config3", qpsk_amd.TIMING_FFT, max(args.steps, 100)), ("hist", qpsk_amd.TIMING_HIST, max(args.steps // 4, 25))):
            mt = qpsk_amd.Modem(fs=FS, rs=RS, frame_size=L, timing_mode=mode, fixed_index=FIXED_INDEX, device=local)
            outs_t = (torch.empty_like(sym), torch.empty_like(freq), torch.empty_like(phase))
            idx_t = torch.full((F,), -1, dtype=torch.int32, device=dev)
            dtt, kmst = timed_region(mt, x, F, outs_t, steps_, max(1, args.warmup), args.settle)
            mt.rx_batch_raw(x, F, *outs_t, index=idx_t)
            mt.sync()
            ent = {"workload": "config 2's batch, %s in front of the fused receive kernel" % (
                       "FFT timing estimate (BASELINE configs[2]; rrc_fir() of 512 samples per frame + the symbol-rate bin of fft.c's transform)"
                       if key == "config3" else "the reference's histogram timing estimate (qpsk.c:127-180: full-rate rrc_fir() + scan)"),
                   "kernels": ([mt.last_kernel()] if ("inside the launch" in mt.last_kernel() or "rx_hist_kernel" in mt.last_kernel()) else
                               ["timing_fft_kernel" if key == "config3" else "timing_scan_kernel", mt.last_kernel()]),
                   "steps": steps_, "ms_per_step": dtt / steps_ * 1e3, "step_ms_events": kmst,
                   "msamples_per_s": F * L * steps_ / dtt / 1e6,
                   "frac_of_hbm_peak_on_8B_per_sample": BYTES_PER_SAMPLE * F * L / (kmst * 1e-3) / 1e9 / HBM_PEAK_GBS,
                   "index_histogram": [int(v) for v in torch.bincount(idx_t.clamp(min=0), minlength=CYCLES).tolist()]}
            if key == "config3":   # every clean frame's FFT estimate is the eye centre (= config 2's fixed offset), so the batch must
                                   # equal config 2's bit for bit; the reference's histogram index is an amplitude-bin number
                ent["symbols_equal_config2"] = bool(torch.equal(outs_t[0], sym))
                ent["freq_bits_equal_config2"] = bool(torch.equal(outs_t[1].view(torch.int32), freq.view(torch.int32)))
            if not args.no_parity:
                from oracle.pyoracle import Oracle, TIMING_FFT, TIMING_HIST
                npar_t = 32
                want_t = Oracle().rx_batch(x[:npar_t].cpu().numpy(), FS, RS, timing_mode=TIMING_FFT if key == "config3" else TIMING_HIST)
                ent["parity_frames_checked"] = npar_t
                ent["symbol_mismatches"] = int(np.sum(outs_t[0][:npar_t].cpu().numpy() != want_t["sym"]))
                ent["index_mismatches"] = int(np.sum(idx_t[:npar_t].cpu().numpy() != want_t["index"]))
            if key == "hist":
                # Round 6: the step above is the ONE-PASS route wherever the context's guess holds (rx_hist_kernel: the scan kernel's workgroup
                # runs the receive path on the previous batch's majority index; every frame of this batch sits on one index).  Beside it, the
                # same batch through the two-launch route of rounds 3-5 (timing_scan_kernel, then the receive kernel: the input read twice):
                mt.tune(hist_onepass=0)
                dt2l, kms2l = timed_region(mt, x, F, outs_t, steps_, max(1, args.warmup), 0.0)
                mt.tune(hist_onepass=None)
                ent["two_launch_route"] = {"kernels": ["timing_scan_kernel", mt.last_kernel()], "ms_per_step": dt2l / steps_ * 1e3, "step_ms_events": kms2l}
                # VERDICT r4 item 6, "histogram mode in one pass" (round 5's measurement of the OTHER one-read route, kept): the route that reads the batch ONCE -- full-rate filter + scan with the
                # filtered block written planar by decimation phase, then the loop kernel on the one plane the index picks (no second read
                # of the input, no decimating filter) -- exists as the streams' kernels (stream_scan_kernel MODE 0 + costas_pipe_kernel);
                # the same batch as 4096 one-block streams, steady state of blocks 2..5 (same kernels, same bytes per block; a stream's
                # loop runs a block behind, qpsk.c:186-197, so the symbols are not the batch's: the streams have their own gate below)
                mo = qpsk_amd.Modem(fs=FS, rs=RS, frame_size=L, timing_mode=qpsk_amd.TIMING_HIST, device=local)
                mo.streams_reset(F, 0.0)
                to = []
                for k_ in range(6):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    mo.streams_rx_cplx(x, want_costas=False)
                    e1.record()
                    torch.cuda.synchronize()
                    to.append(e0.elapsed_time(e1))
                mo.sync()
                ent["planar_write_route"] = {
                    "kernels": mo.last_kernel(), "ms_per_block": float(np.median(to[2:])),
                    "what": "filter + scan once, the filtered block written planar by decimation phase (512 MiB written), loop kernel on the picked plane",
                    "against": "two_launch_route of this key: the input read twice (timing_scan_kernel, then the fused receive kernel's decimating filter)",
                    "faster": bool(float(np.median(to[2:])) < dt2l / steps_ * 1e3)}
                mo.close()
            res[key] = ent
            mt.close()
            del outs_t, idx_t
    if world == 1 and (F, L) == (FRAMES_1GPU, 16384) and not args.no_timing_modes:
        # SURVEY 8(d)'s optional AWGN variant, reported separately (the reference has no channel model: qpsk.c:314-334): config 2's
        # batch plus complex white noise at Es/N0 = 12 dB after the matched filter (sigma per component from the measured mean sample
        # power), same kernel, same fixed offset.  The time is data-independent except through the loop's 2 pi wraps.
        ga = torch.Generator(device=dev)
        ga.manual_seed(4242)
        psig = float((x[:64].double() ** 2).sum(dim=2).mean().item())            # mean |sample|^2 (8 samples per symbol)
        esn0_db = 12.0
        sigma = (psig * CYCLES / (10.0 ** (esn0_db / 10.0)) / 2.0) ** 0.5        # Es = 8 x mean sample power; N0 / 2 per component
        xn = x.clone()
        for f0_ in range(0, F, 512):
            xn[f0_:f0_ + 512] += sigma * torch.randn(xn[f0_:f0_ + 512].shape, generator=ga, device=dev, dtype=torch.float32)
        outs_n = (torch.empty_like(sym), torch.empty_like(freq), torch.empty_like(phase))
        steps_n = max(args.steps, 100)
        dtn, kmsn = timed_region(m, xn, F, outs_n, steps_n, max(1, args.warmup), args.settle)
        hz_n = outs_n[1].double() * RS / (2 * np.pi)
        aw = {"workload": "config 2's batch + complex AWGN, Es/N0 = %.0f dB (sigma %.4f per component), fixed timing offset" % (esn0_db, sigma),
              "kernel": m.last_kernel(), "steps": steps_n, "ms_per_step": dtn / steps_n * 1e3, "step_ms_events": kmsn,
              "frac_of_hbm_peak_on_8B_per_sample": BYTES_PER_SAMPLE * F * L / (kmsn * 1e-3) / 1e9 / HBM_PEAK_GBS,
              "mean_freq_hz": float(hz_n.mean().item()), "freq_hz_std": float(hz_n.std().item()),
              "frames_within_2hz_of_50": int(((hz_n - 50.0).abs() < 2.0).sum().item())}
        if not args.no_parity:
            from oracle.pyoracle import Oracle, TIMING_FIXED
            want_n = Oracle().rx_batch(xn[:32].cpu().numpy(), FS, RS, timing_mode=TIMING_FIXED, fixed_index=FIXED_INDEX)
            aw["parity_frames_checked"] = 32
            aw["symbol_mismatches"] = int(np.sum(outs_n[0][:32].cpu().numpy() != want_n["sym"]))
            aw["freq_bit_mismatches"] = int(np.sum(outs_n[1][:32].cpu().numpy().view(np.uint32) != want_n["freq"].view(np.uint32)))
        res["awgn"] = aw
        del xn, outs_n
    if world == 1 and (F, L) == (FRAMES_1GPU, 16384) and not args.no_shard:
        # BASELINE configs[3]'s per-GPU share (8192 x 16384: the shape of every rank of an N > 1 run, and the one where
        # the filter, not the recurrence, is the limit) measured in the same run, after config 2's region and by the
        # same timed-region code.  `value`, `config` and `roofline` above stay config 2's.
        del x, outs, sym, freq, phase
        m.close()
        torch.cuda.empty_cache()
        F2 = FRAMES_PER_GPU_SHARDED
        m2, x2, outs2 = make_batch(F2, seed=2000)
        dt2, kms2 = timed_region(m2, x2, F2, outs2, args.steps, args.warmup, args.settle)
        n2, bad2, mean2 = hz_check(outs2[1])
        sh = roofline_of(F2, kms2, m2.last_kernel())
        if not args.no_sustained:
            su2 = sustained_region(m2, x2, F2, outs2)
            sh["sustained_ms_per_step"] = su2["ms_per_step"]
            sh["sustained"] = dict(su2, frac=BYTES_PER_SAMPLE * F2 * L / (su2["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS)
        sh.update({"frames": F2, "frame_size": L, "steps": args.steps, "warmup": args.warmup,
                   "ms_per_step": dt2 / args.steps * 1e3, "msamples_per_s": F2 * L * args.steps / dt2 / 1e6,
                   "hz_frames_checked": n2, "hz_out_of_range": bad2, "mean_freq_hz": mean2})
        if not args.no_parity:
            from oracle.pyoracle import Oracle, TIMING_FIXED
            xh = x2.cpu().numpy()              # every frame of this batch too
            want = Oracle().rx_batch(xh, FS, RS, timing_mode=TIMING_FIXED, fixed_index=FIXED_INDEX, threads=nthr)
            del xh
            sh["parity_frames_checked"] = F2
            sh["symbol_mismatches"] = int(np.sum(outs2[0].cpu().numpy() != want["sym"]))
            sh["freq_bit_mismatches"] = int(np.sum(outs2[1].cpu().numpy().view(np.uint32) != want["freq"].view(np.uint32)))
        if not args.no_gather:
            # The result gather of a sharded job ("copied back per device over PCIe and concatenated on the host") through the C
            # host layer include/qpsk_hip.h MULTI (one shard = this GPU, the same resident 8192-frame batch): a frame returns 2048 symbol
            # bytes + freq + phase = 16.1 MiB per step, ~0.27 ms over PCIe Gen5 x16 -- as long as the kernel.  Three schedules, wall
            # time per step on the host clock: serial (a step's copy-back is waited for before the next step is enqueued) and overlapped
            # (two result slots: step k + 1's kernel runs while step k is copied back), both with the library's pinned staging +
            # one concatenating memcpy per shard, and overlapped with the copy-back by DMA straight into the caller's pinned arrays.
            mj = qpsk_amd.MultiJob([local], fs=FS, rs=RS, frame_size=L, timing_mode=qpsk_amd.TIMING_FIXED, fixed_index=FIXED_INDEX)
            mj.load(total=F2)
            mj.use_device_input(0, x2)
            o_a, o_b = mj.outputs(), mj.outputs()
            mj.begin(0); mj.end(0, *o_a)                      # first call: allocation inside the library
            kg = max(10, args.steps)

            def serial():
                t0_ = time.perf_counter()
                for _ in range(kg):
                    mj.begin(0); mj.end(0, *o_a)
                return (time.perf_counter() - t0_) / kg * 1e3

            def overlapped(outs_):
                t0_ = time.perf_counter()
                mj.begin(0)
                for k_ in range(1, kg):
                    mj.begin(k_ & 1)
                    mj.end((k_ - 1) & 1, *outs_[(k_ - 1) & 1])
                mj.end((kg - 1) & 1, *outs_[(kg - 1) & 1])
                return (time.perf_counter() - t0_) / kg * 1e3

            # (each schedule once untimed first: the first copy into freshly allocated page-locked memory pays for its pages)
            serial()
            g_serial = serial()
            overlapped((o_a, o_b))
            g_over = overlapped((o_a, o_b))
            p_a, p_b = mj.pinned_outputs(), mj.pinned_outputs()
            mj.set_direct(0, *p_a); mj.set_direct(1, *p_b)
            none3 = (None, None, None)
            overlapped((none3, none3))
            g_direct = overlapped((none3, none3))
            # ... and with the symbols packed four to a byte on the device (qpsk_pack_symbols: 4 MiB instead of 16 per step)
            mj.set_direct(0); mj.set_direct(1)
            mj.set_packed(True)
            q_a, q_b = mj.pinned_outputs(), mj.pinned_outputs()
            mj.set_direct(0, *q_a); mj.set_direct(1, *q_b)
            overlapped((none3, none3))
            g_packed = overlapped((none3, none3))
            packed_ok = bool(np.array_equal(mj.unpack(q_a[0]), o_a[0]) and np.array_equal(q_b[0], q_a[0]))
            same = bool(np.array_equal(p_a[0], o_a[0]) and np.array_equal(p_b[0], o_b[0]) and np.array_equal(o_a[0], outs2[0].cpu().numpy()) and
                        np.array_equal(p_a[1].view(np.uint32), outs2[1].cpu().numpy().view(np.uint32)))
            gbytes = F2 * (L // 8) + 8 * F2
            res["gather"] = {"frames": F2, "frame_size": L, "steps": kg, "bytes_per_step": gbytes,
                             "kernel_ms_per_step": dt2 / args.steps * 1e3,
                             "ms_per_step_serial": g_serial, "ms_per_step_overlapped": g_over, "ms_per_step_overlapped_direct": g_direct,
                             "ms_per_step_overlapped_direct_packed": g_packed, "bytes_per_step_packed": F2 * (L // 32) + 8 * F2,
                             "packed_equals_unpacked": packed_ok,
                             "pcie_bound_ms": gbytes / 63e9 * 1e3, "pcie_bound_assumes": "PCIe Gen5 x16, 63 GB/s",
                             "gathered_equals_device_results": same,
                             "what": "qpsk_multi_rx_begin/end (one shard on this GPU): kernel + copy-back of symbols, freq, phase to host memory; "
                                     "serial / overlapped through pinned staging with a concatenating memcpy, overlapped_direct by DMA into the caller's pinned arrays, "
                                     "overlapped_direct_packed with the symbols four to a byte (qpsk_multi_set_packed)"}
            mj.close()
        res["shard_8192"] = sh
    if world == 1 and (args.frames, L) == (FRAMES_1GPU, 16384) and not args.no_config5:
        # BASELINE configs[4]: "1200-baud / 8x oversample long-frame variant, 1M samples/frame, Costas loop-BW sweep TAU/100-TAU/200" --
        # FS 9600 / RS 1200, 1,048,576 samples per frame, 11 loop bandwidths TAU/100, TAU/110, ... TAU/200 as independent Costas
        # chains over ONE filter pass (README.md:12; qpsk_rx_batch_bw).  384 frames (384 x 11 = 4224 chains >= 4096, 3 GiB
        # of input; the reference gives no count).  A step is one call; 131,072 serial loop steps per chain bound it by construction
        # (DESIGN.md 4.1.3), so the fraction of the HBM line is reported, not a target.  Gate: 5 frames x 11 loops against the oracle.
        torch.cuda.empty_cache()
        fs5, rs5, L5, F5 = 9600.0, 1200.0, 1 << 20, 384
        bws5 = [np.float32(2.0 * 3.14159265358979323846 / d) for d in range(100, 201, 10)]
        m5 = qpsk_amd.Modem(fs=fs5, rs=rs5, frame_size=L5, timing_mode=qpsk_amd.TIMING_FIXED, fixed_index=FIXED_INDEX, device=local)
        x5 = tx_frames_gpu(torch, dev, qpsk_amd, F5, seed=5000, local=local, fs=fs5, rs=rs5, frame_size=L5)
        torch.cuda.synchronize()
        steps5, t5, out5 = 5, [], None
        for r_ in range(steps5 + 1):                      # first call untimed
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out5 = m5.rx_batch_bw(x5, bws5)
            e1.record()
            torch.cuda.synchronize()
            if r_:
                t5.append(e0.elapsed_time(e1))
        m5.sync()
        ms5 = float(np.median(t5))
        hz5 = out5["freq"].double() * rs5 / (2 * np.pi)
        c5 = {"workload": "BASELINE configs[4]: %d frames x %d samples, FS %.0f / RS %.0f, %d loop bandwidths TAU/100..TAU/200 per frame over one filter pass, fixed timing offset %d" % (
                  F5, L5, fs5, rs5, len(bws5), FIXED_INDEX),
              "kernel": m5.last_kernel(), "steps": steps5, "ms_per_step": ms5, "steps_per_s": 1e3 / ms5,
              "msamples_per_s": F5 * L5 / ms5 / 1e3, "loop_msteps_per_s": F5 * (L5 // 8) * len(bws5) / ms5 / 1e3,
              "frac_of_hbm_peak_on_8B_per_sample": BYTES_PER_SAMPLE * F5 * L5 / (ms5 * 1e-3) / 1e9 / HBM_PEAK_GBS,
              "serial_steps_per_chain": L5 // 8, "chains": F5 * len(bws5),
              "hz_min": float(hz5.min().item()), "hz_max": float(hz5.max().item()),
              "chains_within_2hz_of_50": int(((hz5 - 50.0).abs() < 2.0).sum().item())}
        if not args.no_parity:
            from oracle.pyoracle import Oracle, TIMING_FIXED
            npar5 = 5
            want5 = Oracle().rx_batch_bw(x5[:npar5].cpu().numpy(), fs5, rs5, bws5, timing_mode=TIMING_FIXED, fixed_index=FIXED_INDEX)
            c5["parity_frames_checked"] = npar5
            c5["symbol_mismatches"] = int(np.sum(out5["sym"][:npar5].cpu().numpy() != want5["sym"]))
            c5["freq_bit_mismatches"] = int(np.sum(out5["freq"][:npar5].cpu().numpy().view(np.uint32) != want5["freq"].view(np.uint32)))
            c5["phase_bit_mismatches"] = int(np.sum(out5["phase"][:npar5].cpu().numpy().view(np.uint32) != want5["phase"].view(np.uint32)))
        res["config5"] = c5
        m5.close()
        del x5, out5
        torch.cuda.empty_cache()
    if world == 1 and (args.frames, L) == (FRAMES_1GPU, 16384) and not args.no_streams:
        # SURVEY 8(f) N1, the reference's real input and call pattern (qpsk.c:88, 344-354), measured beside the headline and NOT part of it:
        #  (a) 4096 running streams, one 16384-sample int16 PCM block each per call (PCM from the library's own transmit chain at +50 Hz):
        #      mix + rrc_fir() + histogram timing in one kernel, then the loop kernel; ms per block = median of blocks 2..7, an event pair
        #      and a synchronisation around every call;
        #  (b) ONE stream, one 512-sample block per call through host buffers (the shipped FS 9600 / RS 2400 / FRAME_SIZE 512): the
        #      drop-in rx_frame()'s path, wall time per call.
        import ctypes as C_
        S_ = 4096
        ms_ = qpsk_amd.Modem(fs=FS, rs=RS, frame_size=L, timing_mode=qpsk_amd.TIMING_HIST, device=local)
        ms_.streams_reset(S_, 1500.0)
        mtx = qpsk_amd.Modem(fs=FS, rs=RS, frame_size=L, device=local)
        mtx.tx_reset(S_, 1550.0)
        gen = torch.Generator(device=dev)
        gen.manual_seed(77)
        nblk, tms, pcm_blocks = 8, [], []
        o_last = None
        for k in range(nblk):
            symk = torch.randint(0, 4, (S_, L // CYCLES), generator=gen, device=dev, dtype=torch.uint8)
            pcmk = mtx.tx_symbols(symk, want_pcm=True)["pcm"]
            mtx.sync()
            if k < 3:
                pcm_blocks.append(pcmk[0].cpu().numpy())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            o_ = ms_.streams_rx_pcm(pcmk, want_costas=False)
            e1.record()
            torch.cuda.synchronize()
            tms.append(e0.elapsed_time(e1))
            if k == 2:
                o_last = {kk: vv[0].cpu().numpy() for kk, vv in o_.items() if vv is not None}
        ms_.sync()
        st = {"streams": S_, "block_samples": L, "timing": "histogram (qpsk.c:127-180)", "kernels": ms_.last_kernel(),
              "carrier": "one carrier for all streams (what qpsk_streams_reset() sets up): the block's phases from a table run a block ahead by spare waves",
              "pcm_block_ms": float(np.median(tms[2:])), "pcm_msamples_per_s": S_ * L / (float(np.median(tms[2:])) * 1e-3) / 1e6,
              "first_blocks_ms": [float(t) for t in tms[:2]]}
        if not args.no_parity:      # stream 0, blocks 0..2, against the oracle's modem (state carried)
            from oracle.pyoracle import Oracle, TIMING_HIST
            om = Oracle().modem(FS, RS, L, timing_mode=TIMING_HIST)
            om.set_mixer_hz(1500.0)
            for blk in pcm_blocks:
                om.rx_pcm(blk)
            st["stream0_block2_symbol_mismatches"] = int(np.sum(o_last["sym"] != om.symbols))
            st["stream0_block2_index_ok"] = bool(int(o_last["index"]) == int(om.index))
            st["stream0_block2_loop_bits_ok"] = bool(np.float32(o_last["phase"]) == np.float32(om.phase) and np.float32(o_last["freq"]) == np.float32(om.freq))
        ms_.close(); mtx.close()
        # the same block size at the reference's SHIPPED rates (FS 9600 / RS 2400, CYCLES 4: qpsk.h:16-23): twice the symbols per block
        ms4 = qpsk_amd.Modem(fs=9600.0, rs=2400.0, frame_size=L, timing_mode=qpsk_amd.TIMING_HIST, device=local)
        ms4.streams_reset(S_, 1500.0)
        pcm4 = (6000 * torch.randn((S_, L), generator=gen, device=dev)).to(torch.int16)
        t4 = []
        for k in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ms4.streams_rx_pcm(pcm4, want_costas=False)
            e1.record()
            torch.cuda.synchronize()
            t4.append(e0.elapsed_time(e1))
        ms4.sync()
        st["shipped_rates_pcm_block_ms"] = float(np.median(t4[2:]))
        st["shipped_rates_kernels"] = ms4.last_kernel()
        ms4.close()
        m1 = qpsk_amd.Modem(fs=9600.0, rs=2400.0, frame_size=512, device=local)
        m1.streams_reset(1, 1500.0)
        rng = np.random.default_rng(1)
        nb = 2000
        pc = (3000 * rng.standard_normal((nb, 512))).astype(np.int16)
        lst = np.zeros((1, 2), np.float32); sy = np.zeros((1, m1.nsym), np.uint8); cs = np.zeros((1, m1.nsym, 2), np.float32); ix = np.zeros(1, np.int32)
        ptrs = [C_.c_void_p(pc[k].ctypes.data) for k in range(nb)]
        a_ = (C_.c_void_p(lst.ctypes.data), C_.c_void_p(sy.ctypes.data), C_.c_void_p(cs.ctypes.data), C_.c_void_p(ix.ctypes.data))
        for k in range(100):
            m1.L.qpsk_streams_rx_pcm_host(m1.h, ptrs[k], *a_)
        t0 = time.perf_counter()
        for k in range(nb):
            m1.L.qpsk_streams_rx_pcm_host(m1.h, ptrs[k], *a_)
        st["rx_frame_block_us"] = (time.perf_counter() - t0) / nb * 1e6
        st["rx_frame_msamples_per_s"] = 512.0 / st["rx_frame_block_us"]
        st["rx_frame_kernel"] = m1.last_kernel()
        m1.close()
        res["streams"] = st
    print(json.dumps(res), file=real_stdout, flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
