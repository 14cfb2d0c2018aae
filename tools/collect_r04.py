#!/usr/bin/env python3
"""Copy the outputs of tools/measure_r04.sh (or measure_r05.sh) from gpurun_out/<tag>/ (scratch) into profiles/<tag>_* (tracked) and
rebuild profiles/traffic.json from the HBM counter passes.  No narrative is added here: every file says which command produced it.

    python tools/collect_r04.py [tag]        (default r04; `r05` reads gpurun_out/r05 and writes profiles/r05_*)

traffic.json carries the sha256 of the library the counters were taken on (gpurun_out/<tag>/library.sha256, written by the
measurement script): bench.py compares it with the library it runs and says `traffic_matches_library`.

PMC handling follows /opt/skills/guides/MI355X_MICROARCH.md: counters in their own passes; FETCH_SIZE and WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE counts a 128-byte request of a 16-byte-per-lane streaming read as 64 bytes, hence the factor 2 on reads.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r04"
SRC = os.path.join(ROOT, "gpurun_out", TAG)
OUT = os.path.join(ROOT, "profiles")


def ok(name):
    if os.path.exists(os.path.join(SRC, name + ".failed")):
        print("SKIPPED %s: the pass failed" % name)
        return False
    return True


def newest(sub, pattern):
    found = sorted(glob.glob(os.path.join(SRC, sub, "**", pattern), recursive=True), key=os.path.getmtime)
    return found[-1] if found else None


def copy_stats(sub, name):
    if not ok(sub):
        return
    p = newest(sub, "*kernel_stats.csv")
    if not p:
        return
    with open(p) as f:
        rd = csv.DictReader(f)
        rows = [r for r in rd if "qpsk" in r.get("Name", "")]
        fields = rd.fieldnames
    with open(os.path.join(OUT, "%s_%s_kernel_stats.csv" % (TAG, name)), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=fields)
        w.writeheader()
        w.writerows(rows)
    print("profiles/%s_%s_kernel_stats.csv: %d kernels" % (TAG, name, len(rows)))


def counters(sub):
    """{kernel: {counter: [value per dispatch]}} of one pass (rows of a dispatch are summed: one row per XCD / dimension)"""
    out = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    if not ok(sub):
        return {}
    p = newest(sub, "*counter_collection.csv")
    if not p:
        return {}
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "qpsk" in k:
            out[k][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: {c: sorted(v.values()) for c, v in cs.items()} for k, cs in out.items()}


def median(v):
    return v[len(v) // 2]


def main():
    os.makedirs(OUT, exist_ok=True)
    for n in ("bench", "bench20", "bench8192", "bench_gpus2_shared", "bench_gpus6_shared", "bench_torchrun2_shared", "bench_torchrun4_shared",
              "bench_gpus2_shared_20steps", "bench_gpus6_shared_20steps"):
        p = os.path.join(SRC, n + ".json")
        if ok(n) and os.path.exists(p) and os.path.getsize(p) > 0:
            shutil.copy(p, os.path.join(OUT, "%s_%s.json" % (TAG, n)))
    for sub, name in (("prof_bench", "bench"), ("prof_bench20", "bench_config2_driver_style"), ("prof_config3", "config3"),
                      ("prof_fft_sep", "fft_estimator_kernel"), ("prof_streams", "streams"), ("prof_fir", "fir"), ("prof_hist1", "hist_one_pass")):
        copy_stats(sub, name)

    # ---- HBM traffic per launch and kernel, per workload
    lines = ["HBM traffic per launch (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes over `python3 tools/loop_kernel.py <workload> 0 <frames> 12`,",
             "median over the launches of a pass; reads = FETCH_SIZE x 2 x 1024 B (gfx950 counts a 128-byte request of a 16-byte-per-lane streaming read as 64 B,",
             "MI355X_MICROARCH.md), writes = WRITE_SIZE x 1024 B).  Algorithmic bytes of a 4096 x 16384 batch: 536,870,912 (8 B per input sample).", ""]
    shapes = {}
    for w, frames in (("config2", 4096), ("8192", 8192), ("config3", 4096), ("hist", 4096), ("hist1", 4096), ("fft_est", 4096), ("scan", 4096), ("fir", 4096), ("streams", 4096)):
        fc, wc = counters("pmc_fetch_" + w), counters("pmc_write_" + w)
        for k in sorted(set(fc) | set(wc)):
            fv, wv = fc.get(k, {}).get("FETCH_SIZE"), wc.get(k, {}).get("WRITE_SIZE")
            if not fv or not wv:
                continue
            rb, wb = 2.0 * median(fv) * 1024.0, median(wv) * 1024.0
            alg = 8.0 * frames * 16384
            lines.append("%-8s %-46s reads %8.1f MB (%.4f x the batch's bytes)  writes %7.1f MB   (%d / %d launches)" % (
                w, k.replace("qpsk::", ""), rb / 1e6, rb / alg, wb / 1e6, len(fv), len(wv)))
            if w in ("config2", "8192") and ("rx_fused_pipe_kernel" in k or "rx_lean_kernel" in k):
                shapes["%dx16384" % frames] = {
                    "frames": frames, "frame_size": 16384, "kernel": k.replace("qpsk::", "").split("<")[0],
                    "FETCH_SIZE_KB_raw": median(fv), "WRITE_SIZE_KB_raw": median(wv),
                    "correction": "gfx950: FETCH_SIZE counts 128-B requests as 64 B for 16-B/lane streaming reads -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE taken as is",
                    "read_bytes": rb, "write_bytes": wb, "hbm_bytes_per_launch": rb + wb, "algorithmic_read_bytes": int(alg),
                    "source": "profiles/%s_hbm_traffic.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes over tools/loop_kernel.py, median of %d launches)" % (TAG, len(fv))}
    if any(ln.startswith("hist1") for ln in lines):
        lines += ["", "hist  = histogram mode as the counter pass runs it: its 12 calls are enqueued before the first one's statistics have reached the host, so every",
                  "        call takes the two-launch route (timing_scan_kernel, then the receive kernel: the batch read twice);",
                  "hist1 = the same pass with QPSK_HIST_ONEPASS=1: from the second call on rx_hist_kernel reads the batch ONCE (round 6), the fall-back pass",
                  "        (rx_fused_kernel over the list of frames the guess missed) and the majority kernel read next to nothing."]
    if len(lines) > 4:
        open(os.path.join(OUT, "%s_hbm_traffic.txt" % TAG), "w").write("\n".join(lines) + "\n")
        print("\n".join(lines[4:]))
    if shapes:
        sha = None
        try:
            sha = open(os.path.join(SRC, "library.sha256")).read().split()[0]
        except OSError:
            pass
        for v in shapes.values():
            v["library_sha256"] = sha
        tj = dict(shapes.get("4096x16384", {}))
        tj["shapes"] = shapes
        json.dump(tj, open(os.path.join(OUT, "traffic.json"), "w"), indent=1)

    # ---- SQ counters of the two full-rate FIR kernels
    out = ["SQ counters per launch (rocprofv3 --pmc, three passes each over `python3 tools/loop_kernel.py scan|fir 0 4096 8`; summed over the chip; SQ_* cycle counters in quad-cycles)", ""]
    for w in ("scan", "fir"):
        agg = collections.defaultdict(dict)
        for part in "abc":
            for k, cs in counters("sq_%s_%s" % (w, part)).items():
                for c, v in cs.items():
                    agg[k][c] = (sum(v) / len(v), len(v))
        for k in sorted(agg):
            if "timing_scan" in k or "rrc_fir_stream" in k:
                out.append(k.replace("qpsk::", ""))
                for c in sorted(agg[k]):
                    out.append("  %-24s %14.4g per launch (%d launches)" % (c, agg[k][c][0], agg[k][c][1]))
    if len(out) > 2:
        open(os.path.join(OUT, "%s_sq_counters_fir_kernels.txt" % TAG), "w").write("\n".join(out) + "\n")

    # ---- text logs
    def text(name, dst, header):
        p = os.path.join(SRC, name + ".log")
        if ok(name) and os.path.exists(p):
            body = [ln for ln in open(p).read().splitlines() if "amdgpu.ids" not in ln]
            open(os.path.join(OUT, "%s_%s.txt" % (TAG, dst)), "w").write(header + "\n\n" + "\n".join(body) + "\n")
    text("power", "power", "Board power, shader clock, per-XCD clocks and PPT throttle activity while one library call runs back to back for 6 s\n"
         "(tools/power_probe.py --cmd \"python3 tools/loop_kernel.py <workload> 6 [frames]\"; steady-state ms per call printed by the child).")
    text("power_streams", "power_streams", "4096 running PCM streams, one 16384-sample block per call (qpsk_streams_rx_pcm: stream_scan_kernel + costas_pipe_kernel), back to back for 6 s\n"
         "(tools/power_probe.py --cmd \"python3 tools/loop_kernel.py streams 6\").")
    text("config3", "config3", "BASELINE config 3 against config 2 in one process (tools/bench_config3.py --hist): K back-to-back qpsk_rx_batch calls between two events.")
    text("fir_fast", "fir", "qpsk_rrc_fir_batch on the config-2 block, 4096 x 16384 samples (tools/bench_fir_fast.py): per-launch event times (clocks not settled: see r04_power.txt for steady state).")
    text("dropin", "dropin_rx_frame", "The drop-in rx_frame() (examples/dropin_main.c through libqpsk_hip) against the reference's rx_frame() compiled here (tools/bench_dropin.py 3000).")
    text("config5", "config5", "BASELINE config 5 (tools/bench_config5.py).")
    text("zero_stretches", "zero_stretches", "Stretches of exactly-zero symbols (costas_asm.h `ign`): tools/bench_streams_zero.py.  Before that change every group of a lane with such\n"
         "symbols fell out of the Costas instruction stream into the C++ step, for its whole workgroup: the loop kernel of the first block after\n"
         "qpsk_streams_reset() took 1.06 ms instead of 0.16-0.18 (round 3's and this round's first traces).")
    blocks = []
    for name, label in (("streams", "4096 streams, the library's choice (filter + scan as one kernel, the streams' one carrier from the table: carrier.h)"),
                        ("streams_own_carrier", "4096 streams, QPSK_STREAM_CARRIER=0 (every stream's own carrier recurrence: the kernel's mixer wave)"),
                        ("streams_apart", "4096 streams, QPSK_STREAM_SCAN=0 (mixer, filter, scan kernels apart)"),
                        ("streams_1024", "1024 streams, the library's choice"), ("streams_1024_apart", "1024 streams, QPSK_STREAM_SCAN=0"),
                        ("streams_2560", "2560 streams, the library's choice (before the carrier table)"), ("streams_2560_apart", "2560 streams, QPSK_STREAM_SCAN=0 (before the carrier table)")):
        p = os.path.join(SRC, name + ".log")
        if ok(name) and os.path.exists(p):
            blocks += [label + ":"] + ["  " + ln for ln in open(p).read().splitlines() if ln.startswith("streams")]
    if blocks:
        open(os.path.join(OUT, "%s_streams_blocks.txt" % TAG), "w").write(
            "Streaming mode, 16384-sample blocks, histogram timing: ms per block of all streams, one call per block with an event pair and a synchronisation around it\n"
            "(tools/bench_streams.py; median of blocks 2-6 after qpsk_streams_reset):\n\n" + "\n".join(blocks) + "\n")
    hosts = []
    for n in (1, 8, 64):
        p = os.path.join(SRC, "streams_host_%d.log" % n)
        if ok("streams_host_%d" % n) and os.path.exists(p):
            hosts += [ln for ln in open(p).read().splitlines() if "samples:" in ln]
    if hosts:
        open(os.path.join(OUT, "%s_streams_host.txt" % TAG), "w").write(
            "qpsk_streams_rx_pcm_host, one 512-sample block per stream and call, host buffers (tools/bench_streams_host.py <streams> <blocks>, shipped configuration FS 9600 / RS 2400):\n\n" + "\n".join(hosts) + "\n")
    p = os.path.join(SRC, "stream_block_profile.log")
    if ok("stream_block_profile") and os.path.exists(p):
        body = [ln for ln in open(p).read().splitlines() if ln.startswith("wave")]
        open(os.path.join(OUT, "%s_stream_block.txt" % TAG), "w").write(
            "stream_block_kernel, one stream, 512-sample blocks: shader cycles per phase of its waves (s_memtime stamps, build -DQPSK_SBLK_PROF; the printf adds to the\n"
            "wall time, the cycle counts are the kernel's), tools/bench_streams_host.py 1 60:\n\n" + "\n".join(body[-24:]) + "\n")
    # streams per call, from the kernel trace
    p = newest("prof_streams", "*kernel_trace.csv") if ok("prof_streams") else None
    if p:
        rows = [r for r in csv.DictReader(open(p)) if "qpsk" in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        t0 = int(rows[0]["Start_Timestamp"])
        out = ["Streaming mode, 4096 streams x 16384-sample blocks, histogram timing (tools/bench_streams.py under rocprofv3 --kernel-trace): every launch of the large",
               "kernels in order, complex input first (6 blocks), then PCM input (6 blocks).", ""]
        for r in rows:
            n = r["Kernel_Name"].split("(")[0].replace("qpsk::", "").replace("void ", "")
            if n.split("<")[0] in ("costas_pipe_kernel", "rrc_fir_kernel", "rrc_fir_stream_kernel", "mixer_kernel", "timing_hist8_kernel", "stream_block_kernel", "stream_scan_kernel"):
                out.append("%-26s start %9.1f us   %8.1f us" % (n, (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
        open(os.path.join(OUT, "%s_streams_per_call.txt" % TAG), "w").write("\n".join(out) + "\n")
    print(len(glob.glob(os.path.join(OUT, TAG + "_*"))), "files under profiles/%s_*" % TAG)


if __name__ == "__main__":
    main()
