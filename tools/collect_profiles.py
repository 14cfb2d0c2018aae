#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of a measurement session from gpurun_out/ (scratch) into profiles/ (tracked) and
rebuild profiles/traffic.json from the PMC passes.

    python tools/collect_profiles.py r01b

expects (all optional):
    gpurun_out/prof_bench/**/_kernel_stats.csv   rocprofv3 --kernel-trace --stats -- python3 bench.py ...
    gpurun_out/prof_8192, prof_fft, prof_hist2, prof_streams3   the same for tools/sweep.py / tools/bench_streams.py
    gpurun_out/pmc_fetch, pmc_write              rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes)
PMC handling follows /opt/skills/guides/MI355X_MICROARCH.md: counters in their own passes, FETCH_SIZE and
WRITE_SIZE are in KiB, and on gfx950 FETCH_SIZE counts a 128-byte request of a 16-byte-per-lane streaming read as
64 bytes, hence the factor 2 on reads.
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "profiles")
SRC = os.path.join(ROOT, "gpurun_out")
HOT = "rx_fused_pipe_kernel"     # config 2 (16-frame workgroups); 8192-frame shards run rx_lean_kernel


def qpsk_rows(path):
    with open(path) as f:
        rd = csv.DictReader(f)
        rows = [r for r in rd if "qpsk" in r.get("Name", r.get("Kernel_Name", ""))]
        return rd.fieldnames, rows


def failed(sub):
    """tools/measure_all.sh leaves <pass>.failed behind when a pass did not complete"""
    if os.path.exists(os.path.join(SRC, sub + ".failed")):
        print("SKIPPED %s: the pass failed (gpurun_out/%s.failed)" % (sub, sub))
        return True
    return False


def copy_stats(tag, sub, name):
    if failed(sub):
        return
    found = sorted(glob.glob(os.path.join(SRC, sub, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    if not found:
        return
    fields, rows = qpsk_rows(found[-1])       # the newest run (gpurun merges every call's output into the same directories)
    with open(os.path.join(OUT, "%s_%s_kernel_stats.csv" % (tag, name)), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=fields)
        w.writeheader()
        w.writerows(rows)
    print("profiles/%s_%s_kernel_stats.csv: %d kernels" % (tag, name, len(rows)))


def pmc(sub, counter, hot=HOT):
    vals, rows_out, fields = [], [], None
    if failed(sub):
        return vals, rows_out, fields
    found = sorted(glob.glob(os.path.join(SRC, sub, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for p in found[-1:]:                      # the newest run only
        with open(p) as f:
            rd = csv.DictReader(f)
            fields = rd.fieldnames
            for r in rd:
                if hot in r["Kernel_Name"] and r["Counter_Name"] == counter:
                    vals.append(float(r["Counter_Value"]))
                    rows_out.append(r)
    return vals, rows_out, fields


def traffic_entry(tag, fsub, wsub, hot, frames, frame_size, csvname):
    """median FETCH_SIZE / WRITE_SIZE of `hot` over the launches of two separate PMC passes -> bytes per launch"""
    fv, frows, fields = pmc(fsub, "FETCH_SIZE", hot)
    wv, wrows, _ = pmc(wsub, "WRITE_SIZE", hot)
    if not (fv and wv):
        return None
    with open(os.path.join(OUT, csvname), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=fields)
        w.writeheader()
        w.writerows(frows + wrows)
    fv.sort(); wv.sort()
    fetch_kb, write_kb = fv[len(fv) // 2], wv[len(wv) // 2]
    read_b, write_b = 2.0 * fetch_kb * 1024.0, write_kb * 1024.0
    print("%s %dx%d: read %.1f MB (%.4f x algorithmic), write %.1f MB" % (hot, frames, frame_size, read_b / 1e6,
                                                                          read_b / (8.0 * frames * frame_size), write_b / 1e6))
    return {
        "frames": frames, "frame_size": frame_size, "kernel": hot,
        "FETCH_SIZE_KB_raw": fetch_kb, "WRITE_SIZE_KB_raw": write_kb,
        "correction": "gfx950: FETCH_SIZE counts 128-B requests as 64 B for 16-B/lane streaming reads -> x2 "
                      "(MI355X_MICROARCH.md, HBM); WRITE_SIZE taken as is",
        "read_bytes": read_b, "write_bytes": write_b, "hbm_bytes_per_launch": read_b + write_b,
        "algorithmic_read_bytes": 8 * frames * frame_size,
        "source": "profiles/%s (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes over bench.py, median of "
                  "%d launches)" % (csvname, len(fv)),
    }


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    os.makedirs(OUT, exist_ok=True)
    for sub, name in (("prof_bench", "bench_config2"), ("prof_8192", "config4_shard_8192"), ("prof_fft", "fft_timing"),
                      ("prof_hist2", "hist_timing"), ("prof_streams3", "streams_tx"), ("prof_bench20", "bench_config2_driver_style")):
        copy_stats(tag, sub, name)
    shapes = {}
    for key, fsub, wsub, hot, frames in (("4096x16384", "pmc_fetch", "pmc_write", "rx_fused_pipe_kernel", 4096),
                                         ("8192x16384", "pmc_fetch_8192", "pmc_write_8192", "rx_lean_kernel", 8192)):
        e = traffic_entry(tag, fsub, wsub, hot, frames, 16384, "%s_pmc_fetch_write_%d.csv" % (tag, frames))
        if e:
            shapes[key] = e
    if shapes:
        tj = dict(shapes.get("4096x16384", {}))       # top level = config 2, as bench.py read it in round 1
        tj["shapes"] = shapes
        with open(os.path.join(OUT, "traffic.json"), "w") as f:
            json.dump(tj, f, indent=1)


if __name__ == "__main__":
    main()
