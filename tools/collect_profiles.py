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
HOT = "rx_fused_pipe_kernel"


def qpsk_rows(path):
    with open(path) as f:
        rd = csv.DictReader(f)
        rows = [r for r in rd if "qpsk" in r.get("Name", r.get("Kernel_Name", ""))]
        return rd.fieldnames, rows


def copy_stats(tag, sub, name):
    found = sorted(glob.glob(os.path.join(SRC, sub, "**", "*kernel_stats.csv"), recursive=True))
    if not found:
        return
    fields, rows = qpsk_rows(found[-1])
    with open(os.path.join(OUT, "%s_%s_kernel_stats.csv" % (tag, name)), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=fields)
        w.writeheader()
        w.writerows(rows)
    print("profiles/%s_%s_kernel_stats.csv: %d kernels" % (tag, name, len(rows)))


def pmc(sub, counter):
    vals, rows_out, fields = [], [], None
    for p in sorted(glob.glob(os.path.join(SRC, sub, "**", "*counter_collection.csv"), recursive=True)):
        with open(p) as f:
            rd = csv.DictReader(f)
            fields = rd.fieldnames
            for r in rd:
                if HOT in r["Kernel_Name"] and r["Counter_Name"] == counter:
                    vals.append(float(r["Counter_Value"]))
                    rows_out.append(r)
    return vals, rows_out, fields


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    os.makedirs(OUT, exist_ok=True)
    for sub, name in (("prof_bench", "bench_config2"), ("prof_8192", "config4_shard_8192"), ("prof_fft", "fft_timing"),
                      ("prof_hist2", "hist_timing"), ("prof_streams3", "streams_tx")):
        copy_stats(tag, sub, name)
    fv, frows, fields = pmc("pmc_fetch", "FETCH_SIZE")
    wv, wrows, _ = pmc("pmc_write", "WRITE_SIZE")
    if fv and wv:
        with open(os.path.join(OUT, "%s_pipe_pmc_fetch_write.csv" % tag), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=fields)
            w.writeheader()
            w.writerows(frows + wrows)
        fv.sort(); wv.sort()
        fetch_kb, write_kb = fv[len(fv) // 2], wv[len(wv) // 2]
        frames, frame_size = 4096, 16384
        read_b, write_b = 2.0 * fetch_kb * 1024.0, write_kb * 1024.0
        tj = {
            "frames": frames, "frame_size": frame_size, "kernel": HOT,
            "FETCH_SIZE_KB_raw": fetch_kb, "WRITE_SIZE_KB_raw": write_kb,
            "correction": "gfx950: FETCH_SIZE counts 128-B requests as 64 B for 16-B/lane streaming reads -> x2 "
                          "(MI355X_MICROARCH.md, HBM); WRITE_SIZE taken as is",
            "read_bytes": read_b, "write_bytes": write_b, "hbm_bytes_per_launch": read_b + write_b,
            "algorithmic_read_bytes": 8 * frames * frame_size,
            "source": "profiles/%s_pipe_pmc_fetch_write.csv (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate "
                      "passes over bench.py, median of %d launches)" % (tag, len(fv)),
        }
        with open(os.path.join(OUT, "traffic.json"), "w") as f:
            json.dump(tj, f, indent=1)
        print("traffic.json: read %.1f MB (%.4f x algorithmic), write %.1f MB" % (
            read_b / 1e6, read_b / tj["algorithmic_read_bytes"], write_b / 1e6))


if __name__ == "__main__":
    main()
