#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter rows per kernel over the gpurun_out/<prefix>* passes (default hist_: tools/measure_hist.sh;
sq_4096 / sq_8192: tools/measure_all.sh) and print
per-launch values (counter values of a kernel are summed over its dispatches' rows and divided by its dispatches)."""
import csv
import glob
import collections
import sys

tot = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(lambda: collections.defaultdict(set))
import os
paths = {}
for p in glob.glob((sys.argv[1] if len(sys.argv) > 1 else "gpurun_out") + "/" + (sys.argv[2] if len(sys.argv) > 2 else "hist_") + "*/**/*counter_collection.csv", recursive=True):
    d = p.split(os.sep)[1]          # one pass per directory: the newest file of each (gpurun merges runs into the same tree)
    if d not in paths or os.path.getmtime(p) > os.path.getmtime(paths[d]):
        paths[d] = p
for p in paths.values():
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0]
        if "qpsk" not in k:
            continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k][r["Counter_Name"]].add(r["Dispatch_Id"])
for k in sorted(tot):
    print(k)
    for c in sorted(tot[k]):
        n = len(disp[k][c])
        print("  %-24s %14.4g per launch (%d launches)" % (c, tot[k][c] / n, n))
