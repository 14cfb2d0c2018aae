#!/usr/bin/env python3
"""Print the qpsk:: rows of a rocprofv3 *_kernel_stats.csv (name, calls, average / min / max in microseconds)."""
import csv
import glob
import sys

pat = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/**/*kernel_stats.csv"
for p in sorted(glob.glob(pat, recursive=True)):
    print(p)
    for r in csv.DictReader(open(p)):
        if "qpsk" in r["Name"]:
            print("  %-28s calls %4s  avg %9.1f us  min %9.1f  max %9.1f" % (
                r["Name"].split("(")[0][:28], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3,
                float(r["MaxNs"]) / 1e3))
