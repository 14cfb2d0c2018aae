#!/usr/bin/env python3
"""Turns gpurun_out/ledger.txt (tools/ledger.sh) into the energy ledger of profiles/r04_energy_ledger.txt: energy per launch =
board power (amd-smi / rocm-smi, steady state) x steady-state time per launch, for the full kernel and for each variant with one
part left out; a part's energy = full - variant.

    python tools/ledger_table.py gpurun_out/ledger.txt
"""
import re
import sys

txt = open(sys.argv[1]).read()
rows = {}
power = {}
for m in re.finditer(r"(\d+) frames, dbg (\d+), t\+2s:.*?Power \(W\): ([0-9.]+) \| sclk clock level: \S+ \((\d+)Mhz\)(?: \| amd-smi: (\d+) W, XCD clocks (\d+)-(\d+) MHz \(mean (\d+)\), PPT violation (\S+(?: \S+)?) \((\d+) %\), UMC activity (\d+) %)?", txt):
    key = (int(m.group(1)), int(m.group(2)))
    power.setdefault(key, []).append(m.groups())
for m in re.finditer(r"child: (\d+) frames, dbg (\d+) .*?, (\S+), ([0-9.]+) ms per launch", txt):
    key = (int(m.group(1)), int(m.group(2)))
    rows.setdefault(key, []).append((m.group(3), float(m.group(4))))
idle = float(re.search(r"idle: .*?Power \(W\): ([0-9.]+)", txt).group(1))
NAMES = {0: "the kernel as it ships", 1: "without the filter's multiplies and adds", 16384: "without the filter's window reads",
         32768: "without the flush's arithmetic", 65536: "without the window staging writes", 2: "without the Costas recurrence",
         3: "without filter arithmetic and recurrence", 16386: "without window reads and recurrence", 32770: "without flush arithmetic and recurrence",
         65538: "without staging writes and recurrence"}
print("idle board power %.0f W" % idle)
for frames in (8192, 4096):
    if (frames, 0) not in rows:
        continue
    print("\n%d frames x 16384 samples (%s): energy per launch = power x time, first measurement-build run of each variant" % (frames, rows[(frames, 0)][0][0]))
    print("%-46s %8s %7s %9s %9s %8s %6s %9s" % ("variant", "ms", "W", "mean MHz", "PPT act.", "J", "% full", "part J"))
    full = None
    for dbg in (0, 1, 16384, 32768, 65536, 2, 3, 16386, 32770, 65538):
        key = (frames, dbg)
        if key not in rows:
            continue
        ms = rows[key][0][1]
        g = power[key][0]
        w = float(g[4] or g[2])
        mhz = int(g[7] or g[3])
        e = w * ms * 1e-3
        if dbg == 0:
            full = e
        print("%-46s %8.4f %7.0f %9d %8s%% %8.4f %6.1f %9s" % (NAMES[dbg], ms, w, mhz, g[9] or "?", e, 100 * e / full,
                                                             "" if dbg == 0 else "%.4f" % (full - e)))
    bytes_ = 8.0 * frames * 16384
    print("algorithmic bytes %.0f: the kernel as it ships moves %.2f TB/s = %.1f %% of 8 TB/s at %.3f nJ per input byte" % (
        bytes_, bytes_ / (rows[(frames, 0)][0][1] * 1e-3) / 1e12, bytes_ / (rows[(frames, 0)][0][1] * 1e-3) / 8e12 * 100, full / bytes_ * 1e9))
