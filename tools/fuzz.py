#!/usr/bin/env python3
"""Differential fuzzer (GPU box): tests/fuzzcases.py for --cases seeds from --seed on, one line per case, a summary at the end.

    python tools/fuzz.py --kind batch --seed 1 --cases 400 > gpurun_out/fuzz/batch.log
    python tools/fuzz.py --kind streams --seed 1 --cases 150
    python tools/fuzz.py --kind stages --seed 1 --cases 150

Exit status 1 if any case differs from the oracle.  The oracle is the checker (test infrastructure); nothing here is a product path."""
import argparse
import collections
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzzcases  # noqa: E402
import qpsk_amd  # noqa: E402
from oracle.pyoracle import Oracle  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--kind", choices=["batch", "streams", "stages"], default="batch")
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--cases", type=int, default=100)
ap.add_argument("--max-samples", type=int, default=6_000_000)
args = ap.parse_args()
orc = Oracle()
fails, skipped = 0, 0
kernels = collections.Counter()
t0 = time.time()
for seed in range(args.seed, args.seed + args.cases):
    try:
        if args.kind == "batch":
            desc, bad = fuzzcases.batch_case(orc, qpsk_amd.Modem, seed, args.max_samples)
        elif args.kind == "streams":
            desc, bad = fuzzcases.streams_case(orc, qpsk_amd.Modem, seed)
        else:
            desc, bad = fuzzcases.stages_case(orc, qpsk_amd.Modem, seed)
    except qpsk_amd.QpskError as e:
        desc, bad = "seed %d: library error: %s" % (seed, e), ["error"]
    if bad is None:
        skipped += 1
        print(desc, flush=True)
        continue
    kernels[desc.split(" kernel ")[-1] if " kernel " in desc else "?"] += 1
    fails += bool(bad)
    print("%s -> %s" % (desc, "ok" if not bad else "DIFFERS: %s" % bad), flush=True)
print("%s: %d cases from seed %d, %d skipped, %d differ, %.0f s; kernels that served them: %s" % (
    args.kind, args.cases, args.seed, skipped, fails, time.time() - t0, dict(kernels)))
sys.exit(1 if fails else 0)
