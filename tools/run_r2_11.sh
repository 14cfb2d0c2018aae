cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
rocprofv3 -L > $O/r2_counters_list.txt 2>&1
grep -o "SQ_[A-Z_0-9]*" $O/r2_counters_list.txt | sort -u | tr '\n' ' ' | head -c 6000; echo
export QPSK_PIPE_V=2 QPSK_PIPE_LAYOUT_LO=139810 QPSK_PIPE_LAYOUT_HI=69666 QPSK_PIPE_DBG=1536
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/r2_pmc_sq1 -- python3 bench.py --frames 8192 --steps 5 --warmup 1 --cpu-frames 0 --no-parity > $O/r2_pmc_sq1.log 2>&1
tail -2 $O/r2_pmc_sq1.log
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_SALU --output-format csv -d $O/r2_pmc_sq2 -- python3 bench.py --frames 8192 --steps 5 --warmup 1 --cpu-frames 0 --no-parity > $O/r2_pmc_sq2.log 2>&1
tail -2 $O/r2_pmc_sq2.log
find $O/r2_pmc_sq1 $O/r2_pmc_sq2 -name "*counter_collection.csv" | head
python3 - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/r2_pmc_sq1", "gpurun_out/r2_pmc_sq2"):
    acc = collections.defaultdict(list)
    for p in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if "rx_pipe2" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print("%-24s n=%d mean %.4g" % (k, len(v), sum(v) / len(v)))
PY
