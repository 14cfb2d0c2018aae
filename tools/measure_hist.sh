# histogram timing mode (QPSK_TIMING_HIST qpsk_rx_batch at config 2): kernel times, HBM counters, SQ counters of
# timing_scan_kernel.  GPU box, one gpurun call; summaries: python tools/collect_hist.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
CMD="python3 tools/sweep.py --frames 4096 --timing hist --rounds 1 --per-round 2"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/hist_prof -- python3 tools/sweep.py --timing hist > $O/hist_prof.log 2>&1 || exit 1
pmcrun() { d=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $O/$d -- $CMD > $O/$d.log 2>&1; }
pmcrun hist_pmc_fetch FETCH_SIZE || exit 1
pmcrun hist_pmc_write WRITE_SIZE || exit 1
pmcrun hist_sq_a SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY || exit 1
pmcrun hist_sq_b SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU || exit 1
pmcrun hist_sq_c SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS || exit 1
python3 tools/kstats.py "$O/hist_prof/**/*kernel_stats.csv"
