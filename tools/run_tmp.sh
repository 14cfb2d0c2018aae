cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
sha256sum qpsk_amd/libqpsk_hip.so > $O/library.sha256
pmcrun() { d=$1; c=$2; shift 2; rm -rf $O/$d $O/$d.failed; timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $O/$d -- "$@" > $O/$d.log 2>&1 || echo FAILED > $O/$d.failed; }
prof() { d=$1; shift; rm -rf $O/$d $O/$d.failed; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$d -- "$@" > $O/$d.log 2>&1 || echo FAILED > $O/$d.failed; }
export QPSK_HIST_ONEPASS=1
pmcrun pmc_fetch_hist1 FETCH_SIZE python3 tools/loop_kernel.py hist 0 4096 12
pmcrun pmc_write_hist1 WRITE_SIZE python3 tools/loop_kernel.py hist 0 4096 12
prof prof_hist1 python3 tools/loop_kernel.py hist 2 4096 50
timeout -k 10 300 python3 tools/sweep.py --frames 4096 --timing hist --rounds 12 --per-round 4 "QPSK_LEAN_PAIR=0" "" > $O/hist_pair_sweep.txt 2>&1; tail -3 $O/hist_pair_sweep.txt
unset QPSK_HIST_ONEPASS
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*.db" -delete
ls $O/*.failed 2>/dev/null
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > $O/bench20b.json 2> $O/bench20b.err; python3 -c "
import json;d=json.load(open('$O/bench20b.json'));print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic_source']['traffic_matches_library'], d['gather']['ms_per_step_overlapped_direct'], d['hist']['ms_per_step'])"
