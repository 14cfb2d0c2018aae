#!/bin/bash
# Round-6 measurement session behind profiles/r06_* (GPU box): bash tools/measure_r06.sh [parts]
#   a: bench lines, rocprofv3 kernel stats of the default and the driver-style command, HBM counters (with the library's hash)
#   b: power / clocks, N > 1 rehearsals on this one GPU at the driver's 20 steps (the clock stops in front of the trailing barrier), config 3, drop-in, streams
# Every pass clears its output first and leaves <name>.failed behind when it did not complete; python tools/collect_r04.py r06 copies into profiles/.  (Round 6: the step micro-benchmarks are tools/measure_r06_step.sh.)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O; part=${1:-ab}
run() { name=$1; shift; rm -f $O/$name.failed; timeout -k 10 500 "$@" > $O/$name.json 2> $O/$name.err || echo FAILED > $O/$name.failed; }
txt() { name=$1; shift; rm -f $O/$name.failed; timeout -k 10 400 "$@" > $O/$name.log 2>&1 || echo FAILED > $O/$name.failed; }
prof() { d=$1; shift; rm -rf $O/$d $O/$d.failed; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$d -- "$@" > $O/$d.log 2>&1 || echo FAILED > $O/$d.failed; }
pmcrun() { d=$1; c=$2; shift 2; rm -rf $O/$d $O/$d.failed; timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $O/$d -- "$@" > $O/$d.log 2>&1 || echo FAILED > $O/$d.failed; }
sha256sum qpsk_amd/libqpsk_hip.so > $O/library.sha256
if [[ $part == *a* ]]; then
run bench python3 bench.py
run bench20 python3 bench.py --steps 20 --warmup 5
run bench8192 python3 bench.py --frames 8192 --cpu-frames 0 --no-shard
prof prof_bench python3 bench.py --cpu-frames 0
prof prof_bench20 python3 bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-shard --no-timing-modes --no-config5 --no-streams
for w in config2 config3 hist streams; do
  pmcrun pmc_fetch_$w FETCH_SIZE python3 tools/loop_kernel.py $w 0 4096 12
  pmcrun pmc_write_$w WRITE_SIZE python3 tools/loop_kernel.py $w 0 4096 12
done
# histogram mode on the one-pass route whatever the batch's index mix (tools/loop_kernel.py's torch stimulus puts a few frames off the majority index)
export QPSK_HIST_ONEPASS=1
pmcrun pmc_fetch_hist1 FETCH_SIZE python3 tools/loop_kernel.py hist 0 4096 12
pmcrun pmc_write_hist1 WRITE_SIZE python3 tools/loop_kernel.py hist 0 4096 12
prof prof_hist1 python3 tools/loop_kernel.py hist 2 4096 50
unset QPSK_HIST_ONEPASS
pmcrun pmc_fetch_8192 FETCH_SIZE python3 tools/loop_kernel.py config2 0 8192 12
pmcrun pmc_write_8192 WRITE_SIZE python3 tools/loop_kernel.py config2 0 8192 12
fi
if [[ $part == *b* ]]; then
txt power python3 tools/power_probe.py --cmd "python3 tools/loop_kernel.py config2 6" "python3 tools/loop_kernel.py config2 6 8192" "python3 tools/loop_kernel.py config3 6" "python3 tools/loop_kernel.py hist 6" "python3 tools/loop_kernel.py streams 6"
run bench_gpus2_shared_20steps python3 bench.py --gpus 2 --steps 20 --warmup 5 --cpu-frames 0
run bench_gpus6_shared_20steps python3 bench.py --gpus 6 --steps 20 --warmup 5 --frames 1024 --cpu-frames 0
run bench_torchrun2_shared python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 --cpu-frames 0
txt config3 python3 tools/bench_config3.py --hist
txt fuzz python3 tools/fuzz.py --cases 1200 --seed 6000
txt fuzz_streams python3 tools/fuzz.py --kind streams --cases 300 --seed 6100
txt fuzz_stages python3 tools/fuzz.py --kind stages --cases 300 --seed 6200
txt dropin python3 tools/bench_dropin.py 3000
txt streams python3 tools/bench_streams.py
fi
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*.db" -delete; du -sh $O | tail -1
ls $O/*.failed 2>/dev/null; cut -c1-300 $O/bench.json 2>/dev/null
