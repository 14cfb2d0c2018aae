#!/bin/bash
# Round-4 measurement session behind profiles/r04_* (GPU box, one or two gpurun calls): bash tools/measure_r04.sh [part]
#   part a: bench lines, rocprofv3 kernel stats, HBM counters;  part b: power / clocks, streams, drop-in, rehearsals;  part c: torchrun rehearsal, stream-block profile
# Every pass clears its output first and leaves <name>.failed behind when it did not complete; python tools/collect_r04.py copies into profiles/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O; part=${1:-abc}
run() { name=$1; shift; rm -f $O/$name.failed; "$@" > $O/$name.json 2> $O/$name.err || echo FAILED > $O/$name.failed; }
txt() { name=$1; shift; rm -f $O/$name.failed; timeout -k 10 400 "$@" > $O/$name.log 2>&1 || echo FAILED > $O/$name.failed; }
prof() { d=$1; shift; rm -rf $O/$d $O/$d.failed; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$d -- "$@" > $O/$d.log 2>&1 || echo FAILED > $O/$d.failed; }
pmcrun() { d=$1; c=$2; shift 2; rm -rf $O/$d $O/$d.failed; timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $O/$d -- "$@" > $O/$d.log 2>&1 || echo FAILED > $O/$d.failed; }
if [[ $part == *a* ]]; then
run bench python3 bench.py
run bench20 python3 bench.py --steps 20 --warmup 5
run bench8192 python3 bench.py --frames 8192 --cpu-frames 0 --no-shard
prof prof_bench python3 bench.py --cpu-frames 0                    # config 2, config 3, histogram mode, the 8192-frame shard: the default command, CPU leg off
prof prof_bench20 python3 bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-shard --no-timing-modes   # the driver's command
prof prof_config3 python3 tools/loop_kernel.py config3 2
prof prof_fft_sep python3 tools/loop_kernel.py fft_est 2
prof prof_streams python3 tools/bench_streams.py
prof prof_fir python3 tools/loop_kernel.py fir 2
# HBM counters: separate passes, few launches each
for w in config2 config3 hist fft_est scan fir; do
  pmcrun pmc_fetch_$w FETCH_SIZE python3 tools/loop_kernel.py $w 0 4096 12
  pmcrun pmc_write_$w WRITE_SIZE python3 tools/loop_kernel.py $w 0 4096 12
done
pmcrun pmc_fetch_8192 FETCH_SIZE python3 tools/loop_kernel.py config2 0 8192 12
pmcrun pmc_write_8192 WRITE_SIZE python3 tools/loop_kernel.py config2 0 8192 12
for k in a b c; do
  case $k in a) c="SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY";; b) c="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU";; c) c="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS";; esac
  pmcrun sq_scan_$k "$c" python3 tools/loop_kernel.py scan 0 4096 8
  pmcrun sq_fir_$k "$c" python3 tools/loop_kernel.py fir 0 4096 8
done
fi
if [[ $part == *b* ]]; then
txt power python3 tools/power_probe.py --cmd "python3 tools/loop_kernel.py config2 6" "python3 tools/loop_kernel.py config2 6 8192" "python3 tools/loop_kernel.py config3 6" "python3 tools/loop_kernel.py hist 6" "python3 tools/loop_kernel.py scan 6" "python3 tools/loop_kernel.py fft_est 6" "python3 tools/loop_kernel.py fir 6" "python3 tools/loop_kernel.py fir_generic 6"
txt config3 python3 tools/bench_config3.py --hist
txt fir_fast python3 tools/bench_fir_fast.py
txt dropin python3 tools/bench_dropin.py 3000
txt streams_host_1 python3 tools/bench_streams_host.py 1 3000
txt streams_host_8 python3 tools/bench_streams_host.py 8 2000
txt streams_host_64 python3 tools/bench_streams_host.py 64 1000
txt config5 python3 tools/bench_config5.py
run bench_gpus2_shared python3 bench.py --gpus 2 --cpu-frames 0
run bench_gpus6_shared python3 bench.py --gpus 6 --frames 1024 --cpu-frames 0        # six ranks on this box's one GPU (the pool allows six GPU processes): the launch path, not a scaling number
run bench_torchrun2_shared python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --cpu-frames 0
fi
if [[ $part == *c* ]]; then
# (the elastic agent of torch.distributed.run holds the GPU open too: six ranks under it exceed the pool's limit of six GPU processes)
run bench_torchrun4_shared python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 4 --frames 1024 --cpu-frames 0
make -C qpsk_amd/csrc VARIANT=sbprof EXTRA=-DQPSK_SBLK_PROF > $O/sbprof_build.log 2>&1 && QPSK_HIP_LIB=qpsk_amd/libqpsk_hip_sbprof.so txt stream_block_profile python3 tools/bench_streams_host.py 1 60
fi
if [[ $part == *d* ]]; then      # the streams composition alone (after streamscan.hip)
prof prof_streams python3 tools/bench_streams.py
txt streams python3 tools/bench_streams.py
QPSK_STREAM_SCAN=0 txt streams_apart python3 tools/bench_streams.py
txt streams_2560 python3 tools/bench_streams.py --streams 2560
QPSK_STREAM_SCAN=0 txt streams_2560_apart python3 tools/bench_streams.py --streams 2560
fi
if [[ $part == *e* ]]; then      # final refresh: the bench lines (with the streams key) and the streams path's counters and power
run bench python3 bench.py
run bench20 python3 bench.py --steps 20 --warmup 5
prof prof_bench python3 bench.py --cpu-frames 0
pmcrun pmc_fetch_streams FETCH_SIZE python3 tools/loop_kernel.py streams 0 4096 8
pmcrun pmc_write_streams WRITE_SIZE python3 tools/loop_kernel.py streams 0 4096 8
txt power_streams python3 tools/power_probe.py --cmd "python3 tools/loop_kernel.py streams 6"
fi
if [[ $part == *f* ]]; then      # after the streams' one carrier (carrier.h) and the zero-symbol stretches: bench lines, streams, drop-in
run bench python3 bench.py
run bench20 python3 bench.py --steps 20 --warmup 5
run bench8192 python3 bench.py --frames 8192 --cpu-frames 0 --no-shard
prof prof_bench python3 bench.py --cpu-frames 0
prof prof_bench20 python3 bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-shard --no-timing-modes
prof prof_streams python3 tools/bench_streams.py
txt streams python3 tools/bench_streams.py
QPSK_STREAM_CARRIER=0 txt streams_own_carrier python3 tools/bench_streams.py
QPSK_STREAM_SCAN=0 txt streams_apart python3 tools/bench_streams.py
txt streams_1024 python3 tools/bench_streams.py --streams 1024
QPSK_STREAM_SCAN=0 txt streams_1024_apart python3 tools/bench_streams.py --streams 1024
pmcrun pmc_fetch_streams FETCH_SIZE python3 tools/loop_kernel.py streams 0 4096 8
pmcrun pmc_write_streams WRITE_SIZE python3 tools/loop_kernel.py streams 0 4096 8
txt power_streams python3 tools/power_probe.py --cmd "python3 tools/loop_kernel.py streams 6"
txt dropin python3 tools/bench_dropin.py 3000
txt streams_host_1 python3 tools/bench_streams_host.py 1 3000
txt streams_host_8 python3 tools/bench_streams_host.py 8 2000
txt streams_host_64 python3 tools/bench_streams_host.py 64 1000
txt zero_stretches python3 tools/bench_streams_zero.py
make -C qpsk_amd/csrc VARIANT=sbprof EXTRA=-DQPSK_SBLK_PROF > $O/sbprof_build.log 2>&1 && QPSK_HIP_LIB=qpsk_amd/libqpsk_hip_sbprof.so txt stream_block_profile python3 tools/bench_streams_host.py 1 60
rm -f qpsk_amd/libqpsk_hip_sbprof.so
fi
# what travels back is capped at 64 MiB: keep the summaries, drop the per-launch traces except the streams' (per-call table) and rocprofv3's databases
find $O -name "*kernel_trace.csv" ! -path "*prof_streams*" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*.db" -delete; du -sh $O | tail -1
ls $O/*.failed 2>/dev/null; cut -c1-300 $O/bench.json 2>/dev/null
