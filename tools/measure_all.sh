# measurement session behind profiles/ (run on the GPU box, one gpurun call); then: python tools/collect_profiles.py <tag>
# Every pass clears its output directory first and leaves <dir>.failed behind when it did not complete, so that a stale
# summary of an earlier call (gpurun merges gpurun_out/ across calls) is never taken for this session's.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
rm -f $O/*.failed
run() { name=$1; shift; "$@" > $O/$name.json 2> $O/$name.err || echo FAILED > $O/$name.failed; }
run bench python3 bench.py
run bench20 python3 bench.py --steps 20 --warmup 5 --cpu-frames 0
run bench8192 python3 bench.py --frames 8192 --cpu-frames 0 --no-shard
run bench_gpus2_shared python3 bench.py --gpus 2 --cpu-frames 0     # two ranks on this box's one GPU: the launch path, not a scaling number
prof() { d=$1; shift; rm -rf $O/$d; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$d -- "$@" > $O/$d.log 2>&1 || echo FAILED > $O/$d.failed; }
prof prof_bench python3 bench.py --cpu-frames 0        # bench.py's default steps and warmup (config 2, then the 8192-frame shard): the same command, CPU leg off
prof prof_bench20 python3 bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-shard   # the driver's command
prof prof_8192 python3 bench.py --frames 8192 --cpu-frames 0 --no-shard
prof prof_fft python3 tools/sweep.py --timing fft
prof prof_hist2 python3 tools/sweep.py --timing hist
prof prof_streams3 python3 tools/bench_streams.py
pmcrun() { d=$1; c=$2; shift 2; rm -rf $O/$d; timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/$d -- "$@" > $O/$d.log 2>&1 || echo FAILED > $O/$d.failed; }
pmcrun pmc_fetch FETCH_SIZE python3 bench.py --steps 5 --warmup 1 --cpu-frames 0 --no-parity --no-shard
pmcrun pmc_write WRITE_SIZE python3 bench.py --steps 5 --warmup 1 --cpu-frames 0 --no-parity --no-shard
pmcrun pmc_fetch_8192 FETCH_SIZE python3 bench.py --frames 8192 --steps 5 --warmup 1 --cpu-frames 0 --no-parity --no-shard
pmcrun pmc_write_8192 WRITE_SIZE python3 bench.py --frames 8192 --steps 5 --warmup 1 --cpu-frames 0 --no-parity --no-shard
# SQ counters of the two receive kernels (python tools/collect_hist.py gpurun_out sq_4096 / sq_8192)
for shape in 4096 8192; do
  pmcrun sq_${shape}_a "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" python3 bench.py --frames $shape --steps 5 --warmup 1 --cpu-frames 0 --no-parity --no-shard
  pmcrun sq_${shape}_b "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU" python3 bench.py --frames $shape --steps 5 --warmup 1 --cpu-frames 0 --no-parity --no-shard
  pmcrun sq_${shape}_c "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" python3 bench.py --frames $shape --steps 5 --warmup 1 --cpu-frames 0 --no-parity --no-shard
done
pmcrun pmc_fft_fetch FETCH_SIZE python3 tools/sweep.py --frames 4096 --timing fft --rounds 1 --per-round 2
pmcrun pmc_fft_write WRITE_SIZE python3 tools/sweep.py --frames 4096 --timing fft --rounds 1 --per-round 2
txt() { name=$1; shift; timeout -k 10 300 "$@" > $O/$name.log 2>&1 || echo FAILED > $O/$name.failed; }
txt fir_wave_profile_final python3 tools/fir_wave_profile.py
txt lean_profile_8192 python3 tools/lean_profile.py 8192
txt lean_profile_4096 python3 tools/lean_profile.py 4096
txt power_probe python3 tools/power_probe.py 4096 8192
QPSK_HIP_LIB=qpsk_amd/libqpsk_hip_prof.so txt power_ablations python3 tools/power_probe.py 8192 8192:1 8192:16384 8192:32768 8192:49153
txt pitch_sweep python3 tools/pitch_sweep.py --frames 8192
QPSK_HIP_LIB=qpsk_amd/libqpsk_hip_prof.so txt pitch_sweep_floor python3 tools/pitch_sweep.py --frames 8192 --dbg 49153
# micro-benchmarks are built on the box (build_ubench/ does not travel: .gpurunignore)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench_fetch.hip -o /tmp/ubench_fetch > $O/ubench_fetch_build.log 2>&1 && txt ubench_fetch /tmp/ubench_fetch
txt fir_fast python3 tools/bench_fir_fast.py
txt dropin python3 tools/bench_dropin.py 2000
txt config5 python3 tools/bench_config5.py
ls $O/*.failed 2>/dev/null
cat $O/bench.json | cut -c1-400; tail -3 $O/dropin.log
