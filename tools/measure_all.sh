# measurement session behind profiles/ (run on the GPU box, one gpurun call); then: python tools/collect_profiles.py <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
python3 bench.py > $O/bench.json 2> $O/bench.err
python3 bench.py --steps 20 --warmup 5 --cpu-frames 0 > $O/bench20.json 2> $O/bench20.err
python3 bench.py --frames 8192 --cpu-frames 0 > $O/bench8192.json 2> $O/bench8192.err
python3 bench.py --gpus 2 --cpu-frames 0 > $O/bench_gpus2_shared.json 2> $O/bench_gpus2_shared.err     # two ranks on this box's one GPU: the launch path, not a scaling number
prof() { d=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$d -- "$@" > $O/$d.log 2>&1; }
prof prof_bench python3 bench.py --cpu-frames 0        # bench.py's default steps and warmup: the same command, CPU leg off
prof prof_bench20 python3 bench.py --steps 20 --warmup 5 --cpu-frames 0   # the driver's command
prof prof_8192 python3 bench.py --frames 8192 --cpu-frames 0
prof prof_fft python3 tools/sweep.py --timing fft
prof prof_hist2 python3 tools/sweep.py --timing hist
prof prof_streams3 python3 tools/bench_streams.py
pmcrun() { d=$1; c=$2; shift 2; rm -rf $O/$d; timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/$d -- "$@" > $O/$d.log 2>&1; }
pmcrun pmc_fetch FETCH_SIZE python3 bench.py --steps 5 --warmup 1 --cpu-frames 0 --no-parity
pmcrun pmc_write WRITE_SIZE python3 bench.py --steps 5 --warmup 1 --cpu-frames 0 --no-parity
pmcrun pmc_fetch_8192 FETCH_SIZE python3 bench.py --frames 8192 --steps 5 --warmup 1 --cpu-frames 0 --no-parity
pmcrun pmc_write_8192 WRITE_SIZE python3 bench.py --frames 8192 --steps 5 --warmup 1 --cpu-frames 0 --no-parity
# SQ counters of the two receive kernels (profiles/r02_pipe2_sq_counters.txt, r02_pipe1_sq_counters.txt: python tools/collect_hist.py gpurun_out sq_)
for shape in 4096 8192; do
  pmcrun sq_${shape}_a "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" python3 bench.py --frames $shape --steps 5 --warmup 1 --cpu-frames 0 --no-parity
  pmcrun sq_${shape}_b "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU" python3 bench.py --frames $shape --steps 5 --warmup 1 --cpu-frames 0 --no-parity
  pmcrun sq_${shape}_c "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" python3 bench.py --frames $shape --steps 5 --warmup 1 --cpu-frames 0 --no-parity
done
pmcrun pmc_fft_fetch FETCH_SIZE python3 tools/sweep.py --frames 4096 --timing fft --rounds 1 --per-round 2
pmcrun pmc_fft_write WRITE_SIZE python3 tools/sweep.py --frames 4096 --timing fft --rounds 1 --per-round 2
timeout -k 10 300 python3 tools/fir_wave_profile.py > $O/fir_wave_profile_final.log 2>&1
timeout -k 10 300 python3 tools/bench_dropin.py 2000 > $O/dropin.log 2>&1
timeout -k 10 300 python3 tools/bench_config5.py > $O/config5.log 2>&1
cat $O/bench.json; cat $O/bench8192.json | cut -c1-300; cat $O/bench_gpus2_shared.json | cut -c1-300; tail -3 $O/dropin.log
