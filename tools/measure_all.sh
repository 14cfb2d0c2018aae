# measurement session behind profiles/ (run on the GPU box, one gpurun call); then: python tools/collect_profiles.py <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
set -e
python3 bench.py > $O/bench.json 2> $O/bench.err
prof() { d=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$d -- "$@" > $O/$d.log 2>&1; }
prof prof_bench python3 bench.py --cpu-frames 0        # bench.py's default steps and warmup: the same command, CPU leg off
prof prof_8192 python3 tools/sweep.py --frames 8192
prof prof_fft python3 tools/sweep.py --timing fft
prof prof_hist2 python3 tools/sweep.py --timing hist
prof prof_streams3 python3 tools/bench_streams.py
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 5 --warmup 1 --cpu-frames 0 --no-parity > $O/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 5 --warmup 1 --cpu-frames 0 --no-parity > $O/pmc_write.log 2>&1
timeout -k 10 300 python3 tools/fir_wave_profile.py > $O/fir_wave_profile_final.log 2>&1
cat $O/bench.json
