#!/bin/bash
# A long differential-fuzzing session on the GPU box (round 6: bash tools/fuzz_long.sh [first seed]): batch cases (histogram-mode contexts are called
# twice, the second call through the one-pass route), streams, stages; one summary line each into gpurun_out/fuzz_long/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/fuzz_long; mkdir -p $O; S=${1:-20000}
timeout -k 10 1000 python3 tools/fuzz.py --cases 3600 --seed $S > $O/batch.log 2>&1 || echo "batch FAILED" ; tail -1 $O/batch.log | cut -c1-300
timeout -k 10 300 python3 tools/fuzz.py --kind streams --cases 1500 --seed $((S + 5000)) > $O/streams.log 2>&1 || echo "streams FAILED"; tail -1 $O/streams.log | cut -c1-300
timeout -k 10 100 python3 tools/fuzz.py --kind stages --cases 1500 --seed $((S + 8000)) > $O/stages.log 2>&1 || echo "stages FAILED"; tail -1 $O/stages.log | cut -c1-300
