#!/bin/bash
# Round 6, the serial wave's step (GPU box): bash tools/measure_r06_step.sh
#   the stream alone on a CU, every variant (tools/ubench_step.py); the kernel with one and two lanes per loop and round 5's build, in one process
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
python3 tools/ubench_step.py --align > $O/ubench_build.log 2>&1 || { echo "ubench build failed"; tail -5 $O/ubench_build.log; exit 1; }
: > $O/step_cost.txt
for b in build_ubench/step/r05_as_shipped.bin build_ubench/step/0*.bin build_ubench/step/[1-5]*.bin build_ubench/step/a*.bin build_ubench/step/b*.bin; do
  timeout -k 10 120 $b $(basename $b .bin) >> $O/step_cost.txt 2>&1 || { echo "FAILED $b" >> $O/step_cost.txt; break; }
done
echo "ubench done"; head -4 $O/step_cost.txt | cut -c1-250
for F in 4096 8192 2048; do
  timeout -k 10 300 python3 tools/sweep.py --frames $F --rounds 12 "QPSK_LEAN_PAIR=0" "QPSK_LEAN_PAIR=1" "QPSK_LEAN_PAIR=2" > $O/pair_sweep_$F.txt 2>&1 || echo "sweep $F FAILED"
  tail -3 $O/pair_sweep_$F.txt
done
timeout -k 10 300 python3 tools/ab_libs.py --frames 4096 --rounds 40 qpsk_amd/libqpsk_hip.so qpsk_amd/libqpsk_hip_r05.so > $O/ab_r05_4096.txt 2>&1 || echo "ab FAILED"
tail -4 $O/ab_r05_4096.txt
timeout -k 10 300 python3 tools/ab_libs.py --frames 8192 --rounds 40 qpsk_amd/libqpsk_hip.so qpsk_amd/libqpsk_hip_r05.so > $O/ab_r05_8192.txt 2>&1 || echo "ab FAILED"
tail -4 $O/ab_r05_8192.txt
