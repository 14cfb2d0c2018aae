cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout -k 10 400 python3 tools/sweep.py --frames 4096 --timing fft --rounds 20 --per-round 6 "QPSK_EST_WAVES=8" "QPSK_EST_WAVES=10" "QPSK_EST_WAVES=12" "QPSK_FFT_FUSED=0" > $O/config3_est_waves.txt 2>&1 || echo FAILED
cat $O/config3_est_waves.txt | tail -5
timeout -k 10 400 python3 tools/sweep.py --frames 4096 --timing fixed --rounds 20 --per-round 6 "" > $O/config2_same_session.txt 2>&1; tail -1 $O/config2_same_session.txt
