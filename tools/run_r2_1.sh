set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2_t1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_t1.log
tail -5 gpurun_out/r2_t1.log
timeout -k 10 300 python tools/sweep.py --frames 4096 "QPSK_PIPE_V=1" "QPSK_PIPE_V=2" "QPSK_PIPE_V=2 QPSK_PIPE_NF=4" "QPSK_PIPE_V=2 QPSK_PIPE_NF=6" > gpurun_out/r2_sweep4096.log 2>&1
cat gpurun_out/r2_sweep4096.log
timeout -k 10 300 python tools/sweep.py --frames 8192 "QPSK_PIPE_V=1" "QPSK_PIPE_V=2" "QPSK_PIPE_V=2 QPSK_PIPE_NF=8" "QPSK_PIPE_V=2 QPSK_PIPE_G=16" > gpurun_out/r2_sweep8192.log 2>&1
cat gpurun_out/r2_sweep8192.log
