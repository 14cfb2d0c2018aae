cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "geometries or smallest or randomised or full_size or golden or environment" > gpurun_out/r2_t3.log 2>&1; tail -3 gpurun_out/r2_t3.log
timeout -k 10 300 python tools/sweep.py --frames 8192 "QPSK_PIPE_V=1" "QPSK_PIPE_V=2" "QPSK_PIPE_V=2 QPSK_PIPE_LAYOUT_LO=0x22222 QPSK_PIPE_LAYOUT_HI=0x011022" "QPSK_PIPE_V=2 QPSK_PIPE_LAYOUT_LO=0x22222 QPSK_PIPE_LAYOUT_HI=0x100122" "QPSK_PIPE_V=2 QPSK_PIPE_LAYOUT_LO=0x22222 QPSK_PIPE_LAYOUT_HI=0x110022" "QPSK_PIPE_V=2 QPSK_PIPE_LAYOUT_LO=0x22222 QPSK_PIPE_LAYOUT_HI=0x000222" > gpurun_out/r2_sweep3.log 2>&1
cat gpurun_out/r2_sweep3.log
QPSK_HIP_LIB=qpsk_amd/libqpsk_hip_d1f6.so timeout -k 10 300 python tools/sweep.py --frames 8192 "QPSK_PIPE_V=2" "QPSK_PIPE_V=2 QPSK_PIPE_LAYOUT_LO=0x22233 QPSK_PIPE_LAYOUT_HI=0x22" "QPSK_PIPE_V=2 QPSK_PIPE_LAYOUT_LO=0x23222 QPSK_PIPE_LAYOUT_HI=0x22" "QPSK_PIPE_V=2 QPSK_PIPE_LAYOUT_LO=0x22222 QPSK_PIPE_LAYOUT_HI=0x33" > gpurun_out/r2_sweep3b.log 2>&1
cat gpurun_out/r2_sweep3b.log
timeout -k 10 300 python tools/fir_wave_profile.py "pipe_v=2" "pipe_v=2 pipe_layout_lo=0x22222 pipe_layout_hi=0x011022" "pipe_v=2 pipe_layout_lo=0x22222 pipe_layout_hi=0x000222" > gpurun_out/r2_prof4.log 2>&1
grep -A12 "8192 frames" gpurun_out/r2_prof4.log
