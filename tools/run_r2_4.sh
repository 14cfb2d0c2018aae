cd $GRAFT_REPO_ROOT
L=qpsk_amd
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "geometries or smallest or randomised or full_size or golden or environment" > gpurun_out/r2_t2.log 2>&1; tail -3 gpurun_out/r2_t2.log
export QPSK_PIPE_V=2
for F in 8192 4096; do
echo "== frames $F (pipe_v=2)"
timeout -k 10 300 python tools/ab_libs.py --frames $F --rounds 20 $L/libqpsk_hip.so $L/libqpsk_hip_d1f6.so $L/libqpsk_hip_d2f6.so 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r2_ab2.log 2>&1
cat gpurun_out/r2_ab2.log
unset QPSK_PIPE_V
timeout -k 10 300 python tools/sweep.py --frames 8192 "QPSK_PIPE_V=1" "QPSK_PIPE_V=2" "QPSK_PIPE_V=2 QPSK_PIPE_LAYOUT_LO=139810 QPSK_PIPE_LAYOUT_HI=69666" > gpurun_out/r2_sweep2.log 2>&1
cat gpurun_out/r2_sweep2.log
timeout -k 10 300 python tools/fir_wave_profile.py "pipe_v=2" "pipe_v=2 pipe_layout_lo=139810 pipe_layout_hi=69666" > gpurun_out/r2_prof3.log 2>&1
cat gpurun_out/r2_prof3.log
