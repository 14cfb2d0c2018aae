cd $GRAFT_REPO_ROOT
L=qpsk_amd
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "geometries or smallest or randomised or full_size or golden or environment" > gpurun_out/r2_t4.log 2>&1; tail -3 gpurun_out/r2_t4.log
export QPSK_PIPE_V=2
for F in 8192 4096; do
echo "== frames $F (pipe_v=2)"
timeout -k 10 300 python tools/ab_libs.py --frames $F --rounds 20 $L/libqpsk_hip.so $L/libqpsk_hip_cxx.so $L/libqpsk_hip_a2f9.so $L/libqpsk_hip_a2f6.so 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r2_ab3.log 2>&1
cat gpurun_out/r2_ab3.log
unset QPSK_PIPE_V
P="QPSK_PIPE_V=2"
timeout -k 10 300 python tools/sweep.py --frames 8192 "$P" "$P QPSK_PIPE_LAYOUT_LO=0x22222 QPSK_PIPE_LAYOUT_HI=0x011022" "$P QPSK_PIPE_LAYOUT_LO=0x22222 QPSK_PIPE_LAYOUT_HI=0x100122" "$P QPSK_PIPE_LAYOUT_LO=0x22222 QPSK_PIPE_LAYOUT_HI=0x000222" > gpurun_out/r2_sweep5.log 2>&1
cat gpurun_out/r2_sweep5.log
timeout -k 10 300 python tools/fir_wave_profile.py "pipe_v=2" > gpurun_out/r2_prof6.log 2>&1
grep -A12 "8192 frames" gpurun_out/r2_prof6.log
