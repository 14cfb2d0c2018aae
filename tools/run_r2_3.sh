cd $GRAFT_REPO_ROOT
L=qpsk_amd
for F in 8192 4096; do
echo "== frames $F"
timeout -k 10 300 python tools/ab_libs.py --frames $F --rounds 20 $L/libqpsk_hip.so $L/libqpsk_hip_d1f6.so $L/libqpsk_hip_d2f6.so $L/libqpsk_hip_d3f6.so 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r2_ab1.log 2>&1
cat gpurun_out/r2_ab1.log
timeout -k 10 300 python tools/fir_wave_profile.py "pipe_v=2" > gpurun_out/r2_prof2.log 2>&1
cat gpurun_out/r2_prof2.log
