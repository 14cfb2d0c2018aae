cd $GRAFT_REPO_ROOT
P="QPSK_PIPE_V=2"
timeout -k 10 300 python tools/sweep.py --frames 8192 "QPSK_PIPE_V=1" "$P" "$P QPSK_PIPE_DBG=256" "$P QPSK_PIPE_LAYOUT_LO=0x22222 QPSK_PIPE_LAYOUT_HI=0x011022" "$P QPSK_PIPE_LAYOUT_LO=0x22222 QPSK_PIPE_LAYOUT_HI=0x011022 QPSK_PIPE_DBG=256" "$P QPSK_PIPE_LAYOUT_LO=0x22222 QPSK_PIPE_LAYOUT_HI=0x100122" "$P QPSK_PIPE_LAYOUT_LO=0x22222 QPSK_PIPE_LAYOUT_HI=0x110022" "$P QPSK_PIPE_LAYOUT_LO=0x22222 QPSK_PIPE_LAYOUT_HI=0x000222" > gpurun_out/r2_sweep4.log 2>&1
cat gpurun_out/r2_sweep4.log
for lib in d1f6 d2f6; do
echo "== $lib"
QPSK_HIP_LIB=qpsk_amd/libqpsk_hip_$lib.so timeout -k 10 300 python tools/sweep.py --frames 8192 "$P" "$P QPSK_PIPE_LAYOUT_LO=0x21333 QPSK_PIPE_LAYOUT_HI=0x22" "$P QPSK_PIPE_LAYOUT_LO=0x22233 QPSK_PIPE_LAYOUT_HI=0x22" "$P QPSK_PIPE_LAYOUT_LO=0x22222 QPSK_PIPE_LAYOUT_HI=0x33"  2>&1 | grep -v amdgpu
done > gpurun_out/r2_sweep4b.log 2>&1
cat gpurun_out/r2_sweep4b.log
timeout -k 10 300 python tools/sweep.py --frames 4096 "QPSK_PIPE_V=1" "$P" "$P QPSK_PIPE_DBG=256" > gpurun_out/r2_sweep4c.log 2>&1
cat gpurun_out/r2_sweep4c.log
timeout -k 10 300 python tools/fir_wave_profile.py "pipe_v=2" "pipe_v=2 pipe_layout_lo=0x22222 pipe_layout_hi=0x011022" > gpurun_out/r2_prof5.log 2>&1
grep -A12 "8192 frames" gpurun_out/r2_prof5.log
