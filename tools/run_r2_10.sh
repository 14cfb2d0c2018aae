cd $GRAFT_REPO_ROOT
P="QPSK_PIPE_V=2 QPSK_PIPE_LAYOUT_LO=0x22222 QPSK_PIPE_LAYOUT_HI=0x011022"
export QPSK_HIP_LIB=qpsk_amd/libqpsk_hip_prof.so
timeout -k 10 300 python tools/sweep.py --frames 8192 "$P QPSK_PIPE_DBG=1536" "$P QPSK_PIPE_DBG=1537" "$P QPSK_PIPE_DBG=1538" "$P QPSK_PIPE_DBG=1539" "QPSK_PIPE_V=2 QPSK_PIPE_DBG=2" "QPSK_PIPE_V=2 QPSK_PIPE_DBG=1" 2>&1 | grep -v amdgpu > gpurun_out/r2_sweep8.log
cat gpurun_out/r2_sweep8.log
