cd $GRAFT_REPO_ROOT
P="QPSK_PIPE_V=2"
A="QPSK_PIPE_LAYOUT_LO=0x22222 QPSK_PIPE_LAYOUT_HI=0x011022"
B="QPSK_PIPE_LAYOUT_LO=0x22222 QPSK_PIPE_LAYOUT_HI=0x100122"
timeout -k 10 300 python tools/sweep.py --frames 8192 "$P" "$P $A" "$P $A QPSK_PIPE_DBG=512" "$P $A QPSK_PIPE_DBG=1024" "$P $A QPSK_PIPE_DBG=1536" "$P $B QPSK_PIPE_DBG=512" "$P $B QPSK_PIPE_DBG=1536" "$P QPSK_PIPE_DBG=512" > gpurun_out/r2_sweep6.log 2>&1
cat gpurun_out/r2_sweep6.log
timeout -k 10 300 python tools/fir_wave_profile.py "pipe_v=2 pipe_layout_lo=0x22222 pipe_layout_hi=0x011022 pipe_dbg=1536" > gpurun_out/r2_prof7.log 2>&1
grep -A12 "8192 frames" gpurun_out/r2_prof7.log
