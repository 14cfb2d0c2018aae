cd $GRAFT_REPO_ROOT
P="QPSK_PIPE_V=2"
for lib in a2f6; do
echo "== $lib"
QPSK_HIP_LIB=qpsk_amd/libqpsk_hip_$lib.so timeout -k 10 300 python tools/sweep.py --frames 8192 "$P" "$P QPSK_PIPE_LAYOUT_LO=0x22233 QPSK_PIPE_LAYOUT_HI=0x22 QPSK_PIPE_DBG=1536" "$P QPSK_PIPE_LAYOUT_LO=0x23223 QPSK_PIPE_LAYOUT_HI=0x22 QPSK_PIPE_DBG=1536" "$P QPSK_PIPE_LAYOUT_LO=0x21333 QPSK_PIPE_LAYOUT_HI=0x22 QPSK_PIPE_DBG=1536" "$P QPSK_PIPE_LAYOUT_LO=0x22233 QPSK_PIPE_LAYOUT_HI=0x22" 2>&1 | grep -v amdgpu
done > gpurun_out/r2_sweep7.log 2>&1
cat gpurun_out/r2_sweep7.log
