cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
: > $O/r3_e18.log
for rep in 1 2; do
for v in "" _ntl _ntls _sc; do
  echo "== variant '$v'" >> $O/r3_e18.log
  QPSK_HIP_LIB=qpsk_amd/libqpsk_hip$v.so python3 tools/power_probe.py 8192 2>&1 | grep -v "amdgpu\|^idle\|t+1s" >> $O/r3_e18.log
done
done
python3 tools/ab_libs.py --frames 8192 --rounds 40 qpsk_amd/libqpsk_hip.so qpsk_amd/libqpsk_hip_ntl.so qpsk_amd/libqpsk_hip_ntls.so qpsk_amd/libqpsk_hip_sc.so >> $O/r3_e18.log 2>&1
cat $O/r3_e18.log | grep -v amdgpu
