cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/r3_e19.log 2>&1
for rep in 1 2; do
for v in _base ""; do
  echo "== variant '$v'" >> $O/r3_e19.log
  QPSK_HIP_LIB=qpsk_amd/libqpsk_hip$v.so python3 tools/power_probe.py 4096 8192 2>&1 | grep -v "amdgpu\|^idle\|t+1s" >> $O/r3_e19.log
done
done
grep -v amdgpu $O/r3_e19.log
