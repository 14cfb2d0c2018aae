#!/bin/bash
# Energy ledger of the 8192-frame receive kernel: board power, per-XCD clocks, PPT throttle activity and steady-state
# time per launch of rx_lean_kernel with one part left out at a time (measurement build: make -C qpsk_amd/csrc profile), the
# same for config 2's kernel, and the synthetic operation-mix kernel in the SAME session.  Run on the GPU box:
#     bash tools/ledger.sh       -> gpurun_out/ledger.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/ledger.txt
# the measurement library is a build product that does not travel (.gpurunignore): build it here
[ -f qpsk_amd/libqpsk_hip_prof.so ] || make -C qpsk_amd/csrc profile > gpurun_out/prof_build.log 2>&1
{
echo "== product kernels, measurement build (QPSK_PIPE_DBG: 1 no filter multiplies/adds, 16384 no window reads after the first two blocks, 32768 no flush arithmetic,"
echo "   65536 no window staging writes, 2 no Costas recurrence; WRONG results by construction)"
QPSK_HIP_LIB=qpsk_amd/libqpsk_hip_prof.so timeout -k 10 500 python3 tools/power_probe.py 8192 8192:1 8192:16384 8192:32768 8192:65536 8192:2 8192:3 8192:16386 8192:32770 8192:65538 4096 4096:1 4096:2 4096:3 2>&1 | grep -v amdgpu.ids
echo "== the product library itself (no measurement code)"
timeout -k 10 100 python3 tools/power_probe.py 8192 4096 2>&1 | grep -v amdgpu.ids
echo "== synthetic operation-mix kernel (tools/ubench_valu_power.hip): seconds, waves per SIMD, stream, LDS reads, random data, workgroups per CU"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench_valu_power.hip -o /tmp/ubench_valu 2>&1 | tail -2
timeout -k 10 300 python3 tools/power_probe.py --cmd "/tmp/ubench_valu 6 4 1 1 1 3" "/tmp/ubench_valu 6 4 1 1 1 2" "/tmp/ubench_valu 6 4 0 1 1 2" "/tmp/ubench_valu 6 4 2 0 1 2" "/tmp/ubench_valu 6 3 1 1 1 1" 2>&1 | grep -v amdgpu.ids
} > $O 2>&1
tail -5 $O
