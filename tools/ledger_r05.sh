#!/bin/bash
# Round 5: the un-attacked joules of the 8192-frame receive kernel, one experiment each.  Measurement build
# (make -C qpsk_amd/csrc profile); steady-state board power x steady-state time per launch, as tools/ledger.sh.
#     bash tools/ledger_r05.sh       -> gpurun_out/ledger_r05.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/ledger_r05.txt
[ -f qpsk_amd/libqpsk_hip_prof.so ] || make -C qpsk_amd/csrc profile > gpurun_out/prof_build.log 2>&1
echo "== correctness of the variants that claim the right result (LDS-DMA staging, load order)" > $O
QPSK_HIP_LIB=qpsk_amd/libqpsk_hip_prof.so timeout -k 10 200 python3 tools/check_lean_variants.py 2>&1 | grep -v amdgpu.ids >> $O
if [ ${PIPESTATUS[0]} -ne 0 ] || grep -q "Memory access fault\|DIFFERS" $O; then
    echo "variant check FAILED: nothing else is run in this call" >> $O
    tail -20 $O
    exit 1
fi
{
# NOTE (round 6): bit 131072 selected LDS-DMA staging only in round 5's FIRST build (the one profiles/r05_energy_ledger.txt session 1-2 were taken on);
# DMA has been the product default behind QPSK_LEAN_DMA since, the bit no longer exists, and "8192" vs "8192:0:lean_dma=0" is the pair to run today.
echo "== measurement build.  QPSK_PIPE_DBG: 0 the kernel; 131072 LDS-DMA window staging (first round-5 build only, see the note above); 65536 no staging writes; 262144 no symbol stores; 524288 no hand-over"
echo "   write of the symbols; 1048576 loads frame-alternating (1 KB visits); 2097152 a workgroup's frames a grid apart; 3 floor (no filter arithmetic, no recurrence);"
echo "   pitch=16448: frames 16384 + 64 samples apart"
QPSK_HIP_LIB=qpsk_amd/libqpsk_hip_prof.so timeout -k 10 900 python3 tools/power_probe.py 8192 8192:0:lean_dma=0 8192:65536 8192 8192:0:lean_dma=0 2>&1 | grep -v amdgpu.ids
} >> $O 2>&1
tail -5 $O
