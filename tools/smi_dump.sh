#!/bin/bash
# Measurement helper (GPU box): everything amd-smi reports about the GPU (power, clocks, voltages, throttle status, limits) while one
# tools/loop_kernel.py workload runs back to back: tools/smi_dump.sh <what> [frames]   -> gpurun_out/smi_<what>_<frames>.txt
what=$1; frames=${2:-4096}
out=gpurun_out/smi_${what}_${frames}.txt
python3 tools/loop_kernel.py $what 9 $frames > $out.child 2>&1 &
pid=$!
sleep 5
{ echo "== amd-smi metric (while $what $frames runs)"; timeout 20 amd-smi metric -g 0 2>&1; echo "== rocm-smi"; timeout 20 rocm-smi --showpower --showclocks --showvoltage 2>&1 | grep -v "^=\|^$"; } > $out
wait $pid
cat $out.child >> $out; rm -f $out.child
