#!/bin/bash
# Register / scratch metadata of the kernels of one .hip source (device-only assembly): tools/kmeta.sh rx_fused.hip [-DQPSK_PIPE_PROFILE ...]
src=$1; shift
out=/tmp/kmeta_$$.s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 "$@" --cuda-device-only -S $(dirname $0)/../qpsk_amd/csrc/$src -o $out 2>/dev/null
grep -E "^\s+\.(vgpr_count|vgpr_spill_count|private_segment_fixed_size|sgpr_spill_count|sgpr_count|name):" $out | paste - - - - - - | sed 's/ \+/ /g; s/_ZN4qpsk[0-9]*//' | cut -c1-200
rm -f $out
