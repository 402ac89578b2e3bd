#!/bin/bash
# Occupancy cap of the sampled-action prefetch kernel (engine.hip: step_blocks_per_cu; MT_BLOCKS_PER_CU overrides, 0 = none) by
# batch size: us per step (tools/size_sweep.py: segment / steady incl. resets / fused), interleaved on one box.
#   tools/ab_blocks_per_cu.sh "<caps>" <sizes ...>
caps=$1; shift
for rep in 1 2; do
  for c in $caps; do
    echo "== MT_BLOCKS_PER_CU=$c"
    MT_BLOCKS_PER_CU=$c python tools/size_sweep.py "$@" 2>&1 >/dev/null | grep -v amdgpu | sed -e "s/'kernel': '[^']*', //" | cut -c1-150
  done
done
