#!/bin/bash
# Where the prefetch kernel's FLAT form (kernels.h, LaneOffset<false>) starts to pay: MT_FLAT_FROM=0 (every launch flat)
# against MT_FLAT_FROM=1000000000 (none), us per step by batch size; then the driver's bench command both ways.
for rep in 1 2; do
  for ff in 0 1000000000; do
    echo "== MT_FLAT_FROM=$ff"
    MT_FLAT_FROM=$ff python tools/size_sweep.py 131072 262144 393216 524288 786432 1048576 2097152 4194304 2>&1 >/dev/null | grep -v amdgpu | cut -c1-260
  done
done
for rep in 1 2 3; do
  for ff in 0 1000000000; do
    echo -n "bench MT_FLAT_FROM=$ff "
    MT_FLAT_FROM=$ff python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])"
  done
done
