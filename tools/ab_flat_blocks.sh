#!/bin/bash
# Occupancy cap of the HBM-bound launches (engine.hip: flat_lds_pad, MT_FLAT_BLOCKS_PER_CU; 0 = none): the driver's bench
# command, the staged-action step and the big batches, interleaved on one box.
for rep in 1 2; do
  for c in 0 4 5 6; do
    echo -n "[cap $c] bench: "
    MT_FLAT_BLOCKS_PER_CU=$c python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('value %.4g  wall_us %.2f  device_us %.2f  frac %.3f' % (d['value'], d['ms_per_step']*1e3, d['roofline']['avg_kernel_us'], d['roofline']['frac']))"
  done
done
for rep in 1 2; do
  for c in 0 5 6; do
    echo "[cap $c] staged-action step:"; MT_FLAT_BLOCKS_PER_CU=$c python - <<'PY' 2>&1 | grep -v amdgpu
import os, sys
sys.path.insert(0, os.getcwd())
import bench, manytor_amd as m
for n in (1048576, 2097152):
    us, _ = bench.time_loaded_action_steps(m, n, m.REF_DH_TABLE, 51.3, 7, 0, 0x5EED, steps=400)
    print("   ", n, round(us, 2))
PY
    echo "[cap $c] sampled:"; MT_FLAT_BLOCKS_PER_CU=$c python tools/size_sweep.py 2097152 4194304 2>&1 >/dev/null | grep -v amdgpu | cut -c1-9,100-260
  done
done
