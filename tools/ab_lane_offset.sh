#!/bin/bash
# A/B of the row addressing on one box: the library as built against builds with -DMT_PLAIN_LANE_OFFSET=<mode>
# (kernels.h, lane_offset): 1 = the offset as the optimiser leaves it (one v_lshl_add_u64 per access outside the entry
# block), 2 = the renewal without `volatile`, 3 = renewal on loads only, 4 = on stores only.  us per step, the library's
# own dispatch.   tools/ab_lane_offset.sh [name ...]   names = tools/_build/libmanytor_hip_<name>.so, built by hand with
# those flags; default: plainoff
names=${*:-plainoff}
for rep in 1 2; do
  for name in "" $names; do
    lib=${name:+$PWD/tools/_build/libmanytor_hip_$name.so}
    echo "== ${name:-default build}"
    MT_LIB_OVERRIDE=$lib python tools/size_sweep.py 131072 262144 1048576 4194304 2>&1 >/dev/null | grep -v amdgpu | cut -c1-260
    MT_LIB_OVERRIDE=$lib python tools/size_sweep.py --dh7 1048576 2>&1 >/dev/null | grep -v amdgpu | cut -c1-260
  done
done
