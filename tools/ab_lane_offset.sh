#!/bin/bash
# A/B of kernel builds on one box: the library as built against other builds of it, us per step by batch size
# (tools/size_sweep.py: segment / steady incl. resets / fused), the library's own dispatch.
#   tools/ab_lane_offset.sh [name ...]     names = tools/_build/libmanytor_hip_<name>.so, built by hand, e.g. with
#   -DMT_PLAIN_LANE_OFFSET=1 (kernels.h, LaneOffset: no renewal of the lane offset anywhere) = "plainoff", the default.
# (The loads-only / stores-only / non-volatile renewal variants of profiles/r03_ab_lane_offset_2_loads_stores.txt were
# experiment builds of an earlier revision of kernels.h.)
names=${*:-plainoff}
for rep in 1 2; do
  for name in "" $names; do
    lib=${name:+$PWD/tools/_build/libmanytor_hip_$name.so}
    echo "== ${name:-default build}"
    MT_LIB_OVERRIDE=$lib python tools/size_sweep.py 131072 262144 1048576 4194304 2>&1 >/dev/null | grep -v amdgpu | cut -c1-260
    MT_LIB_OVERRIDE=$lib python tools/size_sweep.py --dh7 1048576 2>&1 >/dev/null | grep -v amdgpu | cut -c1-260
  done
done
