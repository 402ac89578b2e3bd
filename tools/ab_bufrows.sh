#!/bin/bash
# A/B on one box: the library as built against tools/_build/libmanytor_hip_<name>.so (default name: bufrows = -DMT_BUFFER_ROWS=1, the row
# accesses as buffer instructions): the driver's bench command, interleaved, then us per step by batch size.
name=${1:-bufrows}
show() { python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('value %.4g wall %.3f dev %.3f step %.2f' % (d['value'], d['ms_per_step']*1e3, d['device_ms_per_step']*1e3, r['avg_kernel_us']))"; }
for rep in 1 2 3; do
  for nm in "" $name; do
    lib=${nm:+$PWD/tools/_build/libmanytor_hip_$nm.so}
    echo -n "${nm:-default} 1M: "; MT_LIB_OVERRIDE=$lib python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | show
  done
done
for rep in 1 2; do
  for nm in "" $name; do
    lib=${nm:+$PWD/tools/_build/libmanytor_hip_$nm.so}
    for n in 65536 131072 262144 524288; do echo -n "${nm:-default} $n: "; MT_LIB_OVERRIDE=$lib python3 bench.py --steps 20 --warmup 5 --envs-per-gpu $n --episode-phase 10 --no-cpu-baseline --no-secondary 2>/dev/null | show; done
    echo -n "${nm:-default} 1M dof7: "; MT_LIB_OVERRIDE=$lib python3 bench.py --steps 20 --warmup 5 --dof 7 --no-cpu-baseline --no-secondary 2>/dev/null | show
  done
done
