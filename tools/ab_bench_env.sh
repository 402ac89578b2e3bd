#!/bin/bash
# The driver's bench command under alternative environment settings, interleaved on one box.
#   tools/ab_bench_env.sh "MT_GRAPH=0" "MT_GRAPH=1" ...     (each argument: space-separated VAR=value list, "" = defaults)
for rep in 1 2 3; do
  for setting in "$@"; do
    echo -n "[$setting] "
    env $setting python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('value %.4g  wall_us %.2f  device_us %.2f  frac %.3f' % (d['value'], d['ms_per_step']*1e3, d['roofline']['avg_kernel_us'], d['roofline']['frac']))"
  done
done
