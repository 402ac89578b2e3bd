#!/bin/bash
# A/B of the episode end in bench.py on ONE box: chains that stay forked across calls (default) vs joined at the end of every
# call (MT_LAZY_CHAINS=0), and "snapshot the returns, gather, reset" (default) vs "reset, then gather MT_F_LAST_RETURN in
# place" (--reset-before-gather).  Prints value (1e10 env-steps/s), wall us per step, device us per step, gather us.
for steps in "--steps 20 --warmup 5" "--steps 1000 --warmup 50"; do
echo "== $steps"
for rep in 1 2; do
for cfg in "lazy+inplace::--reset-before-gather" "lazy+snapshot::" "eager+inplace:MT_LAZY_CHAINS=0:--reset-before-gather" "eager+snapshot:MT_LAZY_CHAINS=0:"; do
  name=${cfg%%:*}; rest=${cfg#*:}; envv=${rest%%:*}; flag=${rest#*:}
  env $envv python bench.py --gpus 1 $steps --no-secondary --no-cpu-baseline $flag > gpurun_out/ab_$name.json 2>/dev/null
  python -c "
import json
d=json.load(open('gpurun_out/ab_$name.json')); r=d['roofline']; print('$name', round(d['value']/1e10,4), round(d['ms_per_step']*1e3,2), round(r['avg_kernel_us'],2), round(d['gather_us'],1))"
done; done; done
