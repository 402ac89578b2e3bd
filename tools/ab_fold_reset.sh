#!/bin/bash
# A/B of the episode boundary folded into mt_rollout's launches, as bench.py times it (20-step regions).
#   tools/ab_fold_reset.sh chains   : 1 M arms, MT_DEFER_RESET_CHAINS 0 / 1 (the chained launch-per-step form)
#   tools/ab_fold_reset.sh shard    : 131 072 arms, MT_DEFER_RESET 0 / 1 (the multi-step form), episode end at / in the middle of the region
show() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('value %.4g wall %.3f dev %.3f step %.2f head %s' % (d['value'], d['ms_per_step']*1e3, d['device_ms_per_step']*1e3, r['avg_kernel_us'], r['episode_first_launches_outside_laps']))"; }
if [ "$1" != shard ]; then
  for v in 0 1 0 1; do echo "1M MT_DEFER_RESET_CHAINS=$v"; MT_DEFER_RESET_CHAINS=$v python3 bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline $2 2>/dev/null | show; done
fi
if [ "$1" != chains ]; then
  for ph in 0 10; do for v in 0 1 0 1; do echo "131072 phase $ph MT_DEFER_RESET=$v"; MT_DEFER_RESET=$v python3 bench.py --steps 20 --warmup 5 --envs-per-gpu 131072 --episode-phase $ph --no-secondary --no-cpu-baseline 2>/dev/null | show; done; done
fi
