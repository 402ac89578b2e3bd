#!/usr/bin/env python3
"""us per step of mt_rollout on the DEFAULT dispatch by batch size, beside the pure one-launch-per-step form (MT_ROLLOUT_K=1)
and one launch per 50-step episode (mt_rollout_fused) -- the table VERDICT r3 #3 asks for (<= 5.87 us at 131 072 arms, <= 4.4
at 65 536, 1 M arms not worse than 35.4).  Timing as in tools/rollout_k_sweep.py: `idle_T*` = every segment starts on an idle
device, `b2b_T50` = 50-step episodes queued back to back, laps around the rollout calls only.
    python tools/default_dispatch_sweep.py [sizes ...] > gpurun_out/r04_variant_sweep.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import manytor_amd as m  # noqa: E402
from tools.rollout_k_sweep import measure  # noqa: E402


def fused(n, episodes=12):
    e = m.StepEngine(n, 7)
    e.reset_random(1, 0)
    for ep in range(20):
        e.rollout_fused(50, 1, 0)
        e.reset_random(1, ep)
    e.sync()
    e.lap_times()
    for r in range(episodes):
        e.reset_random(1, r)
        e.lap_begin()
        e.rollout_fused(50, 1, 0)
        e.lap_end()
    laps = sorted(e.lap_times())
    e.close()
    return round(laps[len(laps) // 2] * 1e3 / 50, 3)


def main():
    sizes = [int(v) for v in sys.argv[1:]] or [32768, 65536, 98304, 131072, 163840, 196608, 262144, 393216, 524288, 786432, 1048576]
    res = {}
    for n in sizes:
        row = {"default": measure(n, {}), "one_launch_per_step": measure(n, {"MT_ROLLOUT_K": "1"}), "fused_T50_b2b": fused(n)}
        res[n] = row
        print(n, json.dumps(row), file=sys.stderr, flush=True)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
