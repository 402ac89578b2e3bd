#!/usr/bin/env python3
"""Would a placement probe at mt_create help?  Allocate several 4 194 304-arm arenas and HOLD them (so they sit on
different physical pages), time each; then free all and repeat with alloc / free cycles (the same pages come back).
    python tools/placement_hold.py > gpurun_out/placement_hold.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import manytor_amd as m  # noqa: E402

N = int(os.environ.get("MT_PLACE_N", 4194304))


def time_engine(e, reps=2):
    e.reset_random(1, 0)
    e.rollout(30, 1, 0)
    e.sync()
    out = []
    for _ in range(reps):
        e.timer_start()
        e.rollout(20, 1, 30)
        out.append(round(e.timer_stop() * 1e3 / 20, 2))
    return out


res = {"n": N, "chains": os.environ.get("MT_CHAINS", "default")}
held = [m.StepEngine(N, 7) for _ in range(6)]
res["held_simultaneously"] = [time_engine(e) for e in held]
res["held_again_in_reverse_order"] = [time_engine(e) for e in reversed(held)][::-1]
for e in held:
    e.close()
cyc = []
for _ in range(6):
    e = m.StepEngine(N, 7)
    cyc.append(time_engine(e))
    e.close()
res["alloc_free_cycles"] = cyc
print(json.dumps(res, indent=1))
