#!/usr/bin/env python3
"""mt_create's placement by trial (engine.hip: a handle of > 3 M arms is created, a few real steps are timed on it, and it is
destroyed and created again while they are slow) against plain creation: us per step of
successive fresh 4 194 304-arm engines of one process, the two settings interleaved.   python tools/placement_trial.py [engines]"""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import manytor_amd as m  # noqa: E402

n = int(os.environ.get("MT_PLACE_N", 4194304))
res = {"plain": [], "by trial": []}
cost = {"plain": [], "by trial": []}
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    for name, setting in (("plain", "0"), ("by trial", "2")):
        os.environ["MT_PLACEMENT_PROBE"] = setting
        t0 = time.perf_counter()
        e = m.StepEngine(n, 7)
        e.sync()
        cost[name].append((time.perf_counter() - t0) * 1e3)
        e.reset_random(1, 0)
        for _ in range(3):
            e.rollout(50, 1, 0)
        e.sync(); e.lap_times()
        for ep in range(4):
            e.reset_random(1, ep + 1)
            e.lap_begin(); e.rollout(50, 1, 0); e.lap_end()
        e.sync()
        res[name].append(round(sum(e.lap_times()) * 1e3 / 200, 1))
        e.close()
    print(rep, {k: v[-1] for k, v in res.items()}, flush=True)
for name in res:
    print(f"{name:9s} us per step: {res[name]}   median {statistics.median(res[name]):.1f}   create ms median {statistics.median(cost[name]):.1f}")
