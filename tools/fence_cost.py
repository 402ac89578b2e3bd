#!/usr/bin/env python3
"""The closing fence of a timed region (1 048 576 arms, 20 steps + episode end): torch.cuda.synchronize() alone against
mt_sync first and torch.cuda.synchronize() behind it.  Host timestamps, medians of 200 regions, interleaved.
    python tools/fence_cost.py"""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import manytor_amd as m  # noqa: E402

n, k, T = 1048576, 7, 20
e = m.StepEngine(n, k)
e.reset_random(1, 0)
for _ in range(30):
    e.rollout(200, 1, 0)
    e.sync()
buf = None
rows = {"torch.cuda.synchronize() only": [], "mt_sync + torch.cuda.synchronize()": []}
ep = 0
for rep in range(200):
    for own in (False, True):
        e.sync(); torch.cuda.synchronize()
        e.lap_times()
        t0 = time.perf_counter()
        e.lap_begin(); e.rollout(T, 1, 0); e.lap_end()
        ep += 1
        e.gather_wait(); buf = e.gather_begin(buf); e.reset_random(1, ep)
        if own:
            e.sync()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        rows["mt_sync + torch.cuda.synchronize()" if own else "torch.cuda.synchronize() only"].append((t1 - t0) * 1e6)
for name, v in rows.items():
    print(f"{name:36s} median {statistics.median(v):8.1f} us   min {min(v):8.1f}")
