#!/usr/bin/env python3
"""What the HIP-event laps around the step launches cost a timed region in wall clock (default 1 048 576 arms, 20 steps +
episode end, host timestamps around enqueue ... mt_sync + torch.cuda.synchronize, medians of 200 regions, interleaved).
    python tools/lap_cost.py [n_envs]"""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import manytor_amd as m  # noqa: E402

n, k, T = (int(sys.argv[1]) if len(sys.argv) > 1 else 1048576), 7, 20
e = m.StepEngine(n, k)
e.reset_random(1, 0)
for _ in range(30):
    e.rollout(200, 1, 0)
    e.sync()
buf = None
rows = {"no laps": [], "laps": [], "no laps, no episode end": [], "laps, no episode end": []}
ep = 0
for rep in range(200):
    for laps in (False, True):
        for end in (True, False):
            e.sync(); torch.cuda.synchronize()
            e.lap_times()
            t0 = time.perf_counter()
            if laps:
                e.lap_begin()
            e.rollout(T, 1, 0)
            if laps:
                e.lap_end()
            if end:
                ep += 1
                e.gather_wait(); buf = e.gather_begin(buf); e.reset_random(1, ep)
            e.sync(); torch.cuda.synchronize()
            t1 = time.perf_counter()
            rows[("laps" if laps else "no laps") + ("" if end else ", no episode end")].append((t1 - t0) * 1e6)
for name, v in rows.items():
    print(f"{name:28s} median {statistics.median(v):8.1f} us   min {min(v):8.1f}")
