#!/usr/bin/env python3
"""Is a small-batch step sequence bound by the host's launch rate?  Enqueue time (host returns from mt_rollout) vs
completion time of the same launches, per batch size.
    python tools/host_launch_cost.py > gpurun_out/host_launch.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import manytor_amd as m  # noqa: E402

out = {}
for n in (1024, 16384, 65536, 131072, 1048576):
    e = m.StepEngine(n, 7)
    e.reset_random(1, 0)
    for _ in range(5):
        e.rollout(200, 1, 0)
        e.sync()
    rows = []
    for _ in range(5):
        e.reset_random(1, 0)
        e.sync()
        t0 = time.perf_counter()
        e.rollout(50, 1, 0)
        t1 = time.perf_counter()
        e.sync()
        t2 = time.perf_counter()
        rows.append(((t1 - t0) / 50 * 1e6, (t2 - t0) / 50 * 1e6))
    rows.sort()
    out[n] = {"enqueue_us_per_launch": round(rows[2][0], 2), "complete_us_per_launch": round(rows[2][1], 2)}
    e.close()
print(json.dumps(out, indent=1))
