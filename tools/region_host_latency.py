#!/usr/bin/env python3
"""Where a timed region's wall clock goes beyond its device time (1 048 576 arms, 20-step regions as in the driver's
command): host timestamps around the calls of one region, medians over many regions.
    python tools/region_host_latency.py"""
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
rows = {"enqueue_rollout": [], "rollout_sync_total": [], "device_rollout": [], "with_end_total": [], "enqueue_end": [],
        "torch_sync_after_sync": []}
for rep in range(200):
    e.sync(); torch.cuda.synchronize()
    e.lap_times()
    t0 = time.perf_counter()
    e.lap_begin(); e.rollout(T, 1, 0); e.lap_end()
    t1 = time.perf_counter()
    e.sync()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    rows["enqueue_rollout"].append((t1 - t0) * 1e6)
    rows["rollout_sync_total"].append((t2 - t0) * 1e6)
    rows["torch_sync_after_sync"].append((t3 - t2) * 1e6)
    rows["device_rollout"].append(sum(e.lap_times()) * 1e3)
    # the same with the episode end behind it
    t0 = time.perf_counter()
    e.rollout(T, 1, 0)
    ta = time.perf_counter()
    e.gather_wait(); buf = e.gather_begin(buf); e.reset_random(1, rep + 1)
    tb = time.perf_counter()
    e.sync(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    rows["with_end_total"].append((t2 - t0) * 1e6)
    rows["enqueue_end"].append((tb - ta) * 1e6)
for name, v in rows.items():
    print(f"{name:24s} median {statistics.median(v):8.1f} us   min {min(v):8.1f}")
