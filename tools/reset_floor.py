#!/usr/bin/env python3
"""Is reset_kernel store-bound?  us per mt_reset_random at 1 048 576 envs for K = 1 .. 14 targets (bytes written per env:
12 K + 4 D + 33) next to a plain device fill of the same number of bytes (torch.Tensor.fill_), HIP-event timed."""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import manytor_amd as m  # noqa: E402


def fill_us(nbytes, reps=30):
    t = torch.empty(nbytes // 4, dtype=torch.float32, device="cuda")
    for _ in range(5):
        t.fill_(1.0)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        t.fill_(2.0)
        b.record()
    torch.cuda.synchronize()
    return statistics.median(a.elapsed_time(b) for a, b in ev) * 1e3


out = {}
n = 1048576
for k in (1, 3, 7, 14):
    e = m.StepEngine(n, k)
    e.reset_random(1, 0)
    for _ in range(100):
        e.rollout(1, 1, 0)
    e.lap_times()
    for r in range(30):
        e.lap_begin()
        e.reset_random(1, r + 1)
        e.lap_end()
    t = e.lap_times()
    e.close()
    nbytes = (12 * k + 16 + 33) * n
    out[f"K={k}"] = {"reset_random_us": statistics.median(t) * 1e3, "bytes_written": nbytes,
                     "write_rate_gbs": nbytes / (statistics.median(t) * 1e-3) / 1e9, "fill_same_bytes_us": fill_us(nbytes)}
print(json.dumps(out, indent=1))
