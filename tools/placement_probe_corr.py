#!/usr/bin/env python3
"""Does a plain copy between rows of the arena see the fast / slow placement of a 4 194 304-arm engine?  Successive fresh
engines of one process: GB/s of torch copies obs rows <- points rows (21 rows each, read + write) next to us per step.
    python tools/placement_probe_corr.py [engines]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import manytor_amd as m  # noqa: E402

L = m.lib
n = 4194304
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    e = m.StepEngine(n, 7)
    src, dst = e.device_tensor(L.F_POINTS), e.device_tensor(L.F_OBS)
    e.sync()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for _ in range(2):
        dst.copy_(src)
    ev[0].record()
    for _ in range(5):
        dst.copy_(src)
    ev[1].record()
    torch.cuda.synchronize()
    gbs = 5 * 2 * 21 * n * 4 / (ev[0].elapsed_time(ev[1]) * 1e-3) / 1e9
    e.reset_random(1, 0)
    for _ in range(3):
        e.rollout(50, 1, 0)
    e.sync(); e.lap_times()
    for ep in range(4):
        e.reset_random(1, ep + 1)
        e.lap_begin(); e.rollout(50, 1, 0); e.lap_end()
    e.sync()
    us = sum(e.lap_times()) * 1e3 / 200
    print(f"engine {rep}: row copy {gbs:7.0f} GB/s   step {us:6.1f} us", flush=True)
    del src, dst
    e.close()
