#!/usr/bin/env python3
"""Does splitting ONE batch over several HIP streams hide the kernel boundary of the launch-per-step path?

A step of env i depends only on env i's previous step, so a batch cut into C contiguous chunks can run as C independent
chains of launches on C streams: while one chunk's kernel drains and the next one is dispatched, the other chunks'
kernels keep the chip busy.  This probe times it with C separate engines (each has its own non-blocking stream) against
one engine owning the whole batch: wall clock over `episodes` x 50 steps incl. the per-episode reset, host-synchronised
only at the end.
    python tools/multistream_probe.py > gpurun_out/multistream_probe.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import manytor_amd as m  # noqa: E402


def run(n_total, chunks, episodes=40, fused=False):
    n = n_total // chunks
    engs = [m.StepEngine(n, 7, env_id_base=c * n) for c in range(chunks)]

    def episode(ep):
        for e in engs:
            e.reset_random(1, ep)
        if fused:
            for e in engs:
                e.rollout_fused(50, 1, 0)
        else:
            for e in engs:
                e.rollout(50, 1, 0)
    t0 = time.perf_counter()
    ep = 0
    while time.perf_counter() - t0 < 0.2:
        episode(ep)
        ep += 1
        for e in engs:
            e.sync()
    t0 = time.perf_counter()
    for r in range(episodes):
        episode(r)
    for e in engs:
        e.sync()
    dt = time.perf_counter() - t0
    for e in engs:
        e.close()
    return round(dt / (episodes * 50) * 1e6, 3)


res = {}
for n_total in (65536, 131072, 262144, 524288, 1048576):
    row = {}
    for chunks in (1, 2, 4, 8):
        if n_total // chunks < 16384:
            continue
        row[f"{chunks} stream(s)"] = [run(n_total, chunks), run(n_total, chunks)]
    res[n_total] = row
    print(n_total, row, file=sys.stderr, flush=True)
print(json.dumps(res, indent=1))
