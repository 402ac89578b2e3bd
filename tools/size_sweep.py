#!/usr/bin/env python3
"""The library's OWN dispatch (no overrides) over the batch sizes of BASELINE.json's configs and of the strong-scaling
shards: us per step of mt_rollout by HIP events, two protocols -- `segment`: a 50-step segment from an idle device,
reset untimed; `steady`: 12 x (reset + 50 steps) back to back, one pair of events around the lot -- and the fused
rollout's us per step, with the kernel mt_create picked.
    python tools/size_sweep.py [--dh7] > profiles/rNN_variant_sweep.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import manytor_amd as m  # noqa: E402

for k in ("MT_SPLIT", "MT_PREFETCH", "MT_CHAINS", "MT_GRAPH", "MT_TRIG_TABLE"):
    os.environ.pop(k, None)
kw = dict(dh_table=m.DH7_TABLE, radius=92.6) if "--dh7" in sys.argv else {}
sizes = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [8192, 32768, 65536, 131072, 196608, 262144, 524288, 1048576,
                                                                  2097152, 4194304]


def warm(e, fused=False):
    t0 = time.perf_counter()
    ep = 0
    while time.perf_counter() - t0 < 0.2:
        e.reset_random(1, ep)
        (e.rollout_fused if fused else e.rollout)(50, 1, 0)
        ep += 1
        e.sync()


res = {}
for n in sizes:
    e = m.StepEngine(n, 7, **kw)
    warm(e)
    seg = []
    for r in range(6):
        e.reset_random(1, r)
        e.sync()
        e.timer_start()
        e.rollout(50, 1, 0)
        seg.append(e.timer_stop() * 1e3 / 50)
    e.timer_start()
    for r in range(12):
        e.reset_random(1, r)
        e.rollout(50, 1, 0)
    steady = e.timer_stop() * 1e3 / 600
    warm(e, fused=True)
    fus = []
    for r in range(4):
        e.reset_random(1, r)
        e.sync()
        e.timer_start()
        e.rollout_fused(50, 1, 0)
        fus.append(e.timer_stop() * 1e3 / 50)
    res[n] = {"kernel": e.step_kernel_name(), "segment_us_per_step": round(sorted(seg)[len(seg) // 2], 3),
              "steady_us_per_step_incl_reset": round(steady, 3), "fused_us_per_step": round(sorted(fus)[len(fus) // 2], 3)}
    e.close()
    print(n, res[n], file=sys.stderr, flush=True)
print(json.dumps(res, indent=1))
