#!/usr/bin/env python3
"""How long does the GPU take to reach its steady rate?  us per step of consecutive 50-step segments (reset untimed between
them) from a cold start, for 1 048 576 and 4 194 304 arms: the curve behind bench.py's pre-warm time.
    python tools/warmup_curve.py > gpurun_out/warmup_curve.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import manytor_amd as m  # noqa: E402

out = {}
for n in (4194304, 1048576):
    time.sleep(2.0)                                   # let the device go idle
    e = m.StepEngine(n, 7)
    e.reset_random(1, 0)
    e.sync()
    t0 = time.perf_counter()
    pts = []
    while time.perf_counter() - t0 < 3.0:
        e.timer_start()
        e.rollout(50, 1, 0)
        us = e.timer_stop() * 1e3 / 50
        pts.append((round(time.perf_counter() - t0, 3), round(us, 2)))
        e.reset_random(1, len(pts))
    e.close()
    # thin the curve: first 10 points, then every ~0.25 s
    keep, last = pts[:10], pts[9][0] if len(pts) > 9 else 0.0
    for t, us in pts[10:]:
        if t - last >= 0.25:
            keep.append((t, us))
            last = t
    out[n] = {"seconds_us_per_step": keep, "first": pts[0][1], "min": min(p[1] for p in pts), "last": pts[-1][1]}
    print(n, out[n]["first"], out[n]["min"], out[n]["last"], file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
