#!/usr/bin/env python3
"""mt_step with staged actions (policy-in-the-loop shape, step_kernel<SAMPLE = false>) under MT_FLAT_FROM = 0 (every launch in
the FLAT addressing form) against none, per batch size: us per step (bench.time_loaded_action_steps).
    python tools/ab_loaded_step.py [sizes ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import manytor_amd as m  # noqa: E402

sizes = [int(v) for v in sys.argv[1:]] or [262144, 524288, 1048576, 2097152]
for rep in range(2):
    for n in sizes:
        row = {}
        for ff in ("0", "1000000000"):
            os.environ["MT_FLAT_FROM"] = ff
            us, _ = bench.time_loaded_action_steps(m, n, m.REF_DH_TABLE, 51.3, 7, 0, 0x5EED, steps=400)
            row["flat" if ff == "0" else "renewed"] = round(us, 2)
        print(n, row, flush=True)
