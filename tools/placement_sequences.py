#!/usr/bin/env python3
"""Eight successive fresh 4 194 304-arm engines of one process, us per step each, created plainly (MT_PLACEMENT_PROBE=0) or
placed by trial (=2: mt_create reports its timings on stderr).   python tools/placement_sequences.py 0|2"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import manytor_amd as m  # noqa: E402

os.environ["MT_PLACEMENT_PROBE"] = sys.argv[1]
out = []
for rep in range(8):
    e = m.StepEngine(4194304, 7)
    e.reset_random(1, 0)
    for _ in range(3):
        e.rollout(50, 1, 0)
    e.sync(); e.lap_times()
    for ep in range(4):
        e.reset_random(1, ep + 1)
        e.lap_begin(); e.rollout(50, 1, 0); e.lap_end()
    e.sync()
    out.append(round(sum(e.lap_times()) * 1e3 / 200, 1))
    e.close()
print("MT_PLACEMENT_PROBE=" + sys.argv[1], out)
