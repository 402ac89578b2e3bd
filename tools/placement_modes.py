#!/usr/bin/env python3
"""Same binary, same process, same virtual addresses -- two speeds (profiles/r02_4m_spread.md): allocate / step /
free a 4 194 304-arm engine several times.  Run plainly it prints HIP-event times per allocation; run under
`rocprofv3 --pmc ...` the per-dispatch counter rows can be split by allocation (dispatch order = allocation order)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import manytor_amd as m  # noqa: E402

N = int(os.environ.get("MT_PLACE_N", 4194304))
ALLOCS = int(os.environ.get("MT_PLACE_ALLOCS", 6))
out = []
for k in range(ALLOCS):
    e = m.StepEngine(N, 7)
    e.reset_random(1, 0)
    e.rollout(30, 1, 0)          # warm
    e.sync()
    e.timer_start()
    e.rollout(20, 1, 30)         # 20 timed launches per allocation
    out.append(round(e.timer_stop() * 1e3 / 20, 2))
    e.close()
print(json.dumps({"n": N, "us_per_step_by_allocation": out}))
