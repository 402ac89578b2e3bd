#!/usr/bin/env python3
"""Is a 3-5 x slow measurement window a property of the ARENA (placement) or of the MOMENT (clock / power events)?
One engine, the same 50-step episode over and over for `seconds`; the device time of every episode (HIP-event laps, no
host wait in between chunks of 20 episodes) is printed as a timeline of chunk medians and the slow episodes are listed
with their time stamps.  Then the same on a SECOND engine created while the first still exists, alternating between the
two: a slow arena stays slow in every window, a slow moment hits both.
    python tools/throttle_timeline.py [n_envs] [seconds] [dh7]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import manytor_amd as m  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
kw = dict(dh_table=m.DH7_TABLE, radius=92.6) if "dh7" in sys.argv else {}
engines = [m.StepEngine(n, 7, **kw), m.StepEngine(n, 7, **kw)]
for e in engines:
    e.reset_random(1, 0)
t0 = time.perf_counter()
rows = []
ep = 0
while time.perf_counter() - t0 < seconds:
    for which, e in enumerate(engines):
        e.sync()
        e.lap_times()
        for _ in range(20):
            e.lap_begin()
            e.rollout(50, 1, 0)
            e.lap_end()
            ep += 1
            e.reset_random(1, ep)
        laps = np.array(e.lap_times()) * 1e3 / 50
        rows.append((time.perf_counter() - t0, which, float(np.median(laps)), float(laps.max())))
base = np.median([r[2] for r in rows])
print(f"{n} envs{' dh7' if kw else ''}: {len(rows)} chunks of 20 episodes over {seconds:.0f} s, median {base:.2f} us per step")
slow = [r for r in rows if r[3] > 1.3 * base]
print(f"chunks with an episode > 1.3 x median: {len(slow)}")
for t, which, med, mx in slow[:60]:
    print(f"  t = {t:7.3f} s  engine {which}  chunk median {med:7.2f}  worst episode {mx:7.2f} us per step")
per_engine = [np.median([r[2] for r in rows if r[1] == w]) for w in (0, 1)]
print("median per engine:", [round(float(v), 2) for v in per_engine])
