#!/usr/bin/env python3
"""How often does the angle-addition recurrence flip a ground-flag decision relative to evaluating every sub-step
with the polynomial sincos?  Both engines see identical states (re-synchronised every step), so every mismatch is a
pose whose minimum z lies within the recurrence's drift of zero."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import manytor_amd as m  # noqa: E402

n, k, steps = 1 << 20, 7, 20
for table, name, radius in ((m.REF_DH_TABLE, "reference 4-DoF", 51.3), (m.DH7_TABLE, "7-DoF", 92.6)):
    a = m.StepEngine(n, k, dh_table=table, radius=radius)
    b = m.StepEngine(n, k, dh_table=table, radius=radius, direct_trig=True)
    a.reset_random(5, 0)
    b.reset_random(5, 0)
    flips = 0
    worst_ee = 0.0
    for t in range(steps):
        a.step_random(5, t)
        b.step_random(5, t)
        ra, rb = a.reward(), b.reward()
        flips += int(((ra == -1) != (rb == -1)).sum())
        worst_ee = max(worst_ee, float(np.abs(a.ee() - b.ee()).max()))
        # keep the two in lock step: copy b's discrete state into a
        a.set(m.lib.F_ALIVE, b.get(m.lib.F_ALIVE))
        a.set(m.lib.F_TOTAL_REWARD, b.total_reward())
    print(f"{name}: {flips} ground-flag flips in {n * steps} env-steps ({flips / (n * steps):.2e}); "
          f"final-pose EE difference {worst_ee:.1e} (same code path: expected 0)")
