#!/usr/bin/env python3
"""Long-run stability check: 1 048 576 arms, 100 000 fused steps with on-device auto-reset (K = 2, wide pickup box so
episodes end often), then 20 000 launch-per-step steps with a reset_done after each.  Looks for anything non-finite,
out-of-range or stuck."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import manytor_amd as m  # noqa: E402

n = 1 << 20
eng = m.StepEngine(n, 2, pickup_tol=25.0)
eng.reset_random(99, 0)
t0 = time.perf_counter()
for chunk in range(100):
    eng.rollout_fused(1000, 99, chunk * 1000, auto_reset=True)
eng.sync()
dt = time.perf_counter() - t0
ep, last, tot, pts = eng.episodes(), eng.last_return(), eng.total_reward(), eng.points()
assert np.isfinite(last).all() and np.isfinite(tot).all() and np.isfinite(pts).all() and np.isfinite(eng.obs()).all()
assert np.all(last == np.round(last)) and np.all(tot == np.round(tot))
assert (pts[..., 2] >= 0).all() and (np.linalg.norm(pts.astype(np.float64), axis=-1) <= 51.3 * (1 + 1e-6)).all()
g = eng.goals()
assert g.min() >= -180 and g.max() <= 179
print(f"fused + auto-reset: 100000 steps x {n} arms in {dt:.1f} s ({n * 1e5 / dt:.3e} env-steps/s); episodes per env "
      f"min/mean/max = {ep.min()}/{ep.mean():.1f}/{ep.max()}; mean finished return {last[ep > 0].mean():.2f}")
t0 = time.perf_counter()
for t in range(20000):
    eng.step_random(99, 100000 + t)
    eng.reset_done(99)
eng.sync()
dt = time.perf_counter() - t0
ep2 = eng.episodes()
assert (ep2 >= ep).all() and np.isfinite(eng.obs()).all() and np.isfinite(eng.total_reward()).all()
print(f"step + reset_done: 20000 steps in {dt:.1f} s ({n * 2e4 / dt:.3e} env-steps/s); episodes grew by {(ep2 - ep).mean():.2f} per env")
