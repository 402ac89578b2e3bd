#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer boundary: actions come from host numpy arrays and observations /
rewards / done flags are copied back to host arrays every step (what a drop-in user of `Multienv.step(list)` pays).
Never the headline number (bench.py keeps everything resident); quoted in DESIGN.md section 5."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import manytor_amd as m  # noqa: E402

for n in (65536, 1048576):
    eng = m.StepEngine(n, 7)
    eng.reset_random(1, 0)
    acts = [np.random.randint(-180, 180, size=(n, 4)).astype(np.float32) for _ in range(4)]
    eng.step(acts[0]); eng.obs()
    t0 = time.perf_counter()
    steps = 12
    for t in range(steps):
        eng.step(acts[t % 4])          # H2D 16 B/env + transpose
        o, r, d = eng.obs(), eng.reward(), eng.get(m.lib.F_DONE)   # D2H 84 + 4 + 1 B/env
    dt = time.perf_counter() - t0
    print(f"N={n}: {n * steps / dt:.3e} env-steps/s with host actions in and obs/reward/done out every step "
          f"({dt / steps * 1e3:.2f} ms per step, {105 * n * steps / dt / 1e9:.1f} GB/s over PCIe)")
