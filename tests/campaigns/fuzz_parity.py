#!/usr/bin/env python3
"""Randomised parity campaign: many random configurations (joint count, DH table, targets, sub-steps, batch size,
integer / fractional / huge actions, kernel variants) stepped in lock step with the CPU oracle, using the same
comparator and tolerances as tests/test_gpu_parity.py.  Run on the GPU box for as long as you like:

    python tests/campaigns/fuzz_parity.py --minutes 5 --seed 1
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import manytor_amd as m  # noqa: E402
from oracle import manytor_oracle as mo  # noqa: E402
from tests.test_gpu_parity import Lockstep  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--minutes", type=float, default=3.0)
ap.add_argument("--seed", type=int, default=0)
args = ap.parse_args()
rng = np.random.RandomState(args.seed)
deadline = time.time() + 60 * args.minutes
last_note = time.time()
cases = guarded = compared = 0
while time.time() < deadline:
    dof = int(rng.randint(2, 9))
    kind = rng.randint(3)
    if kind == 0 and dof == 4:
        table, radius = np.array(m.REF_DH_TABLE), 51.3
    elif kind == 1 and dof == 7:
        table, radius = np.array(m.DH7_TABLE), 92.6
    else:
        table = np.column_stack([rng.uniform(-10, 10, dof) * (rng.rand(dof) < 0.5),
                                 rng.choice([-np.pi / 2, 0.0, np.pi / 2, 0.37, -1.2], dof),
                                 rng.uniform(0, 20, dof) * (rng.rand(dof) < 0.7),
                                 rng.choice([0.0, -np.pi / 2, np.pi / 2, 0.2], dof)])
        radius = float(np.abs(table[:, 0]).sum() + np.abs(table[:, 2]).sum() + 1.0)
    k = int(rng.choice([1, 2, 3, 7, 10, 17, 32]))
    substeps = int(rng.choice([2, 3, 5, 9, 10, 25, 40]))
    n = int(rng.choice([1, 5, 64, 65, 300, 1024, 3000]))
    tol = float(rng.choice([0.5, 3.0, 8.0, 20.0]))
    variant = [dict(), dict(specialize=False), dict(hw_trig=True), dict(direct_trig=True), dict(dh_in_lds=True)][rng.randint(5)]
    ls = Lockstep(m, mo, n, k, table=table, substeps=substeps, pickup_tol=tol, radius=radius, **variant)
    ls.ora.pickup_tol = tol
    pts = rng.uniform(-radius, radius, size=(n, k, 3)) * rng.uniform(0.2, 1.0)
    pts[..., 2] = np.abs(pts[..., 2])
    ls.reset(pts)
    mode = rng.randint(4)
    for t in range(int(rng.randint(2, 9))):
        if mode == 0:
            a = rng.randint(-180, 180, size=(n, dof)).astype(np.float64)
        elif mode == 1:
            a = rng.uniform(-180, 180, size=(n, dof)).astype(np.float32).astype(np.float64)
        elif mode == 2:
            a = (ls.ora.goals + rng.uniform(-3, 3, size=(n, dof))).astype(np.float32).astype(np.float64)   # small moves
        else:
            a = rng.randint(-720, 720, size=(n, dof)).astype(np.float64)                                   # far outside [-180,180)
        ls.step(a)
    cases += 1
    if time.time() - last_note > 45:                 # a long silent run looks hung to the GPU box's watchdog
        print(f"... {cases} configurations so far", flush=True)
        if os.path.isdir(os.path.join(ROOT, "gpurun_out")):   # a sign of life that survives a `| tail` behind this script
            with open(os.path.join(ROOT, "gpurun_out", ".campaign_progress"), "a") as pf:
                pf.write(f"{os.path.basename(__file__)} {cases}\n")
        last_note = time.time()
    guarded += ls.guarded
    compared += ls.compared
    ls.eng.close()
print(f"fuzz ok: {cases} random configurations, {compared} env-steps compared exactly on reward/done/alive, "
      f"{guarded} inside the guard band ({guarded / max(1, guarded + compared):.2%}), seed {args.seed}")
