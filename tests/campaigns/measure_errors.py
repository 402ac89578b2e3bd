#!/usr/bin/env python3
"""Measured fp32-vs-fp64 error of the HIP path against the CPU oracle on seeded random rollouts (the numbers quoted
next to the tolerances in DESIGN.md section 4)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import manytor_amd as m  # noqa: E402
from oracle import manytor_oracle as mo  # noqa: E402

for name, table, radius in (("reference 4-DoF", m.REF_DH_TABLE, 51.3), ("7-DoF", m.DH7_TABLE, 92.6)):
    n, k, steps = 131072, 7, 8
    eng = m.StepEngine(n, k, dh_table=table, radius=radius)
    ora = mo.BatchOracle(n, k, table=np.asarray(table), radius=radius)
    eng.reset_random(17, 0)
    ora.reset(eng.points().astype(np.float64))
    worst = dict(pos=0.0, dist=0.0, ang_deg=0.0, ang_as_pos=0.0)
    flips = 0
    for t in range(steps):
        eng.step_random(17, t)
        a = eng.goals().astype(np.float64)
        pre_alive = ora.alives.copy()
        obs_ref, rew_ref, _ = ora.step(a)
        jc = eng.joints_coordinates()
        worst["pos"] = max(worst["pos"], float(np.abs(jc - ora.joints_coordinates).max()))
        o = eng.obs().reshape(n, k, 3).astype(np.float64)
        r = obs_ref.reshape(n, k, 3)
        both = pre_alive & eng.alives() | pre_alive
        m_ = np.abs(ora.joints_coordinates[:, -2][:, None, :] - ora.points)
        rho = np.hypot(m_[..., 0], m_[..., 1])
        worst["dist"] = max(worst["dist"], float(np.abs(o[..., 0] - r[..., 0])[pre_alive].max()))
        e_r = np.abs(o[..., 1] - r[..., 1])[pre_alive]
        e_t = np.abs(o[..., 2] - r[..., 2])[pre_alive]
        worst["ang_deg"] = max(worst["ang_deg"], float(max(e_r.max(), e_t.max())))
        worst["ang_as_pos"] = max(worst["ang_as_pos"], float(max((e_r * rho[pre_alive]).max(), (e_t * r[..., 0][pre_alive]).max()) / 57.3))
        mism = eng.reward() != rew_ref
        flips += int(mism.sum())
        idx = np.flatnonzero(mism | (eng.alives() != ora.alives).any(axis=1))
        ora.alives[idx] = eng.alives()[idx]
        ora.total_reward[idx] = eng.total_reward()[idx]
        ora.points[idx] = eng.points()[idx].astype(np.float64)
    print(f"{name}: {n * steps} env-steps | max position error {worst['pos']:.2e} | max distance error {worst['dist']:.2e} | "
          f"max angle error {worst['ang_deg']:.2e} deg (= {worst['ang_as_pos']:.2e} length units at its lever arm) | "
          f"reward mismatches {flips}")
