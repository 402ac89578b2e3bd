#!/usr/bin/env python3
"""The three CPU baselines of SURVEY.md 8(d), timed with the numpy port (oracle/) on this box's host cores.
Reported next to the GPU figures in DESIGN.md; bench.py's `cpu_baseline` object is mode (ii).

  (i)   scalar-faithful: per-env Python loop with the reference's 3-FK-chains-per-sub-step structure, 1 core
  (ii)  vectorised numpy fp64 over N envs, 1 process
  (iii) (ii) fanned out over all host cores with multiprocessing
"""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import manytor_oracle as mo  # noqa: E402
from oracle import philox_ref as px  # noqa: E402


def scalar_faithful(n_envs=16, steps=20, k=7):
    envs = [mo.ScalarEnv(k) for _ in range(n_envs)]
    pts = px.sample_targets(1, np.arange(n_envs, dtype=np.uint64), 0, k, 51.3)
    for e, p in zip(envs, pts):
        e.reset(points=p)
    acts = [px.sample_actions(1, np.arange(n_envs, dtype=np.uint64), t, 4) for t in range(steps)]
    t0 = time.perf_counter()
    for t in range(steps):
        for e, a in zip(envs, acts[t]):
            e.step(a)
    dt = time.perf_counter() - t0
    return n_envs * steps / dt


def _vector_worker(args):
    n, steps, k, seed = args
    ids = np.arange(n, dtype=np.uint64)
    ora = mo.BatchOracle(n, k)
    ora.reset(px.sample_targets(seed, ids, 0, k, 51.3).astype(np.float64))
    acts = [px.sample_actions(seed, ids, t, 4).astype(np.float64) for t in range(steps + 1)]
    ora.step(acts[0])
    t0 = time.perf_counter()
    for t in range(steps):
        ora.step(acts[t + 1])
    return time.perf_counter() - t0


def vectorised(n=65536, steps=6, k=7):
    return n * steps / _vector_worker((n, steps, k, 1))


def vectorised_mp(procs, n_per=16384, steps=6, k=7):
    with mp.get_context("fork").Pool(procs) as pool:
        t0 = time.perf_counter()
        pool.map(_vector_worker, [(n_per, steps, k, s) for s in range(procs)])
        wall = time.perf_counter() - t0
    return procs * n_per * steps / wall


if __name__ == "__main__":
    cores = os.cpu_count()
    use = min(cores, int(os.environ.get("MT_CPU_PROCS", "64")))
    out = {
        "host_cores": cores,
        "scalar_faithful_env_steps_per_s_1core": scalar_faithful(),
        "vectorised_numpy_env_steps_per_s_1proc": vectorised(),
        f"vectorised_numpy_env_steps_per_s_{use}procs": vectorised_mp(use),
    }
    print(json.dumps(out, indent=1))
