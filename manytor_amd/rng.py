"""Host RNG streams for parity mode.

The reference draws targets (manytor.py:231) and actions (manytor.py:216) one
scalar at a time from numpy's *global* RandomState, env by env.  These helpers
consume the same global stream in the same order but in bulk, so that after
``np.random.seed(s)`` a ``Multienv`` of this package sees bit-identical targets
and actions to the reference's (fixture F6, tests/golden/f6_rng_streams.npz).
No arithmetic of the step path happens here -- only input generation.
"""
from __future__ import annotations

import numpy as np


def draw_targets(n_envs: int, obj_number: int, radius: float = 51.3) -> np.ndarray:
    """(n_envs, K, 3) float64 targets; advances np.random exactly as n_envs reference resets do."""
    need = n_envs * obj_number
    state = np.random.get_state()
    m = int(need / 0.26 * 1.15) + 64          # acceptance is pi/12 (z >= 0 and inside the sphere)
    while True:
        np.random.set_state(state)
        cand = np.random.uniform(-radius, radius, size=(m, 3))
        ok = (cand[:, 2] >= 0) & (np.sqrt(np.sqrt(cand[:, 0] ** 2 + cand[:, 1] ** 2) ** 2 + cand[:, 2] ** 2) <= radius)
        idx = np.flatnonzero(ok)
        if idx.size >= need:
            break
        m *= 2
    last = int(idx[need - 1])
    np.random.set_state(state)                 # rewind, then consume exactly the candidates the reference would
    np.random.uniform(-radius, radius, size=(last + 1, 3))
    return cand[idx[:need]].reshape(n_envs, obj_number, 3)


def draw_actions(n_envs: int, dof: int = 4) -> np.ndarray:
    """(n_envs, dof) int64 degrees in [-180, 180): the reference's action_sample for every env in order."""
    return np.random.randint(low=-180, high=180, size=(n_envs, dof))
