"""ctypes front end of oracle/manytor_oracle.c -- TEST INFRASTRUCTURE ONLY.

``COracle`` has the interface of ``manytor_oracle.BatchOracle`` (reset / step / get_observations, state arrays with
a leading env axis, decision margins) but runs the C restatement with OpenMP, so full-size batches (1 M arms) can be
checked env by env in about a second.  The shared object is built on demand with gcc into oracle/_build/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "manytor_oracle.c")
LIB = os.path.join(HERE, "_build", "libmanytor_oracle.so")

_lib = None


def build(force: bool = False) -> str:
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.run(["gcc", "-O2", "-fopenmp", "-fPIC", "-shared", "-ffp-contract=off", SRC, "-o", LIB + ".tmp", "-lm"],
                       check=True)
        os.replace(LIB + ".tmp", LIB)
    return LIB


def load():
    global _lib
    if _lib is None:
        lib = C.CDLL(build())
        dp, u8p, i32p = C.POINTER(C.c_double), C.POINTER(C.c_uint8), C.POINTER(C.c_int32)
        lib.mto_step.restype = None
        lib.mto_step.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_double, dp, dp, dp, u8p, dp, dp, dp, i32p, u8p,
                                 dp, dp, dp, C.c_int, C.c_int, C.c_int, dp]
        lib.mto_observe.restype = None
        lib.mto_observe.argtypes = [C.c_int64, C.c_int, C.c_int, dp, dp, dp, u8p, dp, C.c_int, C.c_int]
        lib.mto_max_threads.restype = C.c_int
        _lib = lib
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class COracle:
    """N lock-stepped envs on the C restatement; mirrors BatchOracle (manytor.py:125-260 semantics)."""

    def __init__(self, n_envs, obj_number, table=None, substeps=25, pickup_tol=8.0, radius=51.3, threads=0,
                 obs_frame=-2, ee_frame=-1):
        from .manytor_oracle import REF_DH_TABLE
        self.lib = load()
        self.n, self.k = int(n_envs), int(obj_number)
        self.table = np.ascontiguousarray(REF_DH_TABLE if table is None else table, dtype=np.float64)
        self.dof = self.table.shape[0]
        assert 2 <= self.dof <= 8 and 1 <= self.k <= 32
        self.fo, self.fe = obs_frame % self.dof, ee_frame % self.dof     # rows of joints_coordinates (Python indexing)
        self.substeps, self.pickup_tol, self.radius = int(substeps), float(pickup_tol), float(radius)
        self.threads = int(threads) or min(self.lib.mto_max_threads(), os.cpu_count() or 1)
        self.goals = np.zeros((self.n, self.dof))
        self.points = np.zeros((self.n, self.k, 3))
        self.alive_u8 = np.ones((self.n, self.k), dtype=np.uint8)
        self.total_reward = np.zeros(self.n)
        self.joints_coordinates = np.zeros((self.n, self.dof, 3))
        self.ground_margin = np.full(self.n, np.inf)
        self.pickup_margin = np.full((self.n, self.k), np.inf)
        self.ground_hit = np.zeros(self.n, dtype=bool)
        self.zmin = np.full(self.n, np.inf)      # signed min z of the two tested frames over the last step's poses

    @property
    def alives(self):
        return self.alive_u8.view(np.bool_)

    def reset(self, points):
        self.goals[:] = 0
        self.total_reward[:] = 0
        self.alive_u8[:] = 1
        self.points = np.ascontiguousarray(np.asarray(points, dtype=np.float64).reshape(self.n, self.k, 3)).copy()
        return self.get_observations()

    def get_observations(self):
        obs = np.empty((self.n, 3 * self.k))
        self.lib.mto_observe(self.n, self.dof, self.k, _p(self.table, C.c_double), _p(self.goals, C.c_double),
                             _p(self.points, C.c_double), _p(self.alive_u8, C.c_uint8), _p(obs, C.c_double), self.threads,
                             self.fo)
        return obs

    def step(self, actions):
        act = np.ascontiguousarray(np.asarray(actions, dtype=np.float64).reshape(self.n, self.dof))
        obs2 = np.empty((self.n, 3 * self.k))
        reward = np.empty(self.n, dtype=np.int32)
        done = np.empty(self.n, dtype=np.uint8)
        self.lib.mto_step(self.n, self.dof, self.k, self.substeps, self.pickup_tol, _p(self.table, C.c_double),
                          _p(self.goals, C.c_double), _p(self.points, C.c_double), _p(self.alive_u8, C.c_uint8),
                          _p(self.total_reward, C.c_double), _p(act, C.c_double), _p(obs2, C.c_double),
                          _p(reward, C.c_int32), _p(done, C.c_uint8), _p(self.joints_coordinates, C.c_double),
                          _p(self.ground_margin, C.c_double), _p(self.pickup_margin, C.c_double), self.threads,
                          self.fo, self.fe, _p(self.zmin, C.c_double))
        self.ground_hit = reward == -1
        return obs2, reward.astype(np.int64), done.astype(bool)
