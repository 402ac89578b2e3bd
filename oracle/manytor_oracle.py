"""CPU oracle for the ManyTor hot path -- TEST INFRASTRUCTURE ONLY.

This module restates, in numpy/fp64, the arithmetic of the reference's
``Environment.reset/step/get_observations/is_done`` and ``dh/fk/r_theta``
(/root/reference/manytor.py).  It exists to *check* the HIP path; it is never
the thing shipped or measured.  Only ``tests/``, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py`` may import it.  The product package
(``manytor_amd``) must not import anything from ``oracle/``.

Parity status: PINNED.  Every function below is checked in
``tests/test_oracle_golden.py`` against fixtures F1..F8 under ``tests/golden/``
which were produced by importing the reference itself in the build container
(``oracle/gen_golden.py``, committed next to this file).

Two restatements live here:

* ``ScalarEnv``  -- one env, Python loops, same 3-FK-chains-per-sub-step
  structure as the reference (manytor.py:183-192).  Used for small cases and as
  the "scalar-faithful" CPU baseline.
* ``BatchOracle`` -- the same semantics vectorised over N envs and generalised
  to any DH table (D joints).  Also returns decision *margins* so parity tests
  can guard discrete outputs (reward / alive / done) near the z=0 and
  |delta|=tol thresholds, where fp32 and fp64 may legitimately disagree.
"""
from __future__ import annotations

import math

import numpy as np

# DH rows (a, alpha, d, theta_offset_rad) of the reference arm,
# manytor.py:42-48 (theta_i = radians(goal_i) + theta_offset).
REF_DH_TABLE = np.array(
    [
        [0.0, -np.pi / 2, 4.3, 0.0],
        [0.0, np.pi / 2, 0.0, 0.0],
        [0.0, -np.pi / 2, 24.3, 0.0],
        [27.0, np.pi / 2, 0.0, -np.pi / 2],
    ],
    dtype=np.float64,
)

REF_SUBSTEPS = 25       # manytor.py:178
REF_PICKUP_TOL = 8.0    # manytor.py:162
REF_RADIUS = 51.3       # manytor.py:231,236


# --------------------------------------------------------------------------
# L1 kinematics (manytor.py:17-53)
# --------------------------------------------------------------------------
def dh_matrix(a, alpha, d, theta):
    """Homogeneous transform of one DH row, manytor.py:25-32
    (Rz(theta) Tz(d) Tx(a) Rx(alpha))."""
    ct, st = np.cos(theta), np.sin(theta)
    ca, sa = np.cos(alpha), np.sin(alpha)
    return np.array(
        [
            [ct, -st * ca, st * sa, a * ct],
            [st, ct * ca, -ct * sa, a * st],
            [0.0, sa, ca, d],
            [0.0, 0.0, 0.0, 1.0],
        ],
        dtype=np.float64,
    )


def fk_matrix(mode, goals_deg, table=REF_DH_TABLE):
    """Product of the first ``mode`` DH transforms, manytor.py:35-53.
    ``goals_deg`` are joint angles in degrees (manytor.py:39)."""
    m = np.eye(4)
    for j in range(mode):
        a, alpha, d, off = table[j]
        m = m.dot(dh_matrix(a, alpha, d, math.radians(goals_deg[j]) + off))
    return m


def r_theta(v1, v2):
    """Bearing / elevation (degrees) of |v1-v2|, manytor.py:17-22."""
    dx, dy, dz = (abs(v1[i] - v2[i]) for i in range(3))
    h = math.sqrt(dx * dx + dy * dy)
    return math.degrees(math.atan2(dx, dy)), math.degrees(math.atan2(h, dz))


def joints_coordinates(goals_deg, table=REF_DH_TABLE):
    """The reference's ``joints_coordinates`` array, manytor.py:188-189:
    row 0 = zeros, row j (j>=1) = translation of fk(mode=j+1)."""
    dof = table.shape[0]
    rows = [np.zeros(3)]
    for mode in range(2, dof + 1):
        rows.append(fk_matrix(mode, goals_deg, table)[0:3, 3])
    return np.array(rows)


# --------------------------------------------------------------------------
# Scalar-faithful single env (manytor.py:125-260)
# --------------------------------------------------------------------------
class ScalarEnv:
    """One environment, restated loop by loop (no rendering, no trajectory)."""

    def __init__(self, obj_number=10, table=REF_DH_TABLE, substeps=REF_SUBSTEPS,
                 pickup_tol=REF_PICKUP_TOL, radius=REF_RADIUS, obs_frame=-2, ee_frame=-1):
        self.table = np.asarray(table, dtype=np.float64)
        self.dof = self.table.shape[0]
        # rows of joints_coordinates the reference hard-codes: observation from [2] = [-2] (manytor.py:143), pickup
        # from [3] = [-1] (:162), ground test on both (:191); selectable for other arms
        self.obs_frame, self.ee_frame = obs_frame, ee_frame
        self.obj_number = obj_number
        self.substeps = substeps
        self.pickup_tol = pickup_tol
        self.radius = radius
        self.goals = np.zeros(self.dof)
        self.alives = np.ones(obj_number, dtype=bool)
        self.points = np.zeros((obj_number, 3))
        self.total_reward = 0.0
        self.joints_coordinates = joints_coordinates(self.goals, self.table)

    # manytor.py:219-253
    def reset(self, points=None, returnable=False):
        self.goals = np.zeros(self.dof)
        self.total_reward = 0.0
        self.alives = np.ones(self.obj_number, dtype=bool)
        self.joints_coordinates = joints_coordinates(self.goals, self.table)
        if points is None:
            pts = []
            while len(pts) < self.obj_number:          # manytor.py:229-239
                c = [np.random.uniform(-self.radius, self.radius) for _ in range(3)]
                if c[2] >= 0:
                    if math.sqrt(math.sqrt(c[0] ** 2 + c[1] ** 2) ** 2 + c[2] ** 2) <= self.radius:
                        pts.append(c)
            points = np.array(pts, dtype=np.float64).reshape(self.obj_number, 3)
        self.points = np.array(points, dtype=np.float64).reshape(self.obj_number, 3).copy()
        if returnable:
            return self.get_observations()

    # manytor.py:141-153 ; measured from joints_coordinates[-2] (the "elbow")
    def get_observations(self):
        jc = self.joints_coordinates[self.obs_frame]
        obs = []
        for p in range(self.obj_number):
            if not self.alives[p]:
                obs += [0.0, 0.0, 0.0]
                self.points[p, :] = 0.0                # manytor.py:148
            else:
                m = [abs(jc[i] - self.points[p, i]) for i in range(3)]
                dist = math.sqrt(math.sqrt(m[0] ** 2 + m[1] ** 2) ** 2 + m[2] ** 2)
                r, th = r_theta(jc, self.points[p])
                obs += [dist, r, th]
        return np.array(obs)

    # manytor.py:155-173 ; measured from joints_coordinates[-1] (end effector)
    def is_done(self):
        ee = self.joints_coordinates[self.ee_frame]
        for p in range(self.obj_number):
            if all(math.isclose(ee[a], self.points[p, a], abs_tol=self.pickup_tol) for a in range(3)):
                self.alives[p] = False
        return not self.alives.any()

    # manytor.py:215-217
    def action_sample(self):
        return [np.random.randint(low=-180, high=180, size=1)[0] for _ in range(self.dof)]

    # manytor.py:175-213
    def action(self, action):
        ground = False
        initial = self.alives.copy()
        route = np.linspace(self.goals, np.asarray(action, dtype=np.float64), num=self.substeps)
        for k in range(self.substeps):
            self.goals = route[k, :]
            self.joints_coordinates = joints_coordinates(self.goals, self.table)
            if self.joints_coordinates[self.obs_frame, 2] < 0 or self.joints_coordinates[self.ee_frame, 2] < 0:
                ground = True
        obs2 = self.get_observations()
        reward = 0
        self.is_done()
        if initial.sum() > self.alives.sum():
            reward = 1
        if ground:
            reward = -1
        return reward, obs2

    # manytor.py:255-260
    def step(self, action):
        self.get_observations()
        reward, obs2 = self.action(action)
        self.total_reward += reward
        done = self.is_done()
        return obs2, reward, done


# --------------------------------------------------------------------------
# Vectorised, generalised batch oracle
# --------------------------------------------------------------------------
def _chain_positions(angles_deg, table, dtype=np.float64):
    """Origins of frames after 1..D joints for a batch of poses.

    angles_deg: (N, D).  Returns (N, D, 3): [:, j] = translation of
    fk(mode=j+1) (manytor.py:35-53), computed as a running rotation/position
    pair rather than 4x4 products."""
    angles_deg = np.asarray(angles_deg, dtype=dtype)
    n, dof = angles_deg.shape
    rot = np.broadcast_to(np.eye(3, dtype=dtype), (n, 3, 3)).copy()
    pos = np.zeros((n, 3), dtype=dtype)
    out = np.empty((n, dof, 3), dtype=dtype)
    for j in range(dof):
        a, alpha, d, off = (dtype(v) for v in table[j])
        th = np.radians(angles_deg[:, j]).astype(dtype) + off
        ct, st = np.cos(th), np.sin(th)
        ca, sa = np.cos(alpha), np.sin(alpha)
        # p += R * (a ct, a st, d)
        local = np.stack([a * ct, a * st, np.full(n, d, dtype=dtype)], axis=1)
        pos = pos + np.einsum("nij,nj->ni", rot, local)
        m = np.zeros((n, 3, 3), dtype=dtype)
        m[:, 0, 0] = ct
        m[:, 0, 1] = -st * ca
        m[:, 0, 2] = st * sa
        m[:, 1, 0] = st
        m[:, 1, 1] = ct * ca
        m[:, 1, 2] = -ct * sa
        m[:, 2, 1] = sa
        m[:, 2, 2] = ca
        rot = np.einsum("nij,njk->nik", rot, m)
        out[:, j] = pos
    return out


def batch_joints_coordinates(angles_deg, table=REF_DH_TABLE, dtype=np.float64):
    """(N, D, 3) ``joints_coordinates`` for a batch: row 0 zeros, row j = frame
    after j+1 joints (manytor.py:188-189)."""
    pos = _chain_positions(angles_deg, np.asarray(table), dtype)
    jc = pos.copy()
    jc[:, 0] = 0.0
    return jc


def observe(elbow, points, alive):
    """Vectorised get_observations (manytor.py:141-153, :17-22).
    elbow (N,3), points (N,K,3), alive (N,K) -> obs (N,3K)."""
    m = np.abs(elbow[:, None, :] - points)
    h = np.sqrt(m[..., 0] ** 2 + m[..., 1] ** 2)
    dist = np.sqrt(h ** 2 + m[..., 2] ** 2)
    r = np.degrees(np.arctan2(m[..., 0], m[..., 1]))
    th = np.degrees(np.arctan2(h, m[..., 2]))
    obs = np.stack([dist, r, th], axis=-1)
    obs = np.where(alive[..., None], obs, 0.0)
    return obs.reshape(obs.shape[0], -1)


class BatchOracle:
    """N lock-stepped envs, reference semantics, any DH table.

    State mirrors the reference attributes (manytor.py:131-139) with a leading
    env axis: goals (N,D), points (N,K,3), alives (N,K), total_reward (N,),
    joints_coordinates (N,D,3)."""

    def __init__(self, n_envs, obj_number, table=REF_DH_TABLE, substeps=REF_SUBSTEPS,
                 pickup_tol=REF_PICKUP_TOL, radius=REF_RADIUS, dtype=np.float64, obs_frame=-2, ee_frame=-1):
        self.obs_frame, self.ee_frame = obs_frame, ee_frame    # rows of joints_coordinates, see ScalarEnv
        self.n = int(n_envs)
        self.k = int(obj_number)
        self.table = np.asarray(table, dtype=np.float64)
        self.dof = self.table.shape[0]
        self.substeps = int(substeps)
        self.pickup_tol = float(pickup_tol)
        self.radius = float(radius)
        self.dtype = dtype
        self.goals = np.zeros((self.n, self.dof), dtype=dtype)
        self.points = np.zeros((self.n, self.k, 3), dtype=dtype)
        self.alives = np.ones((self.n, self.k), dtype=bool)
        self.total_reward = np.zeros(self.n, dtype=dtype)
        self.joints_coordinates = batch_joints_coordinates(self.goals, self.table, dtype)
        # diagnostics of the last step
        self.ground_margin = np.full(self.n, np.inf)
        self.pickup_margin = np.full((self.n, self.k), np.inf)
        self.ground_hit = np.zeros(self.n, dtype=bool)
        self.zmin = np.full(self.n, np.inf)      # signed min z of the two tested frames over the last step's poses

    def reset(self, points):
        """manytor.py:219-253 with the targets supplied by the caller."""
        self.goals[:] = 0
        self.total_reward[:] = 0
        self.alives[:] = True
        self.points = np.array(points, dtype=self.dtype).reshape(self.n, self.k, 3).copy()
        self.joints_coordinates = batch_joints_coordinates(self.goals, self.table, self.dtype)
        return self.get_observations()

    def get_observations(self):
        dead = ~self.alives
        self.points[dead] = 0.0                         # manytor.py:148
        return observe(self.joints_coordinates[:, self.obs_frame], self.points, self.alives)

    def is_done(self):
        """manytor.py:155-173.  Also records the distance of every axis test
        from its threshold in ``pickup_margin``."""
        ee = self.joints_coordinates[:, self.ee_frame]
        delta = np.abs(ee[:, None, :] - self.points)   # (N,K,3)
        hit = np.all(delta <= self.pickup_tol, axis=-1)
        self.pickup_margin = np.min(np.abs(delta - self.pickup_tol), axis=-1)
        self.alives &= ~hit
        return ~self.alives.any(axis=1)

    def step(self, actions):
        """manytor.py:255-260 + :175-213 for all envs at once.
        actions (N,D) degrees.  Returns obs2 (N,3K), reward (N,) int, done (N,) bool."""
        actions = np.asarray(actions, dtype=self.dtype).reshape(self.n, self.dof)
        self.get_observations()                         # pre-action pass: side effect only
        initial = self.alives.copy()
        start = self.goals
        step = (actions - start) / (self.substeps - 1)  # np.linspace arithmetic
        ground = np.zeros(self.n, dtype=bool)
        margin = np.full(self.n, np.inf)
        zmin = np.full(self.n, np.inf)
        for k in range(self.substeps):
            pose = actions if k == self.substeps - 1 else start + k * step
            jc = batch_joints_coordinates(pose, self.table, self.dtype)
            z2 = jc[:, self.obs_frame, 2]
            z3 = jc[:, self.ee_frame, 2]
            ground |= (z2 < 0) | (z3 < 0)
            margin = np.minimum(margin, np.minimum(np.abs(z2), np.abs(z3)))
            zmin = np.minimum(zmin, np.minimum(z2, z3))
        self.goals = actions.copy()
        self.joints_coordinates = jc
        obs2 = self.get_observations()                  # before pickup, manytor.py:204
        self.is_done()
        picked = initial.sum(axis=1) > self.alives.sum(axis=1)
        reward = np.where(ground, -1, np.where(picked, 1, 0)).astype(np.int64)
        self.total_reward = self.total_reward + reward
        done = ~self.alives.any(axis=1)
        self.ground_hit = ground
        self.ground_margin = margin
        self.zmin = zmin
        return obs2, reward, done


# --------------------------------------------------------------------------
# Host RNG streams of the reference (global numpy MT19937)
# --------------------------------------------------------------------------
def reference_target_stream(n_envs, obj_number, radius=REF_RADIUS):
    """Targets for ``n_envs`` resets in env order, drawn scalar by scalar from
    the global numpy RNG exactly as manytor.py:229-239 does."""
    out = np.empty((n_envs, obj_number, 3))
    for e in range(n_envs):
        cnt = 0
        while cnt < obj_number:
            c = [np.random.uniform(-radius, radius) for _ in range(3)]
            if c[2] >= 0 and math.sqrt(math.sqrt(c[0] ** 2 + c[1] ** 2) ** 2 + c[2] ** 2) <= radius:
                out[e, cnt] = c
                cnt += 1
    return out


def reference_action_stream(n_envs, dof=4):
    """manytor.py:215-217 / :111-113: env-major, joint-minor scalar randint."""
    return np.array(
        [[np.random.randint(low=-180, high=180, size=1)[0] for _ in range(dof)] for _ in range(n_envs)],
        dtype=np.int64,
    )
