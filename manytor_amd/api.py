"""The gym-style surface of the reference (``import manytor as tor``) on the HIP engine.

Same names, defaults and return arity as /root/reference/manytor.py:
``Environment`` (:125-283), ``Multienv`` (:72-122), ``dh`` / ``fk`` / ``r_theta``
(:17-53), ``HOST`` / ``PORT`` (:9-10).  Every number is computed on the GPU by
libmanytor_hip.so; there is no CPU path (constructing an env without a GPU raises).

Two RNG modes:
  rng="numpy"  (default) targets and sampled actions come from numpy's global
               RandomState in the reference's draw order -> ``np.random.seed(s)``
               reproduces a reference run bit-for-bit on the inputs.
  rng="device" Philox-4x32-10 on the GPU keyed by (seed, global env id, episode /
               step): nothing crosses PCIe; results independent of sharding.
"""
from __future__ import annotations

import math

import numpy as np

from . import _lib as L
from . import rng as _rng
from .engine import REF_DH_TABLE, StepEngine, fk_batch, r_theta_batch, route_trace
from .viewer import ViewerLink

HOST = "localhost"     # manytor.py:9   (viewer address; rendering itself is out of scope here)
PORT = 5001            # manytor.py:10

_MATERIALIZE_LIMIT = 4096   # up to this many envs, step()/reset() return real Python lists like the reference


# ---- module functions (manytor.py:17-53) --------------------------------------------------------
def r_theta(v1, v2):
    """manytor.py:17-22 -> (r_deg, theta_deg)."""
    out = r_theta_batch(np.asarray(v1, dtype=np.float64)[:3], np.asarray(v2, dtype=np.float64)[:3])[0]
    return float(out[0]), float(out[1])


def dh(a, alfa, d, theta):
    """manytor.py:25-32 -> 4x4 homogeneous transform (theta, alfa in radians)."""
    return fk_batch(1, [[theta]], dh_table=[(a, alfa, d, 0.0)], radians=True)[0].astype(np.float64)


def fk(mode, goals):
    """manytor.py:35-53 -> 4x4 transform of the first `mode` joints of the reference arm (goals in degrees)."""
    return fk_batch(mode, [list(goals)[:4]], dh_table=REF_DH_TABLE)[0].astype(np.float64)


# ---- sequence proxies -----------------------------------------------------------------------
class BatchView:
    """Read-only sequence over the env axis of a device field, for N too large to turn into Python
    objects (never materialise 1 M objects).  ``len``/indexing/iteration behave like the
    reference's lists; ``.numpy()`` is one D2H copy; ``.torch()`` is a zero-copy device view.
    Deliberately no ``__eq__``: ``done == True`` is False, as it is for the reference's list
    (test_multi.py:22)."""

    def __init__(self, engine, field, post=None):
        self._e, self._f, self._post = engine, field, post
        self._host = None
        self._ver = engine.version

    def numpy(self):
        if self._host is None:
            if self._ver != self._e.version:
                raise RuntimeError("stale BatchView: the engine has stepped since this result was returned")
            a = self._e.get(self._f)
            self._host = self._post(a) if self._post else a
        return self._host

    def torch(self):
        t = self._e.device_tensor(self._f)
        return t if t.dim() == 1 else t.T

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a.astype(dtype) if dtype is not None else a

    def __len__(self):
        return self._e.n_envs

    def __getitem__(self, i):
        return self.numpy()[i]

    def __iter__(self):
        return iter(self.numpy())


class _EnvView:
    """``multienv.environment[i]``: one env of the batch with the reference ``Environment``'s attributes and methods
    (the reference's list holds real Environment objects, manytor.py:82, so callers may step / reset one of them).
    Whole-batch calls on the Multienv are the fast path; these touch a single env (mt_env_step / mt_env_reset) and
    leave the others alone."""

    def __init__(self, owner, index):
        self._o, self.id = owner, index

    def _write(self, field, value, dtype):
        full = self._o._engine.get(field)
        full[self.id] = np.asarray(value, dtype=dtype).reshape(full[self.id].shape)
        self._o._engine.set(field, full)

    obj_number = property(lambda s: s._o.obj_number)
    rendering = property(lambda s: s._o.rendering)
    goals = property(lambda s: s._o._cached(L.F_GOALS)[s.id].astype(np.float64),
                     lambda s, v: s._write(L.F_GOALS, v, np.float32))
    alives = property(lambda s: s._o._cached(L.F_ALIVE)[s.id].astype(bool),
                      lambda s, v: s._write(L.F_ALIVE, v, np.uint8))
    points = property(lambda s: s._o._cached(L.F_POINTS)[s.id].astype(np.float64),
                      lambda s, v: s._write(L.F_POINTS, v, np.float32))
    joints_coordinates = property(lambda s: s._o._cached(L.F_JOINTS)[s.id].astype(np.float64))
    total_reward = property(lambda s: float(s._o._cached(L.F_TOTAL_REWARD)[s.id]),
                            lambda s, v: s._write(L.F_TOTAL_REWARD, v, np.float32))

    def get_observations(self):
        """manytor.py:141-153 (recomputed for the batch; row `id` returned)."""
        self._o._engine.observe()
        return self._o._cached(L.F_OBS)[self.id].astype(np.float64)

    get_obs = get_observations

    def is_done(self):
        """manytor.py:155-173."""
        self._o._engine.check_done()
        return bool(self._o._cached(L.F_DONE)[self.id])

    def action_sample(self):
        """manytor.py:215-217."""
        o = self._o
        if o.rng == "numpy":
            return list(_rng.draw_actions(1, o._engine.dof)[0])
        return [np.int64(v) for v in np.random.randint(-180, 180, size=o._engine.dof)]

    def step(self, action):
        """manytor.py:255-260 for this env only -> (obs2, reward, done)."""
        obs, rew, done = self._o._engine.env_step(self.id, action)
        return obs.astype(np.float64), rew, done

    def action(self, action, obs=None):
        """manytor.py:175-213 -> (reward, obs2); does not accumulate into total_reward (that is step's job, :258)."""
        before = self.total_reward
        obs2, rew, _ = self.step(action)
        self.total_reward = before
        return rew, obs2

    def reset(self, returnable=False):
        """manytor.py:219-253 for this env only."""
        o = self._o
        if o.rng == "numpy":
            o._engine.env_reset(self.id, _rng.draw_targets(1, o.obj_number, o.radius)[0])
        else:
            # fresh targets at every call (manytor.py:229 draws anew each time) WITHOUT touching the env's episode
            # counter: the draw is keyed by a per-env count of single-env resets folded into the seed, and the env keeps
            # the episode index it has, so finished() / the next whole-batch reset are unaffected
            cnt = o._env_resets.get(self.id, 0) + 1
            o._env_resets[self.id] = cnt
            seed = (o.seed ^ (cnt * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF
            o._engine.env_reset(self.id, None, seed=seed, episode=int(o._cached(L.F_EPISODES)[self.id]))
        if returnable:
            return self.get_observations()


class _EnvList:
    def __init__(self, owner):
        self._o = owner

    def __len__(self):
        return self._o.env_number

    def __getitem__(self, i):
        n = self._o.env_number
        if isinstance(i, slice):
            return [_EnvView(self._o, j) for j in range(*i.indices(n))]
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError("list index out of range")
        return _EnvView(self._o, i)

    def __iter__(self):
        return (_EnvView(self._o, j) for j in range(self._o.env_number))


# ---- Multienv (manytor.py:72-122) --------------------------------------------------------------
class Multienv:
    """``Multienv(env_shape=(1, 2), obj_number=5)``: env_shape[0]*env_shape[1] arms stepped in one launch."""

    def __init__(self, env_shape=(1, 2), obj_number=5, *, rng="numpy", seed=0x5EED, device=0, env_id_base=0,
                 dh_table=REF_DH_TABLE, substeps=25, pickup_tol=8.0, radius=51.3, terminate_on_ground=False,
                 materialize="auto", viewer=None, view_envs=16, **engine_kwargs):
        if rng not in ("numpy", "device"):
            raise ValueError("rng must be 'numpy' or 'device'")
        self._viewer = ViewerLink(*viewer) if isinstance(viewer, (tuple, list)) else viewer
        self._view_envs = int(view_envs)
        self._dh_table = dh_table
        self._substeps = substeps
        self.env_shape = env_shape
        self.env_number = env_shape[0] * env_shape[1]
        self.obj_number = obj_number
        self.rendering = False
        self.rng = rng
        self.seed = int(seed)
        self.radius = radius
        self._episode = 0
        self._step_idx = 0
        self._engine = StepEngine(self.env_number, obj_number, dh_table=dh_table, substeps=substeps,
                                  pickup_tol=pickup_tol, radius=radius, device=device, env_id_base=env_id_base,
                                  terminate_on_ground=terminate_on_ground, **engine_kwargs)
        self._materialize = (self.env_number <= _MATERIALIZE_LIMIT) if materialize == "auto" else bool(materialize)
        self._cache = {}
        self._env_resets = {}          # env index -> number of environment[i].reset() calls (device RNG mode)
        self.environment = _EnvList(self)

    @property
    def engine(self) -> StepEngine:
        return self._engine

    def _cached(self, field):
        ver = self._engine.version
        hit = self._cache.get(field)
        if hit is None or hit[0] != ver:
            hit = (ver, self._engine.get(field))
            self._cache[field] = hit
        return hit[1]

    def render(self, stop_render=False):
        """manytor.py:84-104.  Keeps the `rendering` flag (control flow of test_multi.py:25-28).  If a ViewerLink was
        given (`viewer=(host, port)`), the reference's init / stop datagrams are sent and, while rendering, every
        step streams the sub-step frames of the first `view_envs` envs (manytor_amd/viewer.py).  The viewer process
        itself is never spawned here."""
        self.rendering = not stop_render
        if self._viewer is not None:
            if stop_render:
                self._viewer.stop()
            else:
                self._viewer.init(min(self.env_number, self._view_envs), self.obj_number, self.env_shape)

    def _stream_frames(self, prev_goals):
        m = min(self.env_number, self._view_envs)
        goals = self._cached(L.F_GOALS)[:m]
        traces = route_trace(prev_goals[:m], goals, dh_table=self._dh_table, substeps=self._substeps,
                             device=self._engine.device)
        self._viewer.frames(list(range(m)), traces, self._cached(L.F_POINTS)[:m], first=False)

    def reset(self, returnable=False):
        """manytor.py:106-109."""
        if self.rng == "numpy":
            self._engine.reset(_rng.draw_targets(self.env_number, self.obj_number, self.radius))
        else:
            self._engine.reset_random(self.seed, self._episode)
        self._episode += 1
        if self.rendering and self._viewer is not None:
            self._viewer.clear()                       # manytor.py:246-249
        if returnable:
            self._engine.observe()
            if self._materialize:
                return [row.astype(np.float64) for row in self._engine.obs()]
            return BatchView(self._engine, L.F_OBS)

    def action_sample(self):
        """manytor.py:111-113.  numpy mode: list of per-env lists of np.int64 (like the reference) for small N,
        an (N, D) int64 array otherwise.  device mode: draws into the device action buffer and returns None-like
        token ``self.DEVICE_ACTIONS``; pass it to step()."""
        if self.rng == "numpy":
            a = _rng.draw_actions(self.env_number, self._engine.dof)
            return [list(row) for row in a] if self._materialize else a
        self._engine.sample_actions(self.seed, self._step_idx)
        return DEVICE_ACTIONS

    def step(self, action):
        """manytor.py:115-122 -> (obs2_list, reward_list, done_list)."""
        streaming = self.rendering and self._viewer is not None
        prev = self._cached(L.F_GOALS).copy() if streaming else None
        host_action = action is not DEVICE_ACTIONS and not (hasattr(action, "is_cuda") and action.is_cuda)
        if self._materialize and host_action:
            obs, rew, done = self._engine.step_host(action)      # one round trip, one synchronisation
            self._step_idx += 1
            if streaming:
                self._stream_frames(prev)
            return [row.astype(np.float64) for row in obs], [int(r) for r in rew], [bool(d) for d in done]
        if action is DEVICE_ACTIONS:
            self._engine.step()
        else:
            self._engine.step(action)
        self._step_idx += 1
        if streaming:
            self._stream_frames(prev)
        if self._materialize:
            e = self._engine
            return ([row.astype(np.float64) for row in e.obs()], [int(r) for r in e.reward()],
                    [bool(d) for d in e.done()])
        e = self._engine
        return (BatchView(e, L.F_OBS), BatchView(e, L.F_REWARD), BatchView(e, L.F_DONE, post=lambda a: a.astype(bool)))

    def close(self):
        self._engine.close()


class _DeviceActions:
    """Token returned by action_sample() in device mode: 'the actions already in the device buffer'."""

    def __repr__(self):
        return "<actions resident on the device>"


DEVICE_ACTIONS = _DeviceActions()
Multienv.DEVICE_ACTIONS = DEVICE_ACTIONS


# ---- Environment (manytor.py:125-283) ---------------------------------------------------------
class Environment:
    """``Environment(obj_number=10, index=0)``: one arm = a batch of one on the same engine."""

    def __init__(self, obj_number=10, index=0, *, rng="numpy", seed=0x5EED, device=0, dh_table=REF_DH_TABLE,
                 substeps=25, pickup_tol=8.0, radius=51.3, terminate_on_ground=False, keep_trajectory=True,
                 viewer=None, obs_frame=-2, ee_frame=-1):
        if rng not in ("numpy", "device"):
            raise ValueError("rng must be 'numpy' or 'device'")
        self._viewer = ViewerLink(*viewer) if isinstance(viewer, (tuple, list)) else viewer
        self._keep_trajectory = bool(keep_trajectory)
        self._dh_table = dh_table
        self._substeps = substeps
        # manytor.py:135: end-effector trace, one row per sub-step, seeded with (0, 0, 51.3).  Kept lazily: a step only
        # notes its (previous pose, action) pair; the rows are computed -- one mt_route_trace call for all pending
        # steps -- when `trajectory` is read.
        self._traj = np.array([0.0, 0.0, 51.3])
        self._traj_pending = []
        self.id = index
        self.obj_number = obj_number
        self.rendering = False
        self.rng = rng
        self.seed = int(seed)
        self.radius = radius
        self._episode = 0
        self._step_idx = 0
        self._engine = StepEngine(1, obj_number, dh_table=dh_table, substeps=substeps, pickup_tol=pickup_tol,
                                  radius=radius, device=device, env_id_base=index,
                                  terminate_on_ground=terminate_on_ground, obs_frame=obs_frame, ee_frame=ee_frame)
        # the reference constructor leaves points/joints empty until reset() (manytor.py:136-137);
        # arm the device state at the zero pose so attribute reads are defined
        self._engine.reset(np.zeros((1, obj_number, 3), dtype=np.float32))
        self._pose = np.zeros((1, self._engine.dof), dtype=np.float32)   # host mirror of `goals` (= the last action)

    # state attributes (manytor.py:131-139); reads are D2H copies, writes are H2D
    def _set_goals(self, v):
        self._pose = np.asarray(v, dtype=np.float32).reshape(1, -1).copy()
        self._engine.set(L.F_GOALS, self._pose)

    goals = property(lambda s: s._engine.goals()[0].astype(np.float64), _set_goals)
    alives = property(lambda s: s._engine.alives()[0],
                      lambda s, v: s._engine.set(L.F_ALIVE, np.asarray(v, dtype=np.uint8).reshape(1, -1)))
    points = property(lambda s: s._engine.points()[0].astype(np.float64),
                      lambda s, v: s._engine.set(L.F_POINTS, np.asarray(v, dtype=np.float32).reshape(1, -1, 3)))
    total_reward = property(lambda s: float(s._engine.total_reward()[0]),
                            lambda s, v: s._engine.set(L.F_TOTAL_REWARD, np.asarray([v], dtype=np.float32)))
    joints_coordinates = property(lambda s: s._engine.joints_coordinates()[0].astype(np.float64))

    @property
    def trajectory(self):
        """manytor.py:135,190: (0, 0, 51.3) + the end effector at every sub-step since the last reset."""
        if self._traj_pending:
            prev = np.concatenate([p for p, _ in self._traj_pending])
            act = np.concatenate([a for _, a in self._traj_pending])
            self._traj_pending = []
            trace = route_trace(prev, act, dh_table=self._dh_table, substeps=self._substeps, device=self._engine.device)
            self._traj = np.vstack((self._traj, trace[:, :, -1, :].reshape(-1, 3).astype(np.float64)))
        return self._traj

    @trajectory.setter
    def trajectory(self, value):
        self._traj_pending = []
        self._traj = np.asarray(value, dtype=np.float64)

    @property
    def engine(self) -> StepEngine:
        return self._engine

    def get_observations(self):
        """manytor.py:141-153."""
        self._engine.observe()
        return self._engine.obs()[0].astype(np.float64)

    get_obs = get_observations      # the name BASELINE.json's north_star uses

    def is_done(self):
        """manytor.py:155-173."""
        self._engine.check_done()
        return bool(self._engine.done()[0])

    @staticmethod
    def _usable(act) -> bool:
        """The kernel's own test (kernels.h: unusable_angle): an action with a NaN / inf / |angle| > 32768 component is
        rejected and the env holds its pose for that step."""
        a = np.asarray(act, dtype=np.float32)
        return bool(np.all(np.isfinite(a)) and np.all(np.abs(a) <= 32768.0))

    def _after_route(self, prev, action):
        """Host-side bookkeeping the reference does per sub-step (manytor.py:190, :194-202): the trajectory rows (noted
        here, computed when read) and, while rendering, the viewer frames of the route just taken."""
        streaming = self.rendering and self._viewer is not None
        if self._keep_trajectory:
            self._traj_pending.append((np.asarray(prev, dtype=np.float32).reshape(1, -1),
                                       np.asarray(action, dtype=np.float32).reshape(1, -1)))
        if streaming:
            trace = route_trace(prev, action, dh_table=self._dh_table, substeps=self._substeps, device=self._engine.device)
            self._viewer.frames([self.id], trace, self._engine.points(), first=False)

    def action(self, action, obs=None):
        """manytor.py:175-213 -> (reward, obs2).  Like the reference it does not add to total_reward;
        the fused step kernel does, so the increment is taken back out."""
        before = self._engine.total_reward()
        prev = self._pose
        act = np.asarray(action, dtype=np.float64).reshape(1, -1)
        self._engine.step(act)
        self._engine.set(L.F_TOTAL_REWARD, before)
        if not self._usable(act):                         # rejected by the kernel: the pose was held
            act = prev.astype(np.float64)
        self._pose = act.astype(np.float32)               # goals = action after a step (manytor.py:184)
        self._after_route(prev, act)
        return int(self._engine.reward()[0]), self._engine.obs()[0].astype(np.float64)

    def action_sample(self):
        """manytor.py:215-217 -> list of D integer degrees in [-180, 180)."""
        if self.rng == "numpy":
            return list(_rng.draw_actions(1, self._engine.dof)[0])
        self._engine.sample_actions(self.seed, self._step_idx)
        return [np.int64(v) for v in self._engine.actions()[0]]

    def reset(self, returnable=False):
        """manytor.py:219-253."""
        if self.rng == "numpy":
            self._engine.reset(_rng.draw_targets(1, self.obj_number, self.radius))
        else:
            self._engine.reset_random(self.seed, self._episode)
        self._episode += 1
        self._pose = np.zeros_like(self._pose)
        self.trajectory = np.array([0.0, 0.0, 51.3])      # manytor.py:223 (drops pending rows)
        if self.rendering and self._viewer is not None:
            self._viewer.clear()                           # manytor.py:246-249
        if returnable:
            return self.get_observations()

    def step(self, action):
        """manytor.py:255-260 -> (obs2, reward, done)."""
        prev = self._pose
        act = np.asarray(action, dtype=np.float64).reshape(1, -1)
        obs, rew, done = self._engine.step_host(act)
        self._step_idx += 1
        if not self._usable(act):                         # rejected by the kernel: the pose was held
            act = prev.astype(np.float64)
        self._pose = act.astype(np.float32)               # goals = action after a step (manytor.py:184)
        self._after_route(prev, act)
        return obs[0].astype(np.float64), int(rew[0]), bool(done[0])

    def render(self, stop_render=False, multienv=False):
        """manytor.py:262-283.  Flag always; init / stop datagrams if a ViewerLink was given (`viewer=(host, port)`).
        The viewer process is never spawned here."""
        self.rendering = not stop_render
        if self._viewer is not None and not multienv:
            if stop_render:
                self._viewer.stop()
            else:
                self._viewer.init(1, self.obj_number)

    def close(self):
        self._engine.close()
