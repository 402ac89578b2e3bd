"""StepEngine: thin object wrapper over one mt_handle (include/manytor_hip.h).

One engine = N lock-stepped arms resident on one MI355X.  All arithmetic runs in
the HIP kernels; this file only marshals arguments.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _lib as L

# DH rows (a, alpha, d, theta_offset) of the reference arm, manytor.py:42-48.
REF_DH_TABLE = (
    (0.0, -math.pi / 2, 4.3, 0.0),
    (0.0, math.pi / 2, 0.0, 0.0),
    (0.0, -math.pi / 2, 24.3, 0.0),
    (27.0, math.pi / 2, 0.0, -math.pi / 2),
)

# A 7-joint table for the generalised-DH configuration (BASELINE.json configs[4]);
# pinned by fixture F7 (tests/golden/f7_dh7_kat.npz).
DH7_TABLE = (
    (0.0, -math.pi / 2, 34.0, 0.0),
    (0.0, math.pi / 2, 0.0, 0.0),
    (4.5, math.pi / 2, 40.0, 0.0),
    (-4.5, -math.pi / 2, 0.0, 0.0),
    (0.0, -math.pi / 2, 40.0, 0.0),
    (8.8, math.pi / 2, 0.0, -math.pi / 2),
    (0.0, 0.0, 12.6, 0.0),
)

_NP_OF_DT = {L.DT_F32: np.float32, L.DT_I32: np.int32, L.DT_U8: np.uint8, L.DT_U32: np.uint32, L.DT_U64: np.uint64}
_TYPESTR = {L.DT_F32: "<f4", L.DT_I32: "<i4", L.DT_U8: "|u1", L.DT_U32: "<u4", L.DT_U64: "<u8"}
_ACTION_DT = {np.dtype(np.float32): L.DT_F32, np.dtype(np.float64): L.DT_F64, np.dtype(np.int32): L.DT_I32,
              np.dtype(np.int64): L.DT_I64}


def _is_torch_tensor(x) -> bool:
    return type(x).__module__.startswith("torch") and hasattr(x, "data_ptr")


class _CudaArrayHolder:
    """Exposes a raw device range through __cuda_array_interface__ (consumed by torch.as_tensor)."""

    def __init__(self, ptr, shape, strides, typestr, owner):
        self.__cuda_array_interface__ = {
            "shape": tuple(shape), "strides": tuple(strides), "typestr": typestr, "data": (int(ptr), False), "version": 2,
        }
        self._owner = owner     # keeps the engine (and its arena) alive


class StepEngine:
    """N lock-stepped manipulator envs on one GPU.

    Mirrors, for a batch, the state and methods of the reference ``Environment``
    (manytor.py:125-260): reset / step / get_observations / is_done /
    action_sample.  Field getters return arrays in the reference's shapes with a
    leading env axis.
    """

    def __init__(self, n_envs, obj_number=10, dh_table=REF_DH_TABLE, substeps=25, pickup_tol=8.0, radius=51.3,
                 device=0, env_id_base=0, terminate_on_ground=False, hw_trig=False, dh_in_lds=False,
                 direct_trig=False, specialize=True, ablate=0, return_ring=4, trace=False, obs_frame=-2, ee_frame=-1,
                 debug_zmin=False):
        self._lib = L.load()
        table = np.asarray(dh_table, dtype=np.float64)
        if table.ndim != 2 or table.shape[1] != 4:
            raise ValueError("dh_table must be (dof, 4): rows (a, alpha, d, theta_offset)")
        self.n_envs = int(n_envs)
        self.obj_number = int(obj_number)
        self.dof = int(table.shape[0])
        self.substeps = int(substeps)
        self.device = int(device)
        self.env_id_base = int(env_id_base)
        self.dh_table = table
        cfg = L.MtConfig()
        cfg.struct_size = C.sizeof(L.MtConfig)
        cfg.device = self.device
        cfg.n_envs = self.n_envs
        cfg.env_id_base = self.env_id_base
        cfg.dof = self.dof
        cfg.n_targets = self.obj_number
        cfg.substeps = self.substeps
        cfg.flags = ((L.FLAG_TERMINATE_ON_GROUND if terminate_on_ground else 0) | (L.FLAG_HW_TRIG if hw_trig else 0)
                     | (L.FLAG_DH_IN_LDS if dh_in_lds else 0) | (L.FLAG_DIRECT_TRIG if direct_trig else 0)
                     | (0 if specialize else L.FLAG_NO_SPECIALIZE) | (L.FLAG_TRACE if trace else 0)
                     | (L.FLAG_DEBUG_ZMIN if debug_zmin else 0)
                     | (L.FLAG_ABLATE_LOOP if ablate in (1, 2) else 0) | (L.FLAG_ABLATE_OBS if ablate in (2, 3) else 0))
        cfg.pickup_tol = float(pickup_tol)
        cfg.radius = float(radius)
        cfg.return_ring = int(return_ring)
        cfg.obs_frame = int(obs_frame)      # rows of joints_coordinates: observation from [-2], pickup from [-1] in the
        cfg.ee_frame = int(ee_frame)        # reference (manytor.py:143, :162); selectable for other arms
        cfg.reserved = 0
        self.return_ring_slots = int(return_ring)
        self.has_trace = bool(trace)
        self.episode0 = 0         # episode index every env got at the last full reset
        if self.dof > L.MT_MAX_DOF:
            raise ValueError(f"dof must be <= {L.MT_MAX_DOF}")
        flat = table.astype(np.float32).ravel()
        for i, v in enumerate(flat):
            cfg.dh_table[i] = float(v)
        self._h = L._HANDLE()
        L.check(self._lib.mt_create(C.byref(self._h), C.byref(cfg)))
        self.version = 0          # bumped by every call that changes device state (host caches key on it)

    # ---- lifetime ---------------------------------------------------------------------------
    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.mt_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _call(self, fn, *args):
        if not self._h:
            raise RuntimeError("StepEngine is closed")
        L.check(fn(self._h, *args), self._h)

    def step_kernel_name(self) -> str:
        """The kernel instantiation step() / step_random() launch for this batch (schedule picked at construction)."""
        if not self._h:
            raise RuntimeError("StepEngine is closed")
        return self._lib.mt_step_kernel_name(self._h).decode()

    def dispatch(self) -> dict:
        """The dispatch of this engine as data (mt_describe_dispatch): the schedule resolved for every entry point, the
        MT_* overrides in effect and the library's table of size thresholds."""
        import json
        if not self._h:
            raise RuntimeError("StepEngine is closed")
        return json.loads(self._lib.mt_describe_dispatch(self._h).decode())

    # ---- stream / sync / timing ---------------------------------------------------------------
    def set_stream(self, hip_stream):
        """Run on a caller-owned hipStream_t (int / pointer).  0 is the legacy default stream (torch's default
        stream); None goes back to the engine's own private stream, which is ordered against nothing else."""
        if hip_stream is None:
            self._call(self._lib.mt_use_own_stream)
        else:
            self._call(self._lib.mt_set_stream, C.c_void_p(int(hip_stream)))

    def use_torch_stream(self):
        """Launch on torch's current stream of this device: torch ops on that stream (device_tensor views, RCCL
        collectives) and engine launches then execute in program order, no manual sync needed."""
        import torch
        self.set_stream(int(torch.cuda.current_stream(self.device).cuda_stream))

    def sync(self):
        self._call(self._lib.mt_sync)

    def timer_start(self):
        self._call(self._lib.mt_timer_start)

    def timer_stop(self) -> float:
        ms = C.c_float(0)
        self._call(self._lib.mt_timer_stop, C.byref(ms))
        return ms.value

    def timer_stop_async(self):
        """End mark of timer_start without joining the chains or waiting on the host; timer_read() waits and returns."""
        self._call(self._lib.mt_timer_stop_async)

    def timer_read(self) -> float:
        """Device milliseconds from timer_start to the last end event of timer_stop_async (all streams of the engine,
        incl. an exchange that was pending on the side stream)."""
        ms = C.c_float(0)
        self._call(self._lib.mt_timer_read, C.byref(ms))
        return ms.value

    def lap_begin(self):
        self._call(self._lib.mt_timer_lap_begin)

    def lap_end(self):
        self._call(self._lib.mt_timer_lap_end)

    def laps_total(self):
        """(summed device milliseconds of all laps since the last call, number of laps); synchronises once."""
        ms, n = C.c_float(0), C.c_int(0)
        self._call(self._lib.mt_timer_laps_total, C.byref(ms), C.byref(n))
        return ms.value, n.value

    def lap_times(self):
        """Device milliseconds of every lap since the last call, in recording order; synchronises once."""
        cap = 4096
        while True:
            buf, n = (C.c_float * cap)(), C.c_int(0)
            rc = self._lib.mt_timer_lap_times(self._h, buf, cap, C.byref(n))
            if rc != L.MT_OK and n.value > cap:
                cap = n.value
                continue
            L.check(rc, self._h)
            return list(buf[: n.value])

    # ---- reset --------------------------------------------------------------------------------
    def reset(self, points):
        """Environment.reset (manytor.py:219-253) with caller-supplied targets: (N, K, 3) host array
        or device tensor (env-major), or a (3K, ld) float32 device tensor (SoA)."""
        n, k = self.n_envs, self.obj_number
        if _is_torch_tensor(points):
            import torch
            t = points
            if not t.is_cuda:
                return self.reset(t.detach().cpu().numpy())
            if t.dtype != torch.float32 or not t.is_contiguous():
                t = t.to(torch.float32).contiguous()
            if t.numel() == n * k * 3:
                layout = L.ENV_MAJOR
            elif tuple(t.shape) == (3 * k, self.ld):
                layout = L.SOA
            else:
                raise ValueError("device points must be (N,K,3) or (3K, ld)")
            torch.cuda.current_stream(self.device).synchronize()
            self._call(self._lib.mt_reset, C.c_void_p(t.data_ptr()), layout, 1)
            self.sync()
        else:
            p = np.ascontiguousarray(np.asarray(points, dtype=np.float32).reshape(n, k, 3))
            self._call(self._lib.mt_reset, p.ctypes.data_as(C.c_void_p), L.ENV_MAJOR, 0)
        self.episode0 = 0
        self._armed = True
        self.version += 1

    def reset_random(self, seed=0x5EED, episode=0):
        self._call(self._lib.mt_reset_random, C.c_uint64(seed), C.c_uint32(episode))
        self.episode0 = int(episode)
        self._armed = True
        self.version += 1

    def env_reset(self, env, points=None, seed=0x5EED, episode=0):
        """``multienv.environment[env].reset()`` (manytor.py:82 + :219-253): reset ONE env; `points` (K, 3) or None
        to draw them on the device."""
        p = None
        if points is not None:
            p = np.ascontiguousarray(np.asarray(points, dtype=np.float32).reshape(self.obj_number, 3))
        self._call(self._lib.mt_env_reset, C.c_int64(int(env)), p.ctypes.data_as(C.c_void_p) if p is not None else None,
                   C.c_uint64(seed), C.c_uint32(episode))
        self.version += 1

    def reset_done(self, seed=0x5EED):
        """Re-arm the envs whose done flag is set (new targets keyed by their own episode counter)."""
        self._call(self._lib.mt_reset_done, C.c_uint64(seed))
        self.version += 1

    # ---- actions ------------------------------------------------------------------------------
    def set_actions(self, actions):
        """(N, D) degrees: nested lists, numpy (f32/f64/i32/i64) or a device tensor; or (D, ld) SoA device tensor."""
        n, d = self.n_envs, self.dof
        if _is_torch_tensor(actions) and actions.is_cuda:
            import torch
            t = actions
            dt = {torch.float32: L.DT_F32, torch.float64: L.DT_F64, torch.int32: L.DT_I32, torch.int64: L.DT_I64}.get(t.dtype)
            if dt is None:
                t, dt = t.to(torch.float32), L.DT_F32
            t = t.contiguous()
            if tuple(t.shape) == (n, d):
                layout = L.ENV_MAJOR
            elif tuple(t.shape) == (d, self.ld):
                layout = L.SOA
            else:
                raise ValueError(f"device actions must be ({n},{d}) or ({d},{self.ld})")
            torch.cuda.current_stream(self.device).synchronize()
            self._call(self._lib.mt_set_actions, C.c_void_p(t.data_ptr()), dt, layout, 1)
            self.sync()
            return
        if _is_torch_tensor(actions):
            actions = actions.detach().cpu().numpy()
        a = np.asarray(actions)
        if a.dtype not in _ACTION_DT:
            a = a.astype(np.float64)
        if a.size != n * d:
            raise ValueError(f"actions must have {n}x{d} elements, got shape {a.shape}")
        a = np.ascontiguousarray(a.reshape(n, d))
        self._call(self._lib.mt_set_actions, a.ctypes.data_as(C.c_void_p), _ACTION_DT[a.dtype], L.ENV_MAJOR, 0)

    def sample_actions(self, seed=0x5EED, step_idx=0):
        """Environment.action_sample (manytor.py:215-217) on the device, into the action buffer."""
        self._call(self._lib.mt_sample_actions, C.c_uint64(seed), C.c_uint32(step_idx))

    # ---- step ---------------------------------------------------------------------------------
    def step(self, actions=None):
        """Environment.step (manytor.py:255-260) for all envs.  Results stay on the device:
        read them with get()/device_tensor()."""
        if actions is not None:
            self.set_actions(actions)
        self._call(self._lib.mt_step)
        self.version += 1

    def step_host(self, actions):
        """One host round trip: (N, D) host actions in -> (obs2 (N, 3K) f32, reward (N,) i32, done (N,) bool) out,
        with a single synchronisation.  The small-batch path of the drop-in classes."""
        n, d = self.n_envs, self.dof
        a = np.asarray(actions)
        if a.dtype not in _ACTION_DT:
            a = a.astype(np.float64)
        if a.size != n * d:
            raise ValueError(f"actions must have {n}x{d} elements, got shape {a.shape}")
        a = np.ascontiguousarray(a.reshape(n, d))
        obs = np.empty((n, 3 * self.obj_number), dtype=np.float32)
        rew = np.empty(n, dtype=np.int32)
        done = np.empty(n, dtype=np.uint8)
        self._call(self._lib.mt_step_host, a.ctypes.data_as(C.c_void_p), _ACTION_DT[a.dtype],
                   obs.ctypes.data_as(C.c_void_p), rew.ctypes.data_as(C.c_void_p), done.ctypes.data_as(C.c_void_p))
        self.version += 1
        return obs, rew, done.astype(bool)

    def env_step(self, env, action):
        """``multienv.environment[env].step(action)`` (manytor.py:82,118): step ONE env of the batch, the others are
        untouched -> (obs2 (3K,) f32, reward int, done bool)."""
        a = np.ascontiguousarray(np.asarray(action, dtype=np.float32).reshape(self.dof))
        obs = np.empty(3 * self.obj_number, dtype=np.float32)
        rew, done = C.c_int32(0), C.c_uint8(0)
        self._call(self._lib.mt_env_step, C.c_int64(int(env)), a.ctypes.data_as(C.c_void_p), obs.ctypes.data_as(C.c_void_p),
                   C.byref(rew), C.byref(done))
        self.version += 1
        return obs, int(rew.value), bool(done.value)

    def bad_action_count(self) -> int:
        """How many (env, step) pairs so far were handed a non-finite / absurd action and held their pose instead."""
        c = C.c_uint64(0)
        self._call(self._lib.mt_bad_action_count, C.byref(c))
        return int(c.value)

    def step_random(self, seed=0x5EED, step_idx=0):
        self._call(self._lib.mt_step_random, C.c_uint64(seed), C.c_uint32(step_idx))
        self.version += 1

    def rollout(self, n_steps, seed=0x5EED, step_idx0=0):
        self._call(self._lib.mt_rollout, int(n_steps), C.c_uint64(seed), C.c_uint32(step_idx0))
        self.version += 1

    def rollout_fused(self, n_steps, seed=0x5EED, step_idx0=0, auto_reset=False):
        """n_steps random-action steps in ONE launch (state in registers / LDS between steps)."""
        self._call(self._lib.mt_rollout_fused, int(n_steps), C.c_uint64(seed), C.c_uint32(step_idx0),
                   1 if auto_reset else 0)
        self.version += 1

    def observe(self):
        """Environment.get_observations (manytor.py:141-153); result in field OBS."""
        self._call(self._lib.mt_observe)
        self.version += 1

    def check_done(self):
        """Environment.is_done (manytor.py:155-173); result in field DONE."""
        self._call(self._lib.mt_check_done)
        self.version += 1

    # ---- state access -------------------------------------------------------------------------
    def _shape_dtype(self, field):
        n, d, k = self.n_envs, self.dof, self.obj_number
        return {
            L.F_ACTIONS: ((n, d), np.float32), L.F_GOALS: ((n, d), np.float32), L.F_POINTS: ((n, k, 3), np.float32),
            L.F_ALIVE: ((n, k), np.uint8), L.F_OBS: ((n, 3 * k), np.float32), L.F_REWARD: ((n,), np.int32),
            L.F_DONE: ((n,), np.uint8), L.F_DONE_BITS: (((n + 63) // 64,), np.uint64), L.F_EE: ((n, 3), np.float32),
            L.F_TOTAL_REWARD: ((n,), np.float32), L.F_JOINTS: ((n, d, 3), np.float32),
            L.F_EPISODES: ((n,), np.uint32), L.F_LAST_RETURN: ((n,), np.float32),
            L.F_RETURN_RING: ((n, self.return_ring_slots), np.float32), L.F_TRACE: ((n, self.substeps, 3), np.float32),
            L.F_ZMIN: ((n,), np.float32),
        }[field]

    def get(self, field) -> np.ndarray:
        """Host copy of a field in the reference's shape (leading env axis)."""
        shape, dt = self._shape_dtype(field)
        out = np.empty(shape, dtype=dt)
        self._call(self._lib.mt_get, field, out.ctypes.data_as(C.c_void_p), C.c_int64(out.nbytes), 0)
        return out

    def set(self, field, value):
        shape, dt = self._shape_dtype(field)
        v = np.ascontiguousarray(np.asarray(value).astype(dt).reshape(shape))
        self._call(self._lib.mt_set, field, v.ctypes.data_as(C.c_void_p), C.c_int64(v.nbytes))
        self.version += 1

    def device_ptr(self, field):
        ptr, rows, ld, dt = C.c_void_p(), C.c_int64(), C.c_int64(), C.c_int()
        self._call(self._lib.mt_device_ptr, field, C.byref(ptr), C.byref(rows), C.byref(ld), C.byref(dt))
        return ptr.value, rows.value, ld.value, dt.value

    @property
    def ld(self) -> int:
        return self.device_ptr(L.F_REWARD)[2]      # any field: the row stride is the handle's (not F_GOALS: see mt_device_ptr)

    def device_tensor(self, field):
        """Zero-copy torch view of the resident SoA buffer: shape (rows, N) (or (N,) for single-row fields,
        (ceil(N/64),) for DONE_BITS).  `.T` gives the reference's (N, rows) orientation."""
        import torch
        ptr, rows, ld, dt = self.device_ptr(field)
        es = np.dtype(_NP_OF_DT[dt]).itemsize
        n = (self.n_envs + 63) // 64 if field == L.F_DONE_BITS else self.n_envs
        if dt in (L.DT_U32, L.DT_U64):      # torch has limited unsigned support: expose as same-width signed
            typestr = "<i4" if dt == L.DT_U32 else "<i8"
        else:
            typestr = _TYPESTR[dt]
        shape, strides = ((n,), (es,)) if rows == 1 else ((rows, n), (ld * es, es))
        holder = _CudaArrayHolder(ptr, shape, strides, typestr, self)
        t = torch.as_tensor(holder, device=f"cuda:{self.device}")
        t._manytor_owner = holder
        return t

    # ---- checkpoint / resume (the reference keeps its state in attributes, manytor.py:131-139) --------
    def get_state(self) -> dict:
        """Host copy of everything later calls depend on: joint angles, targets, alive flags, returns, done flags,
        per-env episode counters, last returns, the return ring and the episode base (+ the step outputs a caller may
        want to keep).  set_state() of this dict on an engine of the same shape resumes bit-identically: step(),
        reset_done(), finished() and return_ring() all continue as in the original run."""
        st = {"goals": self.goals(), "points": self.points(), "alives": self.alives(), "total_reward": self.total_reward(),
              "obs": self.obs(), "reward": self.reward(), "done": self.get(L.F_DONE), "ee": self.ee(),
              "episodes": self.episodes(), "last_return": self.last_return(), "episode0": int(self.episode0)}
        if self.return_ring_slots:
            st["return_ring"] = self.return_ring()
        return st

    def set_state(self, state: dict):
        """Restore from get_state(): goals / points / alives / total_reward, and -- when present -- the raw done bytes
        (incl. 2 = finished and already re-armed), episode counters, last returns, return ring and episode base."""
        if not getattr(self, "_armed", False):
            # a fresh handle has to be reset once before it can step; the values are overwritten right below
            self.reset(np.asarray(state["points"], dtype=np.float32))
        self.set(L.F_POINTS, state["points"])
        self.set(L.F_GOALS, state["goals"])
        self.set(L.F_ALIVE, np.asarray(state["alives"]).astype(np.uint8))
        self.set(L.F_TOTAL_REWARD, state["total_reward"])
        for key, field in (("done", L.F_DONE), ("episodes", L.F_EPISODES), ("last_return", L.F_LAST_RETURN)):
            if key in state:
                self.set(field, state[key])
        if "return_ring" in state and self.return_ring_slots:
            self.set(L.F_RETURN_RING, state["return_ring"])
        if "episode0" in state:
            self.episode0 = int(state["episode0"])
            self._call(self._lib.mt_set_episode_base, C.c_uint32(self.episode0))

    # convenience getters in the reference's vocabulary
    def goals(self):
        return self.get(L.F_GOALS)

    def points(self):
        return self.get(L.F_POINTS)

    def alives(self):
        return self.get(L.F_ALIVE).astype(bool)

    def obs(self):
        return self.get(L.F_OBS)

    def reward(self):
        return self.get(L.F_REWARD)

    def done(self):
        return self.get(L.F_DONE).astype(bool)

    def ground_hit(self):
        """(N,) bool: the arm touched the ground at one of the sub-step poses of the last step (manytor.py:191-192).  The
        reference folds this into reward == -1 (:211-212) and does NOT end the episode on it;
        `terminate_on_ground=True` opts into ending it."""
        return self.get(L.F_REWARD) == -1

    def done_bits(self):
        return self.get(L.F_DONE_BITS)

    def ee(self):
        return self.get(L.F_EE)

    def total_reward(self):
        return self.get(L.F_TOTAL_REWARD)

    def joints_coordinates(self):
        return self.get(L.F_JOINTS)

    def episodes(self):
        return self.get(L.F_EPISODES)

    def last_return(self):
        return self.get(L.F_LAST_RETURN)

    def actions(self):
        return self.get(L.F_ACTIONS)

    def return_ring(self):
        """(N, R): slot c % R holds the return of the c-th episode the env finished since the last full reset."""
        return self.get(L.F_RETURN_RING)

    def finished(self):
        """(N,) number of episodes each env has finished (re-armed by reset_done / auto_reset) since the last full reset."""
        return (self.episodes() - np.uint32(self.episode0)).astype(np.int64)

    def zmin(self):
        """(N,) signed minimum z of the observation / pickup frames over all sub-step poses of the last step: the value
        the kernel's ground test compared with 0 (needs debug_zmin=True; manytor.py:191-192)."""
        return self.get(L.F_ZMIN)

    def trace(self):
        """(N, S, 3): end effector at each sub-step pose of the last step (needs trace=True; manytor.py:190)."""
        return self.get(L.F_TRACE)

    # ---- multi-GPU: the return gather -------------------------------------------------
    def comm_init(self, unique_id: bytes, rank: int, world_size: int):
        """Join the RCCL communicator named by `unique_id` (128 bytes from comm_unique_id() on rank 0).  Collective."""
        if len(unique_id) != L.MT_UNIQUE_ID_BYTES:
            raise ValueError("unique_id must be 128 bytes")
        buf = C.create_string_buffer(bytes(unique_id), L.MT_UNIQUE_ID_BYTES)
        self._call(self._lib.mt_comm_init, C.cast(buf, C.c_void_p), int(rank), int(world_size))

    def comm_destroy(self):
        self._call(self._lib.mt_comm_destroy)

    def total_envs(self) -> int:
        t = C.c_int64(0)
        self._call(self._lib.mt_comm_total_envs, C.byref(t))
        return int(t.value)

    def return_stats(self, field=None, row=0) -> dict:
        """{sum, min, max, mean, count, done} of a return row over ALL ranks (mt_reduce_returns): reduced on the device,
        five numbers per rank exchanged -- what a learner logs per episode without moving the row.  Synchronous;
        collective when a communicator is attached."""
        st = L.MtReturnStats()
        self._call(self._lib.mt_reduce_returns, int(L.F_TOTAL_REWARD if field is None else field), int(row), C.byref(st))
        return {"sum": st.sum, "min": st.min, "max": st.max, "mean": st.sum / max(1, st.count), "count": int(st.count),
                "done": int(st.done)}

    def _gather_out(self, out):
        import torch
        n_total = self.total_envs()
        if out is None:
            out = torch.empty(n_total, dtype=torch.float32, device=f"cuda:{self.device}")
        if out.dtype != torch.float32 or not out.is_cuda or not out.is_contiguous() or out.numel() != n_total:
            raise ValueError(f"out must be a contiguous float32 device tensor of {n_total} elements")
        return out, n_total

    def gather_begin(self, out=None, field=None, row=0, snapshot=True):
        """Start the all-gather of one return row on the engine's side stream and return at once: steps and resets queued
        next overlap with the exchange.  `out` is complete after gather_wait(host=True) or sync().  By default the row is
        snapshotted on the main stream first (so a reset queued next may overwrite it); `snapshot=False` lets the exchange
        read the row in place -- for MT_F_LAST_RETURN right after the reset that ended the episode (that row is written by
        resets only, and the library orders its later resets behind the exchange)."""
        field = L.F_TOTAL_REWARD if field is None else field
        out, n_total = self._gather_out(out)
        fn = self._lib.mt_gather_returns_begin if snapshot else self._lib.mt_gather_returns_begin_inplace
        self._call(fn, int(field), int(row), C.c_void_p(out.data_ptr()), C.c_int64(n_total))
        return out

    def gather_wait(self, host=False):
        """Order the engine's stream behind the last begun gather; host=True also blocks until its result is complete
        and returns the device milliseconds the exchange took."""
        ms = C.c_float(0)
        self._call(self._lib.mt_gather_returns_wait, 1 if host else 0, C.byref(ms))
        return ms.value if host else None

    def gather_returns(self, out=None, field=None, row=0):
        """All-gather one return row of every rank into `out`: a float32 device tensor of total_envs() elements (made
        if None), global env order.  RCCL straight from the arena on the engine's stream; a device copy on one GPU.
        Asynchronous: the result is ordered on the engine's stream (sync() or use_torch_stream() before reading)."""
        field = L.F_TOTAL_REWARD if field is None else field
        out, n_total = self._gather_out(out)
        self._call(self._lib.mt_gather_returns, int(field), int(row), C.c_void_p(out.data_ptr()), C.c_int64(n_total))
        return out


def comm_unique_id() -> bytes:
    """128 bytes naming a new RCCL communicator (call on rank 0, ship to the other ranks, then comm_init everywhere)."""
    lib = L.load()
    buf = C.create_string_buffer(L.MT_UNIQUE_ID_BYTES)
    L.check(lib.mt_comm_unique_id(C.cast(buf, C.c_void_p)))
    return buf.raw


# ---- stateless helpers = module functions of the reference (manytor.py:17-53) -------------------
def stream_probe(n_envs, dof=4, obj_number=7, reps=50, device=0):
    """(us per pass, bytes per pass) of the streaming yardstick: the memory operations of one step_random over n_envs
    envs -- same rows, same stores, same addressing -- without its arithmetic (mt_stream_probe)."""
    us, nbytes = C.c_float(0), C.c_int64(0)
    L.check(L.load().mt_stream_probe(int(device), int(dof), int(obj_number), C.c_int64(int(n_envs)), int(reps), C.byref(us),
                                     C.byref(nbytes)))
    return us.value, nbytes.value


def fk_batch(mode, angles, dh_table=REF_DH_TABLE, radians=False, device=0) -> np.ndarray:
    lib = L.load()
    table = np.ascontiguousarray(np.asarray(dh_table, dtype=np.float32))
    dof = table.shape[0]
    a = np.ascontiguousarray(np.asarray(angles, dtype=np.float32).reshape(-1, dof))
    out = np.empty((a.shape[0], 4, 4), dtype=np.float32)
    L.check(lib.mt_fk_batch(device, table.ctypes.data_as(C.c_void_p), dof, int(mode), a.ctypes.data_as(C.c_void_p),
                            1 if radians else 0, C.c_int64(a.shape[0]), out.ctypes.data_as(C.c_void_p)))
    return out


def route_trace(prev, action, dh_table=REF_DH_TABLE, substeps=25, device=0) -> np.ndarray:
    """joints_coordinates at every sub-step pose of the routes prev[i] -> action[i] (manytor.py:182-190):
    (n, dof) degrees in, (n, substeps, dof, 3) out."""
    lib = L.load()
    table = np.ascontiguousarray(np.asarray(dh_table, dtype=np.float32))
    dof = table.shape[0]
    p = np.ascontiguousarray(np.asarray(prev, dtype=np.float32).reshape(-1, dof))
    a = np.ascontiguousarray(np.asarray(action, dtype=np.float32).reshape(-1, dof))
    if p.shape != a.shape:
        raise ValueError("prev and action must have the same shape")
    out = np.empty((p.shape[0], int(substeps), dof, 3), dtype=np.float32)
    L.check(lib.mt_route_trace(device, table.ctypes.data_as(C.c_void_p), dof, int(substeps), p.ctypes.data_as(C.c_void_p),
                               a.ctypes.data_as(C.c_void_p), C.c_int64(p.shape[0]), out.ctypes.data_as(C.c_void_p)))
    return out


def r_theta_batch(v1, v2, device=0) -> np.ndarray:
    lib = L.load()
    a = np.ascontiguousarray(np.asarray(v1, dtype=np.float32).reshape(-1, 3))
    b = np.ascontiguousarray(np.asarray(v2, dtype=np.float32).reshape(-1, 3))
    out = np.empty((a.shape[0], 2), dtype=np.float32)
    L.check(lib.mt_r_theta_batch(device, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p),
                                 C.c_int64(a.shape[0]), out.ctypes.data_as(C.c_void_p)))
    return out
