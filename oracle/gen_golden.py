#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

Run in the build container only (needs /root/reference, which never travels to
the GPU box):

    python oracle/gen_golden.py

It imports /root/reference/manytor.py unmodified, seeds numpy's global RNG,
drives the reference's own functions/classes with rendering off, and stores
inputs + outputs as small .npz files (pure data: no reference source text).
Fixture ids follow SURVEY.md section 8(c): F1..F7, plus F8 (dh / r_theta KATs) and F9 (the viewer datagrams the
reference emits while stepping, captured without starting its viewer).
"""
import os
import sys

sys.dont_write_bytecode = True
REF = os.environ.get("MANYTOR_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

import numpy as np  # noqa: E402

import manytor as tor  # noqa: E402  (the reference)

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path)} bytes, keys={sorted(arrays)}")


def jc_of(goals):
    """joints_coordinates exactly as manytor.py:188-189 builds them."""
    jc = np.array([tor.fk(mode=i, goals=goals)[0:3, 3] for i in range(2, 5)])
    return np.vstack((np.zeros(3), jc))


# ---------------------------------------------------------------- F1 fk_kat
def gen_f1():
    rng = np.random.RandomState(101)
    fixed = [
        [0, 0, 0, 0], [30, 45, -60, 90], [90, 0, 0, 0], [0, 90, 0, 0], [0, 0, 90, 0], [0, 0, 0, 90],
        [-90, -90, -90, -90], [180, 180, 180, 180], [-180, -180, -180, -180], [0, 180, 0, 0],
        [179, -180, 179, -180], [45, 45, 45, 45], [1, 1, 1, 1], [-1, -1, -1, -1],
    ]
    ints = rng.randint(-180, 180, size=(121, 4)).astype(np.float64)
    frac = rng.uniform(-180, 180, size=(121, 4))
    angles = np.vstack([np.array(fixed, dtype=np.float64), ints, frac])
    mats = np.array([[tor.fk(mode, a) for mode in range(1, 5)] for a in angles])   # (M,4,4,4)
    save("f1_fk_kat", angles=angles, matrices=mats, positions=mats[:, :, 0:3, 3])


# ------------------------------------------------------ F2 single_env_trace
def gen_f2(seed=2, k=10, t=64):
    np.random.seed(seed)
    env = tor.Environment(k)
    obs0 = env.reset(returnable=True)
    points0 = env.points.copy()
    rec = {n: [] for n in ("action", "obs2", "reward", "done", "alives", "jc", "goals", "total", "points")}
    for _ in range(t):
        a = env.action_sample()
        obs2, reward, done = env.step(a)
        rec["action"].append(np.array(a, dtype=np.int64))
        rec["obs2"].append(obs2)
        rec["reward"].append(reward)
        rec["done"].append(done)
        rec["alives"].append(env.alives.copy())
        rec["jc"].append(env.joints_coordinates.copy())
        rec["goals"].append(np.array(env.goals, dtype=np.float64))
        rec["total"].append(env.total_reward)
        rec["points"].append(env.points.copy())
    save("f2_single_env_trace", seed=np.int64(seed), obj_number=np.int64(k), obs0=obs0, points0=points0,
         **{n: np.array(v) for n, v in rec.items()})


# ---------------------------------------------------------- F3 substep_trace
def gen_f3(pairs=32):
    rng = np.random.RandomState(303)
    prev = rng.randint(-180, 180, size=(pairs, 4)).astype(np.float64)
    act = rng.randint(-180, 180, size=(pairs, 4)).astype(np.float64)
    prev[0] = 0
    act[0] = [0, 180, 0, 0]
    prev[1] = [0, 180, 0, 0]
    act[1] = 0
    jcs = np.empty((pairs, 25, 4, 3))
    ground_sub = np.empty((pairs, 25), dtype=bool)
    reward = np.empty(pairs, dtype=np.int64)
    traj_ee = np.empty((pairs, 25, 3))
    for i in range(pairs):
        env = tor.Environment(1)
        np.random.seed(7)
        env.reset()
        env.points = np.array([[1000.0, 1000.0, 1000.0]])   # unreachable: reward isolates the ground flag
        env.goals = prev[i].copy()
        env.joints_coordinates = jc_of(env.goals)
        route = np.linspace(prev[i], act[i], num=25)
        for k in range(25):
            jcs[i, k] = jc_of(route[k])
            ground_sub[i, k] = (jcs[i, k, 2, 2] < 0) or (jcs[i, k, 3, 2] < 0)
        _, r, _ = env.step(list(act[i]))
        reward[i] = r
        traj_ee[i] = env.trajectory[-25:]
    assert np.array_equal(traj_ee, jcs[:, :, 3, :])
    assert np.array_equal(reward == -1, ground_sub.any(axis=1))
    save("f3_substep_trace", prev=prev, action=act, jc=jcs, ground_sub=ground_sub, reward=reward)


# --------------------------------------------------------- F4 multienv_trace
def gen_f4(seed=4, epochs=2, max_steps=50, shape=(3, 2), k=7):
    np.random.seed(seed)
    me = tor.Multienv(env_shape=shape, obj_number=k)
    n = shape[0] * shape[1]
    obs0 = np.array(me.reset(returnable=True))
    pts, acts, obs2s, rews, dones, alives, totals, jcs = [], [], [], [], [], [], [], []
    pts.append(np.array([e.points.copy() for e in me.environment]))
    never_broke = True
    for _ in range(epochs):
        for _ in range(max_steps):
            a = me.action_sample()
            o, r, d = me.step(a)
            if d == True:  # noqa: E712  (test_multi.py:22 compares a list with True)
                never_broke = False
            acts.append(np.array(a, dtype=np.int64))
            obs2s.append(np.array(o))
            rews.append(np.array(r, dtype=np.int64))
            dones.append(np.array(d, dtype=bool))
            alives.append(np.array([e.alives.copy() for e in me.environment]))
            jcs.append(np.array([e.joints_coordinates.copy() for e in me.environment]))
        totals.append(np.array([me.environment[i].total_reward for i in range(n)]))
        me.reset()
        pts.append(np.array([e.points.copy() for e in me.environment]))
    save("f4_multienv_trace", seed=np.int64(seed), env_shape=np.array(shape), obj_number=np.int64(k),
         max_steps=np.int64(max_steps), obs0=obs0, points=np.array(pts), action=np.array(acts),
         obs2=np.array(obs2s), reward=np.array(rews), done=np.array(dones), alives=np.array(alives),
         jc=np.array(jcs), total_reward=np.array(totals), never_broke=np.bool_(never_broke))


# ----------------------------------------------------------- F5 semantics_kat
def _run_scenario(points, actions):
    """Fresh env, hand-placed targets, list of actions -> per-step records."""
    pts = np.array(points, dtype=np.float64)
    env = tor.Environment(len(pts))
    np.random.seed(5)
    env.reset()
    env.points = pts.copy()
    rec = {"obs2": [], "reward": [], "done": [], "alives": [], "points": [], "jc": [], "total": []}
    for a in actions:
        o, r, d = env.step(list(a))
        rec["obs2"].append(o)
        rec["reward"].append(r)
        rec["done"].append(d)
        rec["alives"].append(env.alives.copy())
        rec["points"].append(env.points.copy())
        rec["jc"].append(env.joints_coordinates.copy())
        rec["total"].append(env.total_reward)
    return {k: np.array(v) for k, v in rec.items()}


def gen_f5():
    ee = tor.fk(4, [30, 45, -60, 90])[0:3, 3]          # end effector at the KAT pose
    scen = {
        # two targets inside the 8.0 box of the EE + one far: reward 1 once, obs2 shows them this step, zeros next
        "multi_pickup": (
            [ee + [1.0, -2.0, 3.0], ee + [-7.5, 7.5, 0.0], [-40.0, 5.0, 10.0]],
            [[30, 45, -60, 90], [30, 45, -60, 90], [31, 45, -60, 90]],
        ),
        # ground hit and carry-over: -1, -1 (sub-step 0 is the previous pose), 0
        "ground_carry": ([[40.0, 0.0, 30.0]], [[0, 180, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0]]),
        # single target picked -> (reward 1, done True); further steps stay done, reward 0
        "all_picked": ([ee + [0.5, 0.5, 0.5]], [[30, 45, -60, 90], [30, 45, -60, 90], [10, 10, 10, 10]]),
        # exactly-on-threshold axis (|delta| = 8.0 exactly is inside; 8.000001 outside)
        "threshold": (
            [[0.0, 0.0, 55.6 - 8.0], [8.0, -8.0, 55.6], [8.000001, 0.0, 55.6]],
            [[0, 0, 0, 0]],
        ),
    }
    # pickup and ground in the same step: a (previous pose, action) pair, both well above ground, whose
    # interpolated route dips below z=0 (pair found by a vectorised search; the asserts below check it on
    # the reference itself).  From the zero pose no such single action exists (200k random draws, none).
    prev_pose = [141, -53, -108, -155]
    dip_action = [-12, -31, 176, -122]
    ee2 = tor.fk(4, [float(v) for v in dip_action])[0:3, 3]
    scen["pickup_and_ground"] = ([ee2 + [1.0, 1.0, -1.0], [-30.0, -30.0, 5.0]], [prev_pose, dip_action, dip_action])
    out = {}
    for name, (pts, acts) in scen.items():
        rec = _run_scenario(pts, acts)
        out[f"{name}__points_in"] = np.array(pts, dtype=np.float64)
        out[f"{name}__actions"] = np.array(acts, dtype=np.float64)
        for k, v in rec.items():
            out[f"{name}__{k}"] = v
    # the semantics the survey lists, asserted on the reference's own outputs
    assert list(out["multi_pickup__reward"]) == [1, 0, 0]
    assert list(out["ground_carry__reward"]) == [-1, -1, 0]
    assert list(out["all_picked__reward"]) == [1, 0, 0] and list(out["all_picked__done"]) == [True, True, True]
    assert out["pickup_and_ground__alives"][0][0] and out["pickup_and_ground__jc"][0][2:, 2].min() > 5
    assert out["pickup_and_ground__reward"][1] == -1 and not out["pickup_and_ground__alives"][1][0]
    assert out["pickup_and_ground__jc"][1][2:, 2].min() > 5
    assert list(out["threshold__alives"][0]) == [False, False, True]
    save("f5_semantics_kat", **out)


# -------------------------------------------------------------- F6 rng_streams
def gen_f6(seed=6):
    # R1: reference reset draws for 5 envs x K=7, then R2: action draws for 5 envs, interleaved like test_multi.py
    np.random.seed(seed)
    me = tor.Multienv(env_shape=(5, 1), obj_number=7)
    me.reset()
    pts_a = np.array([e.points.copy() for e in me.environment])
    act_a = np.array(me.action_sample(), dtype=np.int64)
    act_b = np.array(me.action_sample(), dtype=np.int64)
    me.reset()
    pts_b = np.array([e.points.copy() for e in me.environment])
    act_c = np.array(me.action_sample(), dtype=np.int64)
    tail = np.random.random_sample(4)      # pins the stream position after all of the above
    save("f6_rng_streams", seed=np.int64(seed), points_a=pts_a, actions_a=act_a, actions_b=act_b,
         points_b=pts_b, actions_c=act_c, tail=tail)


# ------------------------------------------------------------------ F7 dh7_kat
DH7_TABLE = np.array(
    [
        [0.0, -np.pi / 2, 34.0, 0.0],
        [0.0, np.pi / 2, 0.0, 0.0],
        [4.5, np.pi / 2, 40.0, 0.0],
        [-4.5, -np.pi / 2, 0.0, 0.0],
        [0.0, -np.pi / 2, 40.0, 0.0],
        [8.8, np.pi / 2, 0.0, -np.pi / 2],
        [0.0, 0.0, 12.6, 0.0],
    ]
)


def gen_f7(m=128):
    rng = np.random.RandomState(707)
    angles = np.vstack([
        np.zeros((1, 7)), rng.randint(-180, 180, size=(m // 2 - 1, 7)).astype(np.float64),
        rng.uniform(-180, 180, size=(m // 2, 7)),
    ])
    mats = np.empty((len(angles), 7, 4, 4))
    for i, a in enumerate(angles):
        cur = np.eye(4)
        for j in range(7):
            aa, al, d, off = DH7_TABLE[j]
            cur = cur.dot(tor.dh(aa, al, d, np.radians(a[j]) + off))    # the reference's dh(), manytor.py:25-32
            mats[i, j] = cur
    save("f7_dh7_kat", table=DH7_TABLE, angles=angles, matrices=mats, positions=mats[:, :, 0:3, 3])


# ---------------------------------------------------------- F8 dh / r_theta KAT
def gen_f8(m=64):
    rng = np.random.RandomState(808)
    params = np.column_stack([
        rng.uniform(-30, 30, m), rng.choice([-np.pi / 2, 0.0, np.pi / 2, 0.3, -1.1], m),
        rng.uniform(-30, 30, m), rng.uniform(-np.pi, np.pi, m),
    ])
    dh_mats = np.array([tor.dh(*p) for p in params])
    v1 = rng.uniform(-50, 50, size=(m, 3))
    v2 = rng.uniform(-50, 50, size=(m, 3))
    v2[0] = v1[0]                       # atan2(0,0) = 0 case
    v2[1, 0:2] = v1[1, 0:2]             # bearing 0/0, elevation 0
    rt = np.array([tor.r_theta(a, b) for a, b in zip(v1, v2)])
    save("f8_dh_rtheta_kat", dh_params=params, dh_matrices=dh_mats, v1=v1, v2=v2, r_theta=rt)


# ------------------------------------------------------------- F9 viewer datagrams
class _Capture:
    """Stands in for the UDP socket of a rendering Environment (manytor.py:265-267): keeps what would be sent."""

    def __init__(self):
        self.msgs = []

    def sendto(self, msg, dest):
        self.msgs.append(bytes(msg))


def gen_f9(seed=9, k=7, steps=2):
    """The frames manytor.py:194-201 sends per sub-step and the clear message of :246-249, as parsed numbers.
    render() itself is never called (it would spawn the vispy viewer): only the flag and the socket are set."""
    import json
    np.random.seed(seed)
    env = tor.Environment(k, index=3)
    env.reset()
    cap = _Capture()
    env.rendering, env.udp, env.dest = True, cap, None
    actions, frames = [], []
    for _ in range(steps):
        a = env.action_sample()
        env.step(a)
        actions.append(np.array(a, dtype=np.int64))
    frames = np.array([json.loads(m) for m in cap.msgs], dtype=np.float64)     # (steps*25, 3*(1+4+K+1)); NaN kept
    lengths = np.array([len(m) for m in cap.msgs], dtype=np.int64)
    points = env.points.copy()
    cap.msgs = []
    env.reset()                                                                  # rendering: sends [nan, nan, 4]
    clear = np.array(json.loads(cap.msgs[0]), dtype=np.float64)
    assert len(cap.msgs) == 1 and frames.shape == (steps * 25, 3 * (1 + 4 + k + 1))
    save("f9_viewer_frames", seed=np.int64(seed), obj_number=np.int64(k), env_id=np.int64(3), action=np.array(actions),
         frames=frames, datagram_bytes=lengths, points=points, clear=clear)


if __name__ == "__main__":
    gen_f9()
    gen_f1()
    gen_f2()
    gen_f3()
    gen_f4()
    gen_f5()
    gen_f6()
    gen_f7()
    gen_f8()
