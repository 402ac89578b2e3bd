"""GPU parity tests: the HIP path (through the C ABI / ctypes) against the golden fixtures produced by
the reference and against the CPU oracle on seeded inputs.

Tolerances (fp32 device arithmetic vs the reference's fp64):
  positions            atol 1e-4
  distance             atol 2e-4
  angles (degrees)     atol max(1e-3, 57.3 * 1e-4 / rho)   rho = lever arm of the angle
  reward/done/alives   exact, except where the oracle's own margin to the z = 0 or |delta| = tol threshold is
                       below GUARD = 1e-3 (fp32 and fp64 may legitimately land on different sides); guarded
                       cases are counted and must stay rare.
Bit-exact: device RNG streams vs oracle/philox_ref.py, staged actions, goals, sharding invariance.
"""
import numpy as np
import pytest

from parity_util import DIST_TOL, GUARD, POS_TOL, assert_obs_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def m():
    import manytor_amd
    if manytor_amd.device_count() < 1:
        pytest.fail("gpu tests need a visible MI355X and the in-tree libmanytor_hip.so")
    return manytor_amd


@pytest.fixture(scope="module")
def mo():
    from oracle import manytor_oracle
    return manytor_oracle


class Lockstep:
    """Drive a StepEngine and a BatchOracle with the same inputs; compare every output every step."""

    def __init__(self, m, mo, n, k, table=None, substeps=25, obs_frame=-2, ee_frame=-1, **kw):
        table = m.REF_DH_TABLE if table is None else table
        self.fo, self.fe = obs_frame, ee_frame
        self.eng = m.StepEngine(n, k, dh_table=table, substeps=substeps, obs_frame=obs_frame, ee_frame=ee_frame, **kw)
        self.ora = mo.BatchOracle(n, k, table=np.asarray(table), substeps=substeps, obs_frame=obs_frame, ee_frame=ee_frame)
        self.n, self.k = n, k
        self.valid = np.ones(n, dtype=bool)      # envs whose discrete state is still comparable
        self.guarded = 0
        self.compared = 0

    def reset(self, points):
        p32 = np.asarray(points, dtype=np.float32).reshape(self.n, self.k, 3)
        self.eng.reset(p32)
        self.ora.reset(p32.astype(np.float64))
        self.valid[:] = True
        np.testing.assert_array_equal(self.eng.points(), p32)
        assert np.all(self.eng.goals() == 0) and np.all(self.eng.total_reward() == 0) and self.eng.alives().all()
        np.testing.assert_allclose(self.eng.joints_coordinates(), self.ora.joints_coordinates, atol=POS_TOL)
        np.testing.assert_allclose(self.eng.ee(), self.ora.joints_coordinates[:, self.fe], atol=POS_TOL)

    def step(self, actions):
        e, o = self.eng, self.ora
        pre_alive = o.alives.copy()
        e.step(actions)
        obs_ref, rew_ref, done_ref = o.step(actions)
        pre_points = o.points.copy()              # after the zeroing of dead targets, before pickup bookkeeping
        v = self.valid
        # continuous outputs
        np.testing.assert_array_equal(e.goals(), np.asarray(actions, dtype=np.float32).reshape(self.n, -1))
        jc = e.joints_coordinates()
        assert np.abs(jc - o.joints_coordinates).max() <= POS_TOL
        assert np.abs(e.ee() - o.joints_coordinates[:, self.fe]).max() <= POS_TOL
        if v.any():
            assert_obs_close(e.obs()[v], obs_ref[v], o.joints_coordinates[v, self.fo], pre_points[v], pre_alive[v])
        # discrete outputs under the guard band
        pick_m = np.where(pre_alive, o.pickup_margin, np.inf).min(axis=1)
        risky = (o.ground_margin < GUARD) | (pick_m < GUARD)
        ok = v & ~risky
        self.guarded += int((v & risky).sum())
        self.compared += int(ok.sum())
        np.testing.assert_array_equal(e.reward()[ok], rew_ref[ok])
        np.testing.assert_array_equal(e.done()[ok], done_ref[ok])
        np.testing.assert_array_equal(e.alives()[ok], o.alives[ok])
        np.testing.assert_array_equal(e.total_reward()[ok], o.total_reward[ok].astype(np.float32))
        # an env that was inside the guard band may have diverged legitimately: re-sync its discrete state
        if (v & risky).any():
            idx = np.flatnonzero(v & risky)
            alive = e.alives()
            o.alives[idx] = alive[idx]
            o.total_reward[idx] = e.total_reward()[idx]
            pts = e.points().astype(np.float64)
            o.points[idx] = pts[idx]
        # done bits are the wavefront ballot of the done bytes
        done = e.done()
        bits = e.done_bits()
        unpacked = ((bits[:, None] >> np.arange(64, dtype=np.uint64)) & np.uint64(1)).astype(bool).ravel()[: self.n]
        np.testing.assert_array_equal(unpacked, done)
        return obs_ref, rew_ref, done_ref


# --------------------------------------------------------------------------- L1 kinematics (F1, F7, F8)
def test_f1_fk_positions_and_matrices(m, golden):
    g = golden("f1_fk_kat")
    for mode in range(1, 5):
        mats = m.fk_batch(mode, g["angles"])
        assert np.abs(mats - g["matrices"][:, mode - 1]).max() <= POS_TOL
    out = m.fk(4, [30, 45, -60, 90])
    assert out.shape == (4, 4) and out.dtype == np.float64
    np.testing.assert_allclose(out[0:3, 3], [34.839021, -6.885682, 11.936753], atol=POS_TOL)
    np.testing.assert_allclose(m.fk(4, [0, 0, 0, 0])[0:3, 3], [0, 0, 55.6], atol=1e-5)
    for mode, z in ((1, 4.3), (2, 4.3), (3, 28.6), (4, 55.6)):
        np.testing.assert_allclose(m.fk(mode, [0, 0, 0, 0])[0:3, 3], [0, 0, z], atol=1e-5)


def test_f8_dh_and_r_theta(m, golden):
    g = golden("f8_dh_rtheta_kat")
    for p, ref in zip(g["dh_params"], g["dh_matrices"]):
        assert np.abs(m.dh(*p) - ref).max() <= 2e-5
    rt = m.r_theta_batch(g["v1"], g["v2"])
    d = np.abs(g["v1"] - g["v2"])
    rho = np.hypot(d[:, 0], d[:, 1])
    full = np.sqrt(rho ** 2 + d[:, 2] ** 2)
    with np.errstate(divide="ignore"):
        assert np.all(np.abs(rt[:, 0] - g["r_theta"][:, 0]) <= np.maximum(1e-3, 57.3 * 1e-4 / rho))
        assert np.all(np.abs(rt[:, 1] - g["r_theta"][:, 1]) <= np.maximum(1e-3, 57.3 * 1e-4 / full))
    assert m.r_theta([1, 2, 3], [1, 2, 3]) == (0.0, 0.0)          # atan2(0, 0) = 0 like math.atan2


def test_f7_seven_dof_chain(m, golden):
    g = golden("f7_dh7_kat")
    n = len(g["angles"])
    eng = m.StepEngine(n, 3, dh_table=g["table"], radius=92.6)
    eng.reset(np.zeros((n, 3, 3), dtype=np.float32))
    eng.set(m.lib.F_GOALS, g["angles"].astype(np.float32))
    jc = eng.joints_coordinates()
    ref = g["positions"].copy()
    ref[:, 0] = 0.0
    a32 = g["angles"].astype(np.float32).astype(np.float64)       # compare at the fp32-rounded input angles
    from oracle import manytor_oracle as mo
    ref32 = mo.batch_joints_coordinates(a32, g["table"])
    assert np.abs(jc - ref32).max() <= POS_TOL
    assert np.abs(jc[: n // 2] - ref[: n // 2]).max() <= POS_TOL  # integer-degree half: no input rounding at all
    mats = m.fk_batch(7, g["angles"], dh_table=g["table"])
    assert np.abs(mats[: n // 2] - g["matrices"][: n // 2, 6]).max() <= POS_TOL


def route_ground_margin(mo, prev, action, substeps=25):
    """min |z| of the last two frames over the sub-step poses of the routes prev[i] -> action[i] (manytor.py:182-192),
    from the oracle: how far the reference's ground-flag decision of that step was from flipping."""
    prev, action = np.atleast_2d(np.asarray(prev, dtype=np.float64)), np.atleast_2d(np.asarray(action, dtype=np.float64))
    route = np.linspace(prev, action, substeps)                      # (S, n, D)
    jc = mo.batch_joints_coordinates(route.reshape(-1, route.shape[-1])).reshape(substeps, len(prev), -1, 3)
    return np.abs(jc[:, :, 2:, 2]).min(axis=(0, 2))


# --------------------------------------------------------------------------- single env (F2), drop-in surface
def test_f2_single_env_trace_through_environment_class(m, mo, golden):
    g = golden("f2_single_env_trace")
    k = int(g["obj_number"])
    np.random.seed(int(g["seed"]))
    env = m.Environment(k)
    obs0 = env.reset(returnable=True)
    np.testing.assert_array_equal(env.points, g["points0"].astype(np.float32))   # same draws as the reference
    assert obs0.shape == (3 * k,) and obs0.dtype == np.float64
    zero_elbow = mo.batch_joints_coordinates(np.zeros((1, 4)))[:, 2]
    assert_obs_close(obs0[None], g["obs0"][None], zero_elbow, g["points0"][None], np.ones((1, k), dtype=bool))
    diverged = False
    for t in range(len(g["action"])):
        a = env.action_sample()
        assert [int(v) for v in a] == list(g["action"][t])                        # R2 stream, bit-identical
        obs2, reward, done = env.step(a)
        assert isinstance(reward, int) and isinstance(done, bool)
        jc = env.joints_coordinates
        assert np.abs(jc - g["jc"][t]).max() <= POS_TOL
        np.testing.assert_array_equal(env.goals, g["goals"][t])
        if diverged:
            continue
        alive_before = g["alives"][t - 1] if t else np.ones(k, dtype=bool)
        # distance, bearing and elevation of every target, angles with the lever-arm-scaled tolerance
        assert_obs_close(obs2[None], g["obs2"][t][None], g["jc"][t][2][None], g["points"][t][None], alive_before[None])
        # discrete outputs: guard on the fixture's own margins (sub-step poses replayed through the oracle)
        zmin = route_ground_margin(mo, g["goals"][t - 1] if t else np.zeros(4), g["action"][t])[0]
        dl = np.abs(g["jc"][t][3][None, :] - g["points"][t])
        pm = np.abs(dl - 8.0)[alive_before].min() if alive_before.any() else np.inf
        if pm < GUARD:
            diverged = True
            continue
        np.testing.assert_array_equal(env.alives, g["alives"][t])
        assert done == bool(g["done"][t])
        if zmin >= GUARD:
            assert reward == int(g["reward"][t])
            assert env.total_reward == float(g["total"][t])
    assert not diverged or t > 10


def test_survey_kat_observation_after_seeded_reset(m):
    """SURVEY.md 8(a) A4, measured on the reference: np.random.seed(0); Environment(3).reset(returnable=True)."""
    np.random.seed(0)
    obs = m.Environment(3).reset(returnable=True)
    ref = [28.958179, 12.780675, 51.425159, 16.382115, 30.451181, 33.686785, 41.183154, 21.779136, 51.503084]
    np.testing.assert_allclose(obs, ref, atol=2e-4)


def test_f3_substep_ground_flag(m, mo, golden):
    g = golden("f3_substep_trace")
    n = len(g["prev"])
    eng = m.StepEngine(n, 1)
    eng.reset(np.full((n, 1, 3), 1000.0, dtype=np.float32))
    eng.set(m.lib.F_GOALS, g["prev"].astype(np.float32))
    eng.step(g["action"])
    zmargin = np.abs(g["jc"][:, :, 2:, 2]).reshape(n, -1).min(axis=1)
    ok = zmargin >= GUARD
    assert ok.sum() >= n - 2
    np.testing.assert_array_equal(eng.reward()[ok], g["reward"][ok])
    assert np.abs(eng.joints_coordinates() - g["jc"][:, -1]).max() <= POS_TOL
    assert not eng.done().any()


def test_f4_multienv_trace_drop_in(m, mo, golden):
    """The loop of test_multi.py:11-34 on the HIP engine, seeded like the fixture: same targets, same actions,
    and outputs within tolerance for all 2 x 50 steps of 6 envs."""
    g = golden("f4_multienv_trace")
    shape = tuple(int(v) for v in g["env_shape"])
    n, k, steps = shape[0] * shape[1], int(g["obj_number"]), int(g["max_steps"])
    np.random.seed(int(g["seed"]))
    me = m.Multienv(env_shape=shape, obj_number=k)
    obs = me.reset(returnable=True)
    assert isinstance(obs, list) and len(obs) == n and obs[0].shape == (3 * k,)
    zero_elbow = np.repeat(mo.batch_joints_coordinates(np.zeros((1, 4)))[:, 2], n, axis=0)
    assert_obs_close(np.array(obs), g["obs0"], zero_elbow, g["points"][0], np.ones((n, k), dtype=bool))
    t = 0
    valid = np.ones(n, dtype=bool)
    for ep in range(len(g["total_reward"])):
        np.testing.assert_array_equal(np.array([e.points for e in me.environment]), g["points"][ep].astype(np.float32))
        for _ in range(steps):
            action = me.action_sample()
            assert isinstance(action, list) and isinstance(action[0], list)
            np.testing.assert_array_equal(np.array(action), g["action"][t])
            obs2, reward, done = me.step(action)
            assert isinstance(obs2, list) and isinstance(reward, list) and isinstance(done, list)
            assert not (done == True)  # noqa: E712  test_multi.py:22: a list never equals True, the loop never breaks
            jc = np.array([e.joints_coordinates for e in me.environment])
            assert np.abs(jc - g["jc"][t]).max() <= POS_TOL
            alive_before = g["alives"][t - 1] if t % steps else np.ones((n, k), dtype=bool)
            dl = np.abs(g["jc"][t][:, 3][:, None, :] - _points_at(g, ep, t, steps))
            pm = np.where(alive_before[..., None], np.abs(dl - 8.0), np.inf).reshape(n, -1).min(axis=1)
            valid &= pm >= GUARD
            np.testing.assert_array_equal(np.array([e.alives for e in me.environment])[valid], g["alives"][t][valid])
            np.testing.assert_array_equal(np.array(done)[valid], g["done"][t][valid])
            o = np.array(obs2)
            pts_t = _points_at(g, ep, t, steps)
            assert_obs_close(o[valid], g["obs2"][t][valid], g["jc"][t][valid, 2], pts_t[valid], alive_before[valid])
            # per-step reward, exact wherever the reference's own ground-flag margin of this route is outside the band
            prev = g["action"][t - 1].astype(np.float64) if t % steps else np.zeros((n, 4))
            zok = route_ground_margin(mo, prev, g["action"][t]) >= GUARD
            np.testing.assert_array_equal(np.array(reward)[valid & zok], g["reward"][t][valid & zok])
            assert (valid & zok).sum() >= n - 1
            t += 1
        totals = np.array([me.environment[i].total_reward for i in range(n)])
        # returns can differ only through guarded threshold cases; on this fixture there are none
        np.testing.assert_array_equal(totals[valid], g["total_reward"][ep][valid])
        me.reset()
        valid[:] = True
    assert t == 2 * steps


def _points_at(g, ep, t, steps):
    """Targets of the fixture as they were at step t: dead targets are zeroed from the step after their pickup."""
    pts = g["points"][ep].copy()
    first = ep * steps
    if t > first:
        dead_prev = ~g["alives"][t - 1]
        pts[dead_prev] = 0.0
    return pts


@pytest.mark.parametrize("name", ["multi_pickup", "ground_carry", "all_picked", "pickup_and_ground"])
def test_f5_semantics(m, golden, name):
    g = golden("f5_semantics_kat")
    pts, acts = g[f"{name}__points_in"], g[f"{name}__actions"]
    env = m.Environment(len(pts))
    np.random.seed(0)
    env.reset()
    env.points = pts
    for t, a in enumerate(acts):
        obs2, r, d = env.step(list(a))
        assert r == int(g[f"{name}__reward"][t]) and d == bool(g[f"{name}__done"][t])
        np.testing.assert_array_equal(env.alives, g[f"{name}__alives"][t])
        np.testing.assert_allclose(env.points, g[f"{name}__points"][t], atol=1e-5)   # incl. the zeroing of dead targets
        alive_before = g[f"{name}__alives"][t - 1] if t else np.ones(len(pts), dtype=bool)
        assert_obs_close(obs2[None], g[f"{name}__obs2"][t][None], g[f"{name}__jc"][t][2][None],
                         g[f"{name}__points"][t][None], alive_before[None])
        assert env.total_reward == float(g[f"{name}__total"][t])
        assert np.abs(env.joints_coordinates - g[f"{name}__jc"][t]).max() <= POS_TOL


def test_f5_exact_threshold_case(m, golden):
    """|delta| exactly 8.0 counts as reached (math.isclose is <=, manytor.py:162); 8.000001 does not.  Target 0 of
    the scenario sits within one fp64 ulp of the threshold, i.e. inside the guard band: not compared."""
    g = golden("f5_semantics_kat")
    pts = g["threshold__points_in"]
    env = m.Environment(len(pts))
    np.random.seed(0)
    env.reset()
    env.points = pts
    env.step([0, 0, 0, 0])
    np.testing.assert_array_equal(env.alives[1:], g["threshold__alives"][0][1:])
    assert list(g["threshold__alives"][0][1:]) == [False, True]


def test_environment_piecewise_methods(m, golden):
    """get_observations / is_done / action as separate calls (manytor.py:141,155,175)."""
    g = golden("f5_semantics_kat")
    pts = g["all_picked__points_in"]
    env = m.Environment(len(pts))
    np.random.seed(0)
    env.reset()
    env.points = pts
    assert env.is_done() is False
    reward, obs2 = env.action([30, 45, -60, 90], None)
    assert reward == 1 and env.total_reward == 0.0            # action() does not accumulate (manytor.py:258 is in step)
    assert env.is_done() is True
    np.testing.assert_array_equal(env.get_obs(), np.zeros(3))
    np.testing.assert_array_equal(env.points, np.zeros((1, 3)))


# --------------------------------------------------------------------------- batches vs the oracle
@pytest.mark.parametrize("n,k,steps", [(1, 10, 30), (63, 7, 12), (257, 1, 12), (4096, 7, 25), (1000, 32, 6)])
def test_random_batches_lockstep_reference_arm(m, mo, n, k, steps):
    rng = np.random.RandomState(1000 + n)
    ls = Lockstep(m, mo, n, k)
    from oracle import philox_ref as px
    ls.reset(px.sample_targets(11, np.arange(n, dtype=np.uint64), 0, k, 51.3))
    for t in range(steps):
        ls.step(rng.randint(-180, 180, size=(n, 4)))
    assert ls.guarded <= max(2, 0.01 * n * steps), ls.guarded
    assert ls.compared > 0


def test_random_batch_float_actions_and_small_moves(m, mo):
    """Fractional degrees (goals stop being integers, manytor.py:184) and tiny moves around the pickup box."""
    n, k = 2048, 7
    rng = np.random.RandomState(77)
    ls = Lockstep(m, mo, n, k)
    ls.reset(rng.uniform(-30, 30, size=(n, k, 3)) + [0, 0, 35])
    for t in range(10):
        a = rng.uniform(-60, 60, size=(n, 4)).astype(np.float32)
        ls.step(a.astype(np.float64))
    assert ls.guarded <= 0.01 * n * 10


def test_seven_dof_lockstep(m, mo, golden):
    table = golden("f7_dh7_kat")["table"]
    n, k = 2048, 7
    rng = np.random.RandomState(7)
    ls = Lockstep(m, mo, n, k, table=table, radius=92.6)
    pts = rng.uniform(-60, 60, size=(n, k, 3))
    pts[..., 2] = np.abs(pts[..., 2])
    ls.reset(pts)
    for t in range(8):
        ls.step(rng.randint(-180, 180, size=(n, 7)))
    assert ls.guarded <= 0.01 * n * 8


@pytest.mark.parametrize("dof", [2, 3, 5, 8])
def test_other_joint_counts(m, mo, dof):
    rng = np.random.RandomState(dof)
    table = np.column_stack([rng.uniform(0, 10, dof), rng.choice([-np.pi / 2, 0, np.pi / 2, 0.4], dof),
                             rng.uniform(2, 15, dof), rng.choice([0, -np.pi / 2, 0.25], dof)])
    ls = Lockstep(m, mo, 512, 4, table=table)
    ls.reset(rng.uniform(-20, 20, size=(512, 4, 3)))
    for t in range(5):
        ls.step(rng.randint(-180, 180, size=(512, dof)))


@pytest.mark.parametrize("dof,obs_frame,ee_frame", [(4, 1, -1), (4, -1, -1), (5, -1, 2), (7, 0, -3), (3, -2, -2), (8, 3, 6)])
def test_selectable_observation_and_pickup_frames(m, mo, dof, obs_frame, ee_frame):
    """the rows of joints_coordinates the reference hard-codes -- observation from [2]
    (manytor.py:143), pickup from [3] (:162), ground test on both (:191) -- as constructor arguments (Python indexing,
    row 0 = the origin), in lock step with the oracle given the same rows; fused requests fall back to per-step launches."""
    rng = np.random.RandomState(100 * dof + obs_frame + 10 * ee_frame)
    if dof == 4:
        table = np.array(m.REF_DH_TABLE)
    else:
        table = np.column_stack([rng.uniform(0, 10, dof), rng.choice([-np.pi / 2, 0, np.pi / 2, 0.4], dof),
                                 rng.uniform(2, 15, dof), rng.choice([0, -np.pi / 2, 0.25], dof)])
    n = 777
    ls = Lockstep(m, mo, n, 5, table=table, obs_frame=obs_frame, ee_frame=ee_frame)
    assert "RtTableF" in ls.eng.step_kernel_name()
    ls.reset(rng.uniform(-25, 25, size=(n, 5, 3)))
    for t in range(6):
        ls.step(rng.randint(-180, 180, size=(n, dof)))
    # row 0 is the origin: its z is exactly 0, i.e. ON the ground-test threshold, so with it every env sits inside the
    # guard band by construction and only the continuous outputs are compared (the device says z < 0 is false, like numpy)
    assert ls.compared > 0 or 0 in (obs_frame % dof, ee_frame % dof)
    # the standalone passes and the random-action / fused entry points use the same rows
    a, b = ls.eng, m.StepEngine(n, 5, dh_table=table, obs_frame=obs_frame, ee_frame=ee_frame)
    b.set_state(a.get_state())
    a.rollout(4, 3, 0)
    b.rollout_fused(4, 3, 0)
    for f in ("F_GOALS", "F_OBS", "F_REWARD", "F_DONE", "F_ALIVE", "F_EE", "F_TOTAL_REWARD"):
        np.testing.assert_array_equal(a.get(getattr(m.lib, f)), b.get(getattr(m.lib, f)), err_msg=f)
    obs_step = a.obs().copy()
    a.observe()
    alive = a.alives()
    stay = np.repeat(alive, 3, axis=1)
    np.testing.assert_allclose(a.obs()[stay], obs_step[stay], atol=2e-3)


def test_default_frames_are_the_reference_rows(m):
    eng = m.StepEngine(100, 3, obs_frame=2, ee_frame=3)          # the reference's literal rows, non-negative spelling
    assert "RtTableF" not in eng.step_kernel_name() and "Ref4Table" in eng.step_kernel_name() + "Ref4Table"
    with pytest.raises(ValueError):
        m.StepEngine(10, 3, obs_frame=4)
    with pytest.raises(ValueError):
        m.StepEngine(10, 3, ee_frame=-5)


def test_substeps_and_tolerance_parameters(m, mo):
    rng = np.random.RandomState(5)
    ls = Lockstep(m, mo, 512, 5, substeps=2, pickup_tol=3.0)
    ls.ora.pickup_tol = 3.0
    ls.reset(rng.uniform(-40, 40, size=(512, 5, 3)))
    for t in range(5):
        ls.step(rng.randint(-180, 180, size=(512, 4)))


@pytest.mark.parametrize("kw", [dict(hw_trig=True), dict(dh_in_lds=True), dict(direct_trig=True), dict(specialize=False),
                                dict(specialize=False, direct_trig=True), dict(specialize=False, hw_trig=True)])
def test_kernel_variants_agree_with_oracle(m, mo, kw):
    rng = np.random.RandomState(9)
    ls = Lockstep(m, mo, 4096, 7, **kw)
    from oracle import philox_ref as px
    ls.reset(px.sample_targets(3, np.arange(4096, dtype=np.uint64), 0, 7, 51.3))
    for t in range(10):
        ls.step(rng.randint(-180, 180, size=(4096, 4)))
    assert ls.guarded <= 0.01 * 4096 * 10


def test_terminate_on_ground_flag(m):
    eng = m.StepEngine(2, 1, terminate_on_ground=True)
    eng.reset(np.full((2, 1, 3), 1000.0, dtype=np.float32))
    eng.step([[0, 180, 0, 0], [0, 10, 0, 0]])
    np.testing.assert_array_equal(eng.reward(), [-1, 0])
    np.testing.assert_array_equal(eng.ground_hit(), [True, False])
    np.testing.assert_array_equal(eng.done(), [True, False])


# --------------------------------------------------------------------------- device RNG (bit-exact)
def test_device_rng_streams_bit_exact(m):
    from oracle import philox_ref as px
    n, k, base = 5000, 7, 123456789012
    eng = m.StepEngine(n, k, env_id_base=base)
    ids = np.arange(base, base + n, dtype=np.uint64)
    eng.reset_random(seed=0xABCDEF0123, episode=3)
    np.testing.assert_array_equal(eng.points(), px.sample_targets(0xABCDEF0123, ids, 3, k, 51.3))
    eng.sample_actions(seed=42, step_idx=17)
    np.testing.assert_array_equal(eng.actions(), px.sample_actions(42, ids, 17, 4))
    e7 = m.StepEngine(300, 2, dh_table=m.DH7_TABLE, radius=92.6)
    e7.sample_actions(seed=42, step_idx=5)
    np.testing.assert_array_equal(e7.actions(), px.sample_actions(42, np.arange(300, dtype=np.uint64), 5, 7))


@pytest.mark.parametrize("split", [0, 1])
@pytest.mark.parametrize("k", [1, 21, 22, 32])
def test_device_target_draw_every_k(m, k, split, monkeypatch):
    """reset_kernel parks the accepted targets in a [3K][256] LDS tile before writing them out: 21 -> 22 targets
    crosses the 64 KiB default limit for dynamic LDS, 32 is the largest K (96 KiB).  Ragged last block, and the re-arm
    of finished envs only (mt_reset_done) through the same tile.  split = 1: reset_split_kernel (an env's draw over 4
    lanes, what small batches get by default), split = 0: one env per lane."""
    from oracle import philox_ref as px
    monkeypatch.setenv("MT_RESET_SPLIT", str(split))
    n = 1000 + k
    ids = np.arange(n, dtype=np.uint64)
    eng = m.StepEngine(n, k, dh_table=m.DH7_TABLE if k == 22 else m.REF_DH_TABLE, radius=51.3)
    eng.reset_random(seed=77, episode=2)
    np.testing.assert_array_equal(eng.points(), px.sample_targets(77, ids, 2, k, 51.3))
    # finish every third env by hand (all targets picked), then re-arm only those
    before = eng.points().copy()
    fin = np.arange(n) % 3 == 0
    eng.set(m.lib.F_ALIVE, np.where(fin[:, None], False, eng.alives()).astype(np.uint8))
    eng.check_done()                    # may also pick a target sitting at the zero-pose end-effector (K = 1: finishes)
    fin2 = eng.done()
    assert fin2[fin].all() and fin2.sum() < fin.sum() + n // 50
    eng.reset_done(seed=77)
    expect = before.copy()
    expect[fin2] = px.sample_targets(77, ids[fin2], 3, k, 51.3)     # their own episode counter advanced 2 -> 3
    np.testing.assert_array_equal(eng.points(), expect)
    assert not eng.done().any() and eng.alives()[fin2].all()
    np.testing.assert_array_equal(eng.episodes(), np.where(fin2, 3, 2))


@pytest.mark.parametrize("n", [1, 63, 4097, 70000])
def test_reset_split_kernel_equals_reset_kernel(m, n, monkeypatch):
    """Every field after mt_reset_random and after mt_reset_done (ring, last return, episode counters, done bits
    included) is the same whether the draw runs one env per lane or over 4 lanes per env."""
    fields = ("F_GOALS", "F_POINTS", "F_ALIVE", "F_TOTAL_REWARD", "F_REWARD", "F_DONE", "F_DONE_BITS", "F_EE",
              "F_EPISODES", "F_LAST_RETURN", "F_RETURN_RING")
    outs = []
    for split in (0, 1):
        monkeypatch.setenv("MT_RESET_SPLIT", str(split))
        eng = m.StepEngine(n, 5, pickup_tol=30.0, return_ring=3)
        eng.reset_random(21, 4)
        snap = [{f: eng.get(getattr(m.lib, f)) for f in fields}]
        for rep in range(3):
            eng.rollout(6, 21, 6 * rep)            # a wide pickup box: some envs finish
            eng.reset_done(21)
            snap.append({f: eng.get(getattr(m.lib, f)) for f in fields})
        eng.rollout_fused(8, 21, 50, auto_reset=True)      # leaves done == 2 behind for the envs it re-armed itself
        eng.reset_done(21)
        snap.append({f: eng.get(getattr(m.lib, f)) for f in fields})
        eng.reset_random(21, 9)
        snap.append({f: eng.get(getattr(m.lib, f)) for f in fields})
        outs.append(snap)
        eng.close()
    assert outs[0][1]["F_EPISODES"].max() > 4 or n < 64       # somebody did finish and was re-armed
    for a, b in zip(*outs):
        for f in fields:
            np.testing.assert_array_equal(a[f], b[f], err_msg=f)


def test_step_random_equals_sample_then_step(m):
    n, k = 10000, 7
    a, b = m.StepEngine(n, k), m.StepEngine(n, k)
    a.reset_random(1, 0)
    b.reset_random(1, 0)
    for t in range(5):
        a.sample_actions(9, t)
        a.step()
        b.step_random(9, t)
    for f in (m.lib.F_GOALS, m.lib.F_OBS, m.lib.F_REWARD, m.lib.F_DONE, m.lib.F_ALIVE, m.lib.F_EE,
              m.lib.F_TOTAL_REWARD, m.lib.F_POINTS, m.lib.F_DONE_BITS):
        np.testing.assert_array_equal(a.get(f), b.get(f))
    c = m.StepEngine(n, k)
    c.reset_random(1, 0)
    c.rollout(5, 9, 0)
    np.testing.assert_array_equal(c.get(m.lib.F_OBS), b.get(m.lib.F_OBS))
    np.testing.assert_array_equal(c.total_reward(), b.total_reward())


@pytest.mark.parametrize("case", ["ref", "ref_k32", "runtime", "dh7", "rt5", "two_substeps", "terminate"])
def test_step_kernel_variants_are_bit_identical(m, monkeypatch, case):
    """The host picks one of four schedules of the same arithmetic by batch size: one env per lane (streaming, or with
    the targets prefetched into registers) or one env spread over 2 / 4 lanes (step_split_kernel).  Forced here through
    MT_SPLIT / MT_PREFETCH on the same batch: every field equal bit for bit, staged and in-kernel actions, ragged sizes."""
    rng = np.random.RandomState(2)
    rt5 = np.column_stack([rng.uniform(0, 9, 5), rng.choice([-np.pi / 2, 0.3, np.pi / 2], 5), rng.uniform(2, 12, 5), np.zeros(5)])
    kw, n, k = {"ref": (dict(), 100003, 7), "ref_k32": (dict(), 65, 32), "runtime": (dict(specialize=False), 5000, 10),
                "dh7": (dict(dh_table=m.DH7_TABLE, radius=92.6), 70001, 7), "rt5": (dict(dh_table=rt5, radius=40.0), 1234, 3),
                "two_substeps": (dict(substeps=2), 999, 5), "terminate": (dict(terminate_on_ground=True), 4097, 7)}[case]
    fields = ("F_GOALS", "F_ALIVE", "F_TOTAL_REWARD", "F_POINTS", "F_OBS", "F_REWARD", "F_DONE", "F_EE", "F_DONE_BITS")
    outs = {}
    for name, split, pf in (("streaming", 0, 0), ("prefetch", 0, 1), ("split2", 2, 0), ("split4", 4, 0)):
        monkeypatch.setenv("MT_SPLIT", str(split))
        monkeypatch.setenv("MT_PREFETCH", str(pf))
        e = m.StepEngine(n, k, pickup_tol=20.0, **kw)
        e.reset_random(3, 0)
        e.rollout(5, 3, 0)                                      # in-kernel actions
        acts = np.random.RandomState(1).randint(-180, 180, size=(n, e.dof)).astype(np.float32)
        acts[n // 2, 0] = np.nan                                # one unusable action: holds its pose in every variant
        e.step(acts)                                            # staged actions
        outs[name] = {f: e.get(getattr(m.lib, f)) for f in fields}
        outs[name]["bad"] = np.array(e.bad_action_count())
        e.close()
    for name in ("prefetch", "split2", "split4"):
        for f, v in outs["streaming"].items():
            np.testing.assert_array_equal(outs[name][f], v, err_msg=f"{name} {f}")
    assert outs["streaming"]["bad"] == 1


@pytest.mark.parametrize("case", ["ref_k2", "ref_k32", "dh7", "runtime", "rt5_k1"])
@pytest.mark.parametrize("auto_reset", [False, True])
def test_rollout_kernel_variants_are_bit_identical(m, monkeypatch, case, auto_reset):
    """mt_rollout_fused has the same lane-split schedules for tiny batches (rollout_split_kernel): forced through
    MT_SPLIT on one batch, state and last-step outputs, ring and episode counters equal bit for bit."""
    rng = np.random.RandomState(2)
    rt5 = np.column_stack([rng.uniform(0, 9, 5), rng.choice([-np.pi / 2, 0.3, np.pi / 2], 5), rng.uniform(2, 12, 5), np.zeros(5)])
    kw, n, k = {"ref_k2": (dict(), 20000, 2), "ref_k32": (dict(), 777, 32), "dh7": (dict(dh_table=m.DH7_TABLE, radius=92.6), 3001, 3),
                "runtime": (dict(specialize=False), 1000, 5), "rt5_k1": (dict(dh_table=rt5, radius=40.0), 513, 1)}[case]
    outs = {}
    for split in (0, 2, 4):
        monkeypatch.setenv("MT_SPLIT", str(split))
        e = m.StepEngine(n, k, pickup_tol=25.0, **kw)
        e.reset_random(9, 0)
        e.rollout_fused(3, 9, 0, auto_reset=auto_reset)
        e.rollout_fused(27, 9, 3, auto_reset=auto_reset)
        outs[split] = {f: e.get(getattr(m.lib, f)) for f in STATE_FIELDS + STEP_FIELDS + ("F_RETURN_RING",)}
        e.close()
    for split in (2, 4):
        for f, v in outs[0].items():
            np.testing.assert_array_equal(outs[split][f], v, err_msg=f"split {split} {f}")
    if auto_reset and case in ("ref_k2", "rt5_k1"):
        assert outs[0]["F_EPISODES"].max() >= 2


def test_shard_invariance(m):
    """Same seed => same per-env results however the envs are split over handles."""
    from manytor_amd.distributed import shard_range
    n, k = 30000, 7
    whole = m.StepEngine(n, k)
    whole.reset_random(5, 0)
    whole.rollout(6, 5, 0)
    for world in (2, 4, 8):
        parts = []
        for r in range(world):
            base, cnt = shard_range(n, r, world)
            e = m.StepEngine(cnt, k, env_id_base=base)
            e.reset_random(5, 0)
            e.rollout(6, 5, 0)
            parts.append(e)
        for f in (m.lib.F_OBS, m.lib.F_TOTAL_REWARD, m.lib.F_ALIVE, m.lib.F_GOALS, m.lib.F_REWARD):
            np.testing.assert_array_equal(np.concatenate([p.get(f) for p in parts]), whole.get(f))


# --------------------------------------------------------------------------- full-size properties (configs 2, 3, 5)
@pytest.mark.parametrize("n,table_name", [(65536, "ref"), (1048576, "ref"), (1048576, "dh7")])
def test_full_size_properties(m, mo, n, table_name):
    """At BASELINE.json's sizes: size-independent invariants + an oracle check on a strided sample."""
    table = m.REF_DH_TABLE if table_name == "ref" else m.DH7_TABLE
    radius = 51.3 if table_name == "ref" else 92.6
    dof, k = len(table), 7
    from oracle import philox_ref as px
    eng = m.StepEngine(n, k, dh_table=table, radius=radius)
    eng.reset_random(0x5EED, 0)
    p0 = eng.points()
    assert (p0[..., 2] >= 0).all() and (np.linalg.norm(p0.astype(np.float64), axis=-1) <= radius * (1 + 1e-6)).all()
    sample = np.arange(0, n, max(1, n // 4096))[:4096]
    ora = mo.BatchOracle(len(sample), k, table=np.asarray(table), radius=radius)
    ora.reset(p0[sample].astype(np.float64))
    ret = np.zeros(n, dtype=np.float64)
    ever_guarded = np.zeros(len(sample), dtype=bool)
    for t in range(4):
        eng.step_random(0x5EED, t)
        a = eng.goals()               # the action drawn in-kernel is the new goals (manytor.py:184)
        assert a.min() >= -180 and a.max() <= 179 and np.all(a == np.round(a))
        rew, done, alive, obs = eng.reward(), eng.done(), eng.alives(), eng.obs()
        ret += rew
        assert set(np.unique(rew)) <= {-1, 0, 1}
        np.testing.assert_array_equal(done, ~alive.any(axis=1))
        np.testing.assert_array_equal(a, px.sample_actions(0x5EED, np.arange(n, dtype=np.uint64), t, dof))
        o = obs.reshape(n, k, 3)
        assert np.isfinite(o).all() and (o >= 0).all() and (o[..., 1:] <= 90.0 + 1e-3).all()
        pre_alive = ora.alives.copy()
        obs_ref, rew_ref, _ = ora.step(a[sample].astype(np.float64))
        assert np.abs(eng.ee()[sample] - ora.joints_coordinates[:, -1]).max() <= POS_TOL
        pm = np.where(pre_alive, ora.pickup_margin, np.inf).min(axis=1)
        ever_guarded |= (ora.ground_margin < GUARD) | (pm < GUARD)
        okk = ~ever_guarded
        np.testing.assert_array_equal(rew[sample][okk], rew_ref[okk])
        np.testing.assert_array_equal(alive[sample][okk], ora.alives[okk])
        assert_obs_close(obs[sample][okk], obs_ref[okk], ora.joints_coordinates[okk, -2], ora.points[okk], pre_alive[okk])
    np.testing.assert_array_equal(eng.total_reward(), ret.astype(np.float32))      # return = sum of rewards
    assert ever_guarded.mean() < 0.02
    bits = eng.done_bits()
    assert int(sum(bin(int(b)).count("1") for b in bits)) == int(eng.done().sum())
    # idempotence of the standalone passes at a fixed pose
    obs_a = eng.obs().copy()
    eng.observe()
    obs_b = eng.obs()
    alive_now = eng.alives()
    dead = np.repeat(~alive_now, 3, axis=1)
    assert np.all(obs_b[dead] == 0)
    # same pose, same targets: equal up to the rounding of two differently scheduled fp32 evaluations
    stay = alive_now & (obs_a.reshape(n, k, 3)[..., 0] > 0)           # targets alive in both passes
    idx = np.flatnonzero(stay.any(axis=1))[:: max(1, n // 8192)]
    elbow = eng.joints_coordinates()[idx, -2].astype(np.float64)
    assert_obs_close(obs_b[idx], obs_a[idx].astype(np.float64), elbow, eng.points()[idx].astype(np.float64), stay[idx])
    alive_a = eng.alives().copy()
    eng.check_done()
    np.testing.assert_array_equal(eng.alives(), alive_a)


def test_random_action_statistics_match_the_reference(m):
    """Distribution-level check at scale (measured on the reference with random actions, K = 7:
    reward -1 on 80.0 % of the steps, 0 on 18.9 %, +1 on 1.1 %; nobody finishes an episode of 50 steps)."""
    n = 65536
    eng = m.StepEngine(n, 7)
    eng.reset_random(123, 0)
    counts = np.zeros(3)
    for t in range(50):
        eng.step_random(123, t)
        r = eng.reward()
        counts += [(r == -1).sum(), (r == 0).sum(), (r == 1).sum()]
    frac = counts / counts.sum()
    assert abs(frac[0] - 0.800) < 0.02 and abs(frac[1] - 0.189) < 0.02 and abs(frac[2] - 0.011) < 0.004, frac
    assert eng.done().mean() < 1e-3
    picked = 7 - eng.alives().sum(axis=1)
    assert abs(picked.mean() - 1.74) < 0.1 and picked.max() <= 7      # oracle on the same streams (3000 envs): 1.742


@pytest.mark.parametrize("n", [1, 63, 64, 65, 255, 257, 16385, 1000003])
def test_ragged_sizes_tail_handling(m, n):
    """Sizes around wavefront / block / padding boundaries and a large prime: tail lanes exit cleanly, the last
    done_bits word only carries live lanes, fused and per-step paths agree, neighbours of the arena are untouched."""
    k = 3
    a, b = m.StepEngine(n, k, pickup_tol=20.0), m.StepEngine(n, k, pickup_tol=20.0)
    for e in (a, b):
        e.reset_random(4, 0)
    a.rollout(6, 4, 0)
    b.rollout_fused(6, 4, 0)
    for f in ("F_GOALS", "F_ALIVE", "F_TOTAL_REWARD", "F_POINTS", "F_OBS", "F_REWARD", "F_DONE", "F_EE", "F_DONE_BITS"):
        np.testing.assert_array_equal(a.get(getattr(m.lib, f)), b.get(getattr(m.lib, f)), err_msg=f)
    bits = a.done_bits()
    assert bits.shape == ((n + 63) // 64,)
    unpacked = ((bits[:, None] >> np.arange(64, dtype=np.uint64)) & np.uint64(1)).astype(bool).ravel()
    np.testing.assert_array_equal(unpacked[:n], a.done())
    assert not unpacked[n:].any()
    from oracle import philox_ref as px
    np.testing.assert_array_equal(a.goals(), px.sample_actions(4, np.arange(n, dtype=np.uint64), 5, 4))


# the kernel mt_create picks by batch size (engine.hip): 4 lanes per env, 2 lanes per env (+ reset_split_kernel), one env
# per lane with the targets prefetched (+ the cached HIP graph of mt_rollout), the same above the graph limit, and the
# headline size (prefetch for the reference arm, streaming for other tables)
DISPATCH_REGIMES = [(32768, "L=4"), (65536, "L=2"), (131072, "pf=8"), (262144, None), (1048576, None)]


@pytest.mark.parametrize("table_name", ["ref", "dh7"])
@pytest.mark.parametrize("n,expect", DISPATCH_REGIMES)
def test_full_size_every_env_against_the_c_oracle(m, monkeypatch, table_name, n, expect):
    """BASELINE.json's sizes (configs[1] = 65 536, configs[2] / [4] = 1 048 576, the 131 072-arm shard of the 1 M strong
    scaling point) and every dispatch regime in between, ALL envs compared (not a sample): the C restatement of the
    reference (oracle/manytor_oracle.c, OpenMP) steps the same targets and actions; positions, observations and --
    outside the guard band -- rewards / alive masks / done flags must agree for every env, three steps in a row, on the
    kernel mt_create actually selects for that size.  Up to 262 144 arms the same steps are then run as ONE mt_rollout
    segment, requested twice -- the default there is k steps per launch through the rollout kernels (round 4); with
    MT_ROLLOUT_K=1 it is a launch per step, replayed from the cached HIP graph at the second request up to 131 072 arms --
    and must give the same bits either way."""
    from oracle import c_oracle
    table = m.REF_DH_TABLE if table_name == "ref" else m.DH7_TABLE
    radius = 51.3 if table_name == "ref" else 92.6
    k = 7
    eng = m.StepEngine(n, k, dh_table=table, radius=radius)
    if expect:
        assert expect in eng.step_kernel_name(), eng.step_kernel_name()
    ora = c_oracle.COracle(n, k, table=np.asarray(table), radius=radius, threads=16)
    eng.reset_random(0xC0FFEE, 0)
    from oracle import philox_ref as px
    np.testing.assert_array_equal(eng.points(), px.sample_targets(0xC0FFEE, np.arange(n, dtype=np.uint64), 0, k, radius))
    ora.reset(eng.points().astype(np.float64))
    guarded_total = 0
    for t in range(3):
        eng.step_random(0xC0FFEE, t)
        pre_alive = ora.alives.copy()
        obs_ref, rew_ref, done_ref = ora.step(eng.goals().astype(np.float64))
        assert np.abs(eng.joints_coordinates() - ora.joints_coordinates).max() <= POS_TOL
        pm = np.where(pre_alive, ora.pickup_margin, np.inf).min(axis=1)
        risky = (ora.ground_margin < GUARD) | (pm < GUARD)
        ok = ~risky
        guarded_total += int(risky.sum())
        np.testing.assert_array_equal(eng.reward()[ok], rew_ref[ok])
        np.testing.assert_array_equal(eng.done()[ok], done_ref[ok])
        alive_gpu = eng.alives()
        np.testing.assert_array_equal(alive_gpu[ok], ora.alives[ok])
        assert_obs_close(eng.obs(), obs_ref, ora.joints_coordinates[:, -2], ora.points, pre_alive)
        idx = np.flatnonzero(risky)                      # re-synchronise the few envs inside the guard band
        ora.alive_u8[idx] = alive_gpu[idx]
        ora.total_reward[idx] = eng.total_reward()[idx]
        ora.points[idx] = eng.points()[idx].astype(np.float64)
        np.testing.assert_array_equal(eng.total_reward(), ora.total_reward.astype(np.float32))
    assert guarded_total < 3 * n * 2e-3 + 8, guarded_total
    if n <= 262144:
        # the same three steps as ONE mt_rollout segment, requested twice (not F_LAST_RETURN: a full reset stores the return
        # of the episode it ends): the default dispatch, then one launch per step (the second request replays the cached graph)
        want = {f: eng.get(getattr(m.lib, f)) for f in STATE_FIELDS + STEP_FIELDS if f != "F_LAST_RETURN"}
        assert eng.dispatch()["rollout"]["form"] == "multi_step"
        monkeypatch.setenv("MT_ROLLOUT_K", "1")
        per_step = m.StepEngine(n, k, dh_table=table, radius=radius)
        assert per_step.dispatch()["rollout"]["form"] == ("graph_replay" if n <= 131072 else "chained_steps")
        for e in (eng, per_step):
            for attempt in range(2):
                e.reset_random(0xC0FFEE, 0)
                e.rollout(3, 0xC0FFEE, 0)
                for f, v in want.items():
                    np.testing.assert_array_equal(e.get(getattr(m.lib, f)), v, err_msg=f"rollout attempt {attempt}: {f}")
        per_step.close()


def test_full_size_fused_equals_per_step(m):
    """1 048 576 arms x one 50-step episode: one fused launch vs 50 launches, every state bit equal."""
    n, k, T = 1048576, 7, 50
    a, b = m.StepEngine(n, k), m.StepEngine(n, k)
    for e in (a, b):
        e.reset_random(77, 0)
    a.rollout(T, 77, 0)
    b.rollout_fused(T, 77, 0)
    for f in ("F_GOALS", "F_ALIVE", "F_TOTAL_REWARD", "F_POINTS", "F_OBS", "F_REWARD", "F_DONE", "F_EE", "F_DONE_BITS"):
        np.testing.assert_array_equal(a.get(getattr(m.lib, f)), b.get(getattr(m.lib, f)), err_msg=f)
    ret = a.total_reward()
    assert -50 <= ret.min() and ret.max() <= 50 and abs(ret.mean() + 38.9) < 0.5      # oracle on the same streams: -38.93


def test_reset_done_rearms_only_finished_envs(m):
    from oracle import philox_ref as px
    n, k = 4096, 1
    eng = m.StepEngine(n, k, pickup_tol=30.0)
    eng.reset_random(3, 5)
    assert np.all(eng.episodes() == 5)
    for t in range(6):
        eng.step_random(3, t)
    done = eng.done()
    assert 0 < done.sum() < n
    goals, total, pts = eng.goals(), eng.total_reward(), eng.points()
    eng.reset_done(3)
    assert eng.alives()[done].all() and not eng.done().any() and not eng.done_bits().any()
    assert np.all(eng.goals()[done] == 0) and np.all(eng.total_reward()[done] == 0)
    np.testing.assert_array_equal(eng.episodes(), np.where(done, 6, 5))            # own counter advanced
    np.testing.assert_array_equal(eng.last_return()[done], total[done])              # finished return kept
    np.testing.assert_array_equal(eng.goals()[~done], goals[~done])
    np.testing.assert_array_equal(eng.total_reward()[~done], total[~done])
    np.testing.assert_array_equal(eng.points()[~done], pts[~done])
    fresh = px.sample_targets(3, np.arange(n, dtype=np.uint64), 6, k, 51.3)          # keyed by (env, episode 6)
    np.testing.assert_array_equal(eng.points()[done], fresh[done])


STATE_FIELDS = ("F_GOALS", "F_ALIVE", "F_TOTAL_REWARD", "F_POINTS", "F_EPISODES", "F_LAST_RETURN")
STEP_FIELDS = ("F_OBS", "F_REWARD", "F_DONE", "F_EE", "F_DONE_BITS")


@pytest.mark.parametrize("cfg", [dict(n=10000, k=7, table="ref"), dict(n=3000, k=10, table="ref"),
                                 dict(n=2048, k=7, table="dh7"), dict(n=1000, k=3, table="rt5"),
                                 dict(n=777, k=32, table="ref")])
def test_fused_rollout_equals_launch_per_step(m, cfg):
    """mt_rollout_fused(T) is bit-identical to T x mt_step_random (state in registers/LDS vs HBM round trips)."""
    rng = np.random.RandomState(1)
    table = {"ref": m.REF_DH_TABLE, "dh7": m.DH7_TABLE,
             "rt5": np.column_stack([rng.uniform(0, 9, 5), rng.choice([-np.pi / 2, 0.3, np.pi / 2], 5),
                                     rng.uniform(2, 12, 5), np.zeros(5)])}[cfg["table"]]
    radius = 92.6 if cfg["table"] == "dh7" else 51.3
    a = m.StepEngine(cfg["n"], cfg["k"], dh_table=table, radius=radius)
    b = m.StepEngine(cfg["n"], cfg["k"], dh_table=table, radius=radius)
    for e in (a, b):
        e.reset_random(21, 0)
    for t0, T in ((0, 1), (1, 7), (8, 25)):
        for t in range(t0, t0 + T):
            a.step_random(21, t)
        b.rollout_fused(T, 21, t0)
        for f in STATE_FIELDS + STEP_FIELDS:
            np.testing.assert_array_equal(a.get(getattr(m.lib, f)), b.get(getattr(m.lib, f)), err_msg=f)


def test_fused_rollout_with_auto_reset_equals_step_plus_reset_done(m):
    n, k = 20000, 2
    a = m.StepEngine(n, k, pickup_tol=25.0)
    b = m.StepEngine(n, k, pickup_tol=25.0)
    for e in (a, b):
        e.reset_random(9, 0)
    T = 30
    for t in range(T):
        a.step_random(9, t)
        if t == T - 1:
            last = {f: a.get(getattr(m.lib, f)) for f in STEP_FIELDS}
        a.reset_done(9)
    b.rollout_fused(T, 9, 0, auto_reset=True)
    for f in STATE_FIELDS:
        np.testing.assert_array_equal(a.get(getattr(m.lib, f)), b.get(getattr(m.lib, f)), err_msg=f)
    for f in STEP_FIELDS:                                   # the last step's outputs survive in the fused path
        got = b.get(getattr(m.lib, f))
        if f == "F_DONE":                                   # ... with "finished" spelled 2 = already re-armed in-kernel
            assert set(np.unique(got)) <= {0, 2}
            got = (got != 0).astype(np.uint8)
        np.testing.assert_array_equal(last[f], got, err_msg=f)
    ep = b.episodes()
    assert ep.max() >= 2 and (ep == 0).any()                # some envs finished several episodes, some none
    assert np.all(b.last_return()[ep > 0] == np.round(b.last_return()[ep > 0]))


# --------------------------------------------------------------------------- sub-step trace / viewer (8f ranks 3-4)
def test_route_trace_matches_f3_substeps(m, golden):
    g = golden("f3_substep_trace")
    tr = m.route_trace(g["prev"], g["action"])
    assert tr.shape == g["jc"].shape
    assert np.abs(tr - g["jc"]).max() <= POS_TOL
    t7 = m.route_trace(np.zeros((3, 7)), np.full((3, 7), 30.0), dh_table=m.DH7_TABLE, substeps=5)
    assert t7.shape == (3, 5, 7, 3) and np.all(t7[:, :, 0] == 0)
    np.testing.assert_allclose(t7[0, 0, -1], [0, 0, 34 + 40 + 40 + 12.6 - 0.0], atol=20)   # zero pose: arm upright-ish


def test_environment_trajectory_attribute(m, golden):
    """manytor.py:135,190: `trajectory` = (0, 0, 51.3) seed + one end-effector row per sub-step."""
    g = golden("f3_substep_trace")
    env = m.Environment(1)
    np.random.seed(0)
    env.reset()
    assert env.trajectory.shape == (3,)
    env.goals = g["prev"][5]
    env.step(list(g["action"][5]))
    assert env.trajectory.shape == (26, 3)
    np.testing.assert_allclose(env.trajectory[0], [0, 0, 51.3])
    assert np.abs(env.trajectory[1:] - g["jc"][5][:, 3, :]).max() <= POS_TOL
    env.step(list(g["action"][6]))
    assert env.trajectory.shape == (51, 3)
    env.reset()
    assert env.trajectory.shape == (3,)


def test_multienv_streams_reference_wire_format(m):
    import json
    import socket
    recv = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    recv.setsockopt(socket.SOL_SOCKET, socket.SO_RCVBUF, 1 << 22)
    recv.bind(("127.0.0.1", 0))
    recv.settimeout(5.0)
    np.random.seed(3)
    me = m.Multienv((2, 2), 7, viewer=("127.0.0.1", recv.getsockname()[1]), view_envs=2)
    me.reset()
    me.step(me.action_sample())                      # not rendering yet: nothing is sent
    me.render()
    prev = np.array([e.goals for e in me.environment[:2]])
    a = me.action_sample()
    me.step(a)
    me.render(stop_render=True)
    got = [recv.recvfrom(2048)[0] for _ in range(1 + 2 * 25 + 1)]
    assert json.loads(got[0]) == [2, 7, 3, [2, 2]]
    frames = [np.array(json.loads(gm), dtype=np.float64) for gm in got[1:-1]]
    assert all(f.size == 3 * (1 + 4 + 7 + 1) for f in frames)
    last_env0 = [f for f in frames if int(f[0]) == 0][-1].reshape(-1, 3)
    np.testing.assert_allclose(last_env0[1:5], me.environment[0].joints_coordinates, atol=1e-3)
    first_env1 = [f for f in frames if int(f[0]) == 1][0].reshape(-1, 3)
    ref = m.route_trace(prev[1:2], prev[1:2])[0, 0]
    np.testing.assert_allclose(first_env1[1:5], ref, atol=1e-3)       # sub-step 0 = the previous pose
    assert json.loads(got[-1])[2] == 2
    recv.close()


# --------------------------------------------------------------------------- boundary behaviour
def test_errors_are_loud(m):
    eng = m.StepEngine(8, 3)
    with pytest.raises(m.ManytorError):
        eng.step(np.zeros((8, 4)))                      # before reset
    eng.reset(np.zeros((8, 3, 3), dtype=np.float32))
    with pytest.raises(ValueError):
        eng.step(np.zeros((7, 4)))
    with pytest.raises(ValueError):
        m.StepEngine(8, 40)
    with pytest.raises(ValueError):
        m.StepEngine(8, 3, substeps=1)
    eng.close()
    with pytest.raises(RuntimeError):
        eng.step(np.zeros((8, 4)))


def test_step_host_equals_set_step_get(m):
    n, k = 777, 5
    rng = np.random.RandomState(3)
    a, b = m.StepEngine(n, k), m.StepEngine(n, k)
    pts = rng.uniform(-30, 30, size=(n, k, 3)).astype(np.float32)
    a.reset(pts)
    b.reset(pts)
    for cast in (np.int64, np.float32, np.float64, np.int32):
        act = rng.randint(-180, 180, size=(n, 4)).astype(cast)
        a.step(act)
        obs, rew, done = b.step_host(act)
        np.testing.assert_array_equal(obs, a.obs())
        np.testing.assert_array_equal(rew, a.reward())
        np.testing.assert_array_equal(done, a.done())
        np.testing.assert_array_equal(b.goals(), a.goals())
    with pytest.raises(ValueError):
        b.step_host(np.zeros((n - 1, 4)))


def test_action_dtypes_and_layouts(m):
    n = 300
    rng = np.random.RandomState(0)
    a = rng.randint(-180, 180, size=(n, 4))
    eng = m.StepEngine(n, 2)
    for cast in (np.int64, np.int32, np.float32, np.float64, list):
        eng.set_actions(a.tolist() if cast is list else a.astype(cast))
        np.testing.assert_array_equal(eng.actions(), a.astype(np.float32))


def test_torch_device_views_and_device_inputs(m):
    import torch
    n, k = 1000, 7
    eng = m.StepEngine(n, k)
    eng.reset_random(1, 0)
    eng.step_random(1, 0)
    eng.sync()
    for f in (m.lib.F_OBS, m.lib.F_GOALS, m.lib.F_POINTS, m.lib.F_EE):
        t = eng.device_tensor(f)
        ref = eng.get(f).reshape(n, -1)
        assert t.is_cuda and tuple(t.shape) == (ref.shape[1], n)
        np.testing.assert_array_equal(t.T.cpu().numpy(), ref)
    np.testing.assert_array_equal(eng.device_tensor(m.lib.F_REWARD).cpu().numpy(), eng.reward())
    np.testing.assert_array_equal(eng.device_tensor(m.lib.F_DONE).cpu().numpy(), eng.get(m.lib.F_DONE))
    np.testing.assert_array_equal(eng.device_tensor(m.lib.F_TOTAL_REWARD).cpu().numpy(), eng.total_reward())
    # device-resident actions in, both layouts
    a = torch.randint(-180, 180, (n, 4), device="cuda")
    eng.set_actions(a)
    np.testing.assert_array_equal(eng.actions(), a.cpu().numpy().astype(np.float32))
    soa = torch.zeros((4, eng.ld), dtype=torch.float32, device="cuda")
    soa[:, :n] = a.T.to(torch.float32) + 1
    eng.set_actions(soa)
    np.testing.assert_array_equal(eng.actions(), a.cpu().numpy().astype(np.float32) + 1)
    # device-resident targets in
    pts = torch.rand((n, k, 3), device="cuda") * 20
    eng.reset(pts)
    np.testing.assert_array_equal(eng.points(), pts.cpu().numpy())


def test_rccl_single_rank_gather_of_engine_memory(m):
    """The one collective of the multi-GPU path on the backend the 8-GPU run uses, with one rank: torch's "nccl"
    (= RCCL) process group as control plane, the unique id shipped through it (D.exchange_unique_id), the engine's own
    communicator (mt_comm_init), and mt_gather_returns reading the arena; torch's RCCL reading the same arena memory
    through a zero-copy view gives the same values."""
    import socket

    import torch
    import torch.distributed as dist
    from manytor_amd import distributed as D
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    n = 70000
    eng = m.StepEngine(n, 7)
    eng.use_torch_stream()
    eng.reset_random(2, 0)
    eng.rollout(5, 2, 0)
    dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        view = eng.device_tensor(m.lib.F_TOTAL_REWARD)
        out = torch.empty(n, dtype=torch.float32, device="cuda")
        dist.all_gather_into_tensor(out, view)                     # torch's RCCL, straight from the arena
        uid = D.exchange_unique_id(m.comm_unique_id, 0)
        eng.comm_init(uid, 0, 1)
        full = eng.gather_returns()                                # the engine's own RCCL communicator (C ABI)
        dist.barrier()
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    ref = eng.total_reward()
    np.testing.assert_array_equal(out.cpu().numpy(), ref)
    np.testing.assert_array_equal(full.cpu().numpy(), ref)
    eng.comm_destroy()


@pytest.mark.parametrize("case", ["ref_small", "ref_split", "dh7", "runtime5", "ref_large_forced"])
def test_rollout_through_a_replayed_graph_equals_plain_launches(m, case, monkeypatch):
    """mt_rollout replays a captured HIP graph of its T launches on small batches (MT_GRAPH forces it on / off): the
    same bits as T plain launches, across segment lengths, step offsets, seeds, resets, a stream change and the
    eviction of cached graphs."""
    kw, n, k = {"ref_small": (dict(), 100003, 7), "ref_split": (dict(), 20000, 3),
                "dh7": (dict(dh_table=m.DH7_TABLE, radius=92.6), 3001, 7),
                "runtime5": (dict(dh_table=[[0, -1.2, 5, 0], [6, 0.37, 0, 0.2], [0, 1.57, 7, 0], [4, 0, 0, 0], [3, -1.57, 2, 0]],
                                  radius=25.0), 7777, 5),
                "ref_large_forced": (dict(), 300000, 7)}[case]
    fields = ("F_GOALS", "F_ALIVE", "F_TOTAL_REWARD", "F_POINTS", "F_OBS", "F_REWARD", "F_DONE", "F_DONE_BITS", "F_EE")
    outs = []
    monkeypatch.setenv("MT_ROLLOUT_K", "1")      # one launch per step: the form the graphs exist for (tests/test_gpu_r04.py has k > 1)
    for mode in ("0", "1", None):                                     # None: the library's own choice (graph from the
        if mode is None:                                              # second request of a segment length, small batches)
            monkeypatch.delenv("MT_GRAPH", raising=False)
        else:
            monkeypatch.setenv("MT_GRAPH", mode)
        eng = m.StepEngine(n, k, pickup_tol=15.0, **kw)
        eng.reset_random(8, 0)
        step = 0
        snaps = []
        for rep, T in enumerate((50, 20, 50, 7, 4, 3, 50)):           # 3 < 4: plain launches in both modes
            seed = 8 if rep < 5 else 99                               # a new seed makes new graphs
            eng.rollout(T, seed, step)
            step += T
            if rep == 2:
                eng.reset_random(8, 1)
            if rep == 3:
                eng.step_random(8, 1234)                              # a plain launch between two replays
            snaps.append({f: eng.get(getattr(m.lib, f)) for f in fields})
        for T in range(10, 20):                                       # more segment lengths than the cache holds
            eng.rollout(T, 8, 5000 + T)
        eng.use_torch_stream()                                        # the graphs are not tied to the stream they were captured on
        eng.rollout(50, 8, 7000)
        eng.set_stream(None)
        eng.rollout(50, 8, 8000)
        snaps.append({f: eng.get(getattr(m.lib, f)) for f in fields})
        outs.append(snaps)
        eng.close()
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            for f in fields:
                np.testing.assert_array_equal(a[f], b[f], err_msg=f)


def test_step_is_capturable_in_a_hip_graph(m):
    """mt_step launches on the caller's stream and does nothing capture-hostile (no allocation, no sync), so a
    torch stream capture records `write actions -> env step` as one HIP graph; replays equal eager stepping."""
    import torch
    n, k = 20000, 7
    eager, graphed = m.StepEngine(n, k), m.StepEngine(n, k)
    for e in (eager, graphed):
        e.use_torch_stream()
        e.reset_random(6, 0)
    acts = torch.randint(-180, 180, (6, 4, n), device="cuda").to(torch.float32)
    for t in range(6):
        eager.device_tensor(m.lib.F_ACTIONS).copy_(acts[t])
        eager.step()
    static_in = torch.zeros((4, n), device="cuda")
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        graphed.use_torch_stream()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            graphed.device_tensor(m.lib.F_ACTIONS).copy_(static_in)
            graphed.step()
    torch.cuda.synchronize()
    graphed.reset_random(6, 0)            # the capture pass itself does not execute; start from the same state anyway
    for t in range(6):
        static_in.copy_(acts[t])
        graph.replay()
    torch.cuda.synchronize()
    for f in ("F_GOALS", "F_ALIVE", "F_TOTAL_REWARD", "F_OBS", "F_REWARD", "F_DONE", "F_EE"):
        np.testing.assert_array_equal(eager.get(getattr(m.lib, f)), graphed.get(getattr(m.lib, f)), err_msg=f)


def test_large_multienv_returns_views_not_lists(m):
    me = m.Multienv(env_shape=(128, 64), obj_number=7, rng="device")
    obs = me.reset(returnable=True)
    assert isinstance(obs, m.BatchView) and len(obs) == 8192
    a = me.action_sample()
    obs2, reward, done = me.step(a)
    assert not (done == True)  # noqa: E712
    assert np.asarray(obs2).shape == (8192, 21) and np.asarray(reward).shape == (8192,)
    assert obs2.torch().shape == (8192, 21)
    assert isinstance(me.environment[5].total_reward, float)
    assert len(me.environment) == 8192


@pytest.mark.parametrize("graph", [0, 1])
@pytest.mark.parametrize("chains", [2, 3, 4])
@pytest.mark.parametrize("n,k,table_name", [(100003, 7, "ref"), (777, 3, "ref"), (300, 2, "ref"), (70001, 5, "dh7"),
                                            (262144, 7, "ref")])
def test_rollout_in_independent_chains_equals_plain_launches(m, monkeypatch, n, k, table_name, chains, graph):
    """mt_rollout may run as several chains of launches -- contiguous 256-aligned env ranges on separate streams, forked
    from and joined to the handle's stream (MT_CHAINS; engine.hip: launch_chained_steps), also inside the cached HIP
    graph -- because a step of env i depends only on env i.  Same kernels on row views + the global env id in the RNG
    key => every field equal bit for bit to the single-stream sequence, ragged tails and tiny batches included; and
    work queued on the handle's stream right behind the rollout sees the joined result."""
    table = m.REF_DH_TABLE if table_name == "ref" else m.DH7_TABLE
    radius = 51.3 if table_name == "ref" else 92.6
    fields = STATE_FIELDS + STEP_FIELDS
    monkeypatch.setenv("MT_ROLLOUT_K", "1")      # chains of STEP kernels (what batches > 262 144 envs run); k > 1: test_gpu_r04.py
    monkeypatch.setenv("MT_CHAINS", "1")
    monkeypatch.setenv("MT_GRAPH", "0")
    ref = m.StepEngine(n, k, dh_table=table, radius=radius, pickup_tol=20.0)
    ref.reset_random(6, 0)
    ref.rollout(9, 6, 0)
    ref.reset_done(6)
    ref.rollout(9, 6, 9)
    want = {f: ref.get(getattr(m.lib, f)) for f in fields}
    monkeypatch.setenv("MT_CHAINS", str(chains))
    monkeypatch.setenv("MT_GRAPH", str(graph))
    e = m.StepEngine(n, k, dh_table=table, radius=radius, pickup_tol=20.0)
    e.reset_random(6, 0)
    e.rollout(9, 6, 0)
    e.reset_done(6)                          # queued on the handle's stream: must see every chain's last step
    e.rollout(9, 6, 9)                       # second request of the segment length: the cached graph when graph = 1
    for f, v in want.items():
        np.testing.assert_array_equal(e.get(getattr(m.lib, f)), v, err_msg=f)
    e.rollout(9, 6, 18)                      # and once more through the (now certainly cached) graph / plain chains
    ref.rollout(9, 6, 18)
    for f in fields:
        np.testing.assert_array_equal(e.get(getattr(m.lib, f)), ref.get(getattr(m.lib, f)), err_msg=f)


@pytest.mark.parametrize("cap", [1, 3, 5, 8])
def test_occupancy_cap_of_the_chain_launches_changes_nothing_but_the_launch(m, monkeypatch, cap):
    """The chain launches of the sampled-action prefetch kernel may be capped to fewer resident blocks per CU with dynamic
    LDS nobody touches (engine.hip: step_blocks_per_cu; MT_BLOCKS_PER_CU overrides the size-based choice).  Every cap
    must launch (the pad stays within the 64 KB a launch may ask for without an attribute) and give the bits of the
    uncapped engine -- at a size where the default applies a cap (400 003 arms: three blocks) and for both addressing
    forms of the kernel."""
    fields = STATE_FIELDS + STEP_FIELDS
    n, k = 400003, 7
    got = {}
    for setting in ("0", str(cap), None):
        for flat_from in ("0", str(1 << 40)):
            if setting is None:
                monkeypatch.delenv("MT_BLOCKS_PER_CU", raising=False)
            else:
                monkeypatch.setenv("MT_BLOCKS_PER_CU", setting)
            monkeypatch.setenv("MT_FLAT_FROM", flat_from)
            e = m.StepEngine(n, k, pickup_tol=20.0)
            name = e.step_kernel_name()
            assert "2 chains" in name and "pf=8" in name, name
            assert ("blocks/CU" in name) == (setting != "0"), name
            e.reset_random(21, 0)
            e.rollout(6, 21, 0)
            e.reset_random(21, 1)
            e.rollout(5, 21, 0)
            got[(setting, flat_from)] = {f: e.get(getattr(m.lib, f)) for f in fields}
            e.close()
    ref = got[("0", "0")]
    for key, vals in got.items():
        for f in fields:
            np.testing.assert_array_equal(vals[f], ref[f], err_msg=f"{key}: {f}")


@pytest.mark.parametrize("sampled", [True, False])
@pytest.mark.parametrize("n,k,table_name", [(100003, 7, "ref"), (262144, 3, "ref"), (70001, 7, "dh7"), (999, 9, "ref")])
def test_flat_row_addressing_equals_renewed_lane_offsets(m, monkeypatch, n, k, table_name, sampled):
    """The prefetch kernel exists in two addressing forms (kernels.h, LaneOffset<true | false>; the host takes the FLAT one
    from MT_FLAT_FROM envs per launch, default 393 216: the HBM-bound launches).  Same arithmetic, same memory operations:
    every field equal bit for bit, sampled (TT) and staged actions, ragged sizes, more targets than prefetch slots."""
    table = m.REF_DH_TABLE if table_name == "ref" else m.DH7_TABLE
    radius = 51.3 if table_name == "ref" else 92.6
    fields = STATE_FIELDS + STEP_FIELDS
    monkeypatch.setenv("MT_SPLIT", "0")
    monkeypatch.setenv("MT_PREFETCH", "1")
    monkeypatch.setenv("MT_CHAINS", "1")
    got = {}
    for flat_from in (0, 1 << 40):
        monkeypatch.setenv("MT_FLAT_FROM", str(flat_from))
        e = m.StepEngine(n, k, dh_table=table, radius=radius, pickup_tol=20.0)
        name = e.step_kernel_name()
        assert "pf=8" in name and ("flat=1" in name) == (flat_from == 0), name
        e.reset_random(11, 0)
        if sampled:
            e.rollout(7, 11, 0)
        else:
            rng = np.random.RandomState(5)
            for _ in range(3):
                e.step(rng.uniform(-190.0, 190.0, size=(n, e.dof)).astype(np.float32))
        got[flat_from] = {f: e.get(getattr(m.lib, f)) for f in fields}
        e.close()
    for f in fields:
        np.testing.assert_array_equal(got[0][f], got[1 << 40][f], err_msg=f)


@pytest.mark.parametrize("table_name", ["ref", "dh7"])
@pytest.mark.parametrize("n,expect", [(32768, "L=4"), (65536, "L=2"), (131072, "pf=8"), (1048576, None)])
def test_staged_action_step_every_env_against_the_c_oracle(m, table_name, n, expect):
    """The policy-in-the-loop path -- mt_set_actions + mt_step, i.e. step_kernel / step_split_kernel with SAMPLE = false and
    fractional-degree actions -- at every dispatch regime, every env against the C restatement of the reference
    (manytor.py:255-260): the sampled-action tests above never run these instantiations at these sizes."""
    from oracle import c_oracle
    table = m.REF_DH_TABLE if table_name == "ref" else m.DH7_TABLE
    radius = 51.3 if table_name == "ref" else 92.6
    k, dof = 7, len(table)
    rng = np.random.RandomState(n % 9973)
    eng = m.StepEngine(n, k, dh_table=table, radius=radius)
    if expect:
        assert expect in eng.step_kernel_name(), eng.step_kernel_name()
    ora = c_oracle.COracle(n, k, table=np.asarray(table), radius=radius, threads=16)
    eng.reset_random(0xAC7, 0)
    ora.reset(eng.points().astype(np.float64))
    guarded_total = 0
    for t in range(3):
        # a policy's output: floats, some far outside [-180, 180), some tiny moves
        act = rng.uniform(-200.0, 200.0, size=(n, dof)).astype(np.float32)
        small = rng.rand(n) < 0.25
        act[small] = (ora.goals[small] + rng.uniform(-2.0, 2.0, size=(int(small.sum()), dof))).astype(np.float32)
        eng.step(act)
        pre_alive = ora.alives.copy()
        obs_ref, rew_ref, done_ref = ora.step(act.astype(np.float64))
        np.testing.assert_array_equal(eng.goals(), act)                      # goals = action, manytor.py:184
        assert np.abs(eng.joints_coordinates() - ora.joints_coordinates).max() <= POS_TOL
        pm = np.where(pre_alive, ora.pickup_margin, np.inf).min(axis=1)
        risky = (ora.ground_margin < GUARD) | (pm < GUARD)
        ok = ~risky
        guarded_total += int(risky.sum())
        np.testing.assert_array_equal(eng.reward()[ok], rew_ref[ok])
        np.testing.assert_array_equal(eng.done()[ok], done_ref[ok])
        alive_gpu = eng.alives()
        np.testing.assert_array_equal(alive_gpu[ok], ora.alives[ok])
        assert_obs_close(eng.obs(), obs_ref, ora.joints_coordinates[:, -2], ora.points, pre_alive)
        idx = np.flatnonzero(risky)                      # re-synchronise the few envs inside the guard band
        ora.alive_u8[idx] = alive_gpu[idx]
        ora.total_reward[idx] = eng.total_reward()[idx]
        ora.points[idx] = eng.points()[idx].astype(np.float64)
        np.testing.assert_array_equal(eng.total_reward(), ora.total_reward.astype(np.float32))
    assert guarded_total < 3 * n * 3e-3 + 8, guarded_total
    assert eng.bad_action_count() == 0
