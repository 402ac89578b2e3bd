"""mt_rollout_fused against the CPU oracle directly (not against the per-step HIP path), every env, at BASELINE.json's
sizes; and BASELINE.json configs[3]: 4 194 304 arms as one handle vs 8 shards of 524 288 (what 8 ranks would own).

The GPU runs the whole rollout in ONE launch; the C restatement of the reference (oracle/manytor_oracle.c) is then
stepped T times on the host with the same Philox action stream (oracle/philox_ref.py), re-arming finished envs the way
mt_rollout_fused(auto_reset) / mt_reset_done do (return -> ring, episode + 1, zero pose, fresh targets keyed by
(env, new episode)).  Reference semantics: manytor.py:255-260 per step, test_multi.py:17-34 for the loop.

An env is compared only while the oracle's own decision margins (z = 0 of the ground flag, |delta| = tol of a pickup)
stay above GUARD at EVERY step of the rollout: inside the band fp32 and fp64 may legitimately decide differently, and
in a fused rollout the two sides cannot be re-synchronised mid-way.  The clean fraction of every case is measured,
recorded and asserted against that measurement (minus two points).
"""
import numpy as np
import pytest

from parity_util import GUARD, POS_TOL, assert_obs_close

pytestmark = pytest.mark.gpu

RING = 4


@pytest.fixture(scope="module")
def m():
    import manytor_amd
    if manytor_amd.device_count() < 1:
        pytest.fail("gpu tests need a visible MI355X and the in-tree libmanytor_hip.so")
    return manytor_amd


def _tables(m):
    rng = np.random.RandomState(55)
    rt5 = np.column_stack([rng.uniform(0, 9, 5), rng.choice([-np.pi / 2, 0.3, np.pi / 2], 5), rng.uniform(2, 12, 5),
                           np.zeros(5)])
    return {"ref": (m.REF_DH_TABLE, 51.3), "dh7": (m.DH7_TABLE, 92.6), "rt5": (rt5, 40.0)}


def record_clean(*row):
    """The measured clean fraction of every case goes to gpurun_out/ (scratch): the floors asserted below are these
    measurements minus two points, not a guess."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "fused_clean_fraction.jsonl")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "a") as f:
            f.write(json.dumps(dict(zip(("table", "n", "k", "T", "auto_reset", "tol", "clean", "touching_env_steps",
                                         "env_steps_behind_a_touch"), row))) + "\n")
    except OSError:
        pass
    print(f"[fused-vs-oracle] {row[0]} n={row[1]} K={row[2]} T={row[3]} auto_reset={row[4]} tol={row[5]}: "
          f"clean fraction {row[6]:.4f}, env-steps inside the guard band {row[7]:.5f}, env-steps behind an env's first touch "
          f"(EE check only) {row[8]:.4f}")


def run_fused_against_oracle(m, n, k, table_name, T, auto_reset, tol, seed, min_clean):
    from oracle import c_oracle
    from oracle import philox_ref as px
    table, radius = _tables(m)[table_name]
    dof = len(table)
    eng = m.StepEngine(n, k, dh_table=table, radius=radius, pickup_tol=tol, return_ring=RING)
    ora = c_oracle.COracle(n, k, table=np.asarray(table), radius=radius, pickup_tol=tol, threads=16)
    ids = np.arange(n, dtype=np.uint64)
    eng.reset_random(seed, 0)
    ora.reset(eng.points().astype(np.float64))
    eng.rollout_fused(T, seed, 0, auto_reset=auto_reset)          # ONE launch on the GPU

    clean = np.ones(n, dtype=bool)
    touches = 0                                                    # env-steps whose own decision sat inside the guard band
    behind = 0                                                     # env-steps of envs that had touched it before (or just did)
    episodes = np.zeros(n, dtype=np.int64)
    last_ret = np.zeros(n)
    ring = np.zeros((n, RING))
    last = None
    for t in range(T):
        act = px.sample_actions(seed, ids, t, dof).astype(np.float64)
        pre_alive = ora.alives.copy()
        obs_ref, rew_ref, done_ref = ora.step(act)
        pm = np.where(pre_alive, ora.pickup_margin, np.inf).min(axis=1)
        risky = (ora.ground_margin < GUARD) | (pm < GUARD)
        touches += int(risky.sum())
        clean &= ~risky
        behind += int((~clean).sum())
        if t == T - 1:
            last = dict(obs=obs_ref, rew=rew_ref, done=done_ref, jc=ora.joints_coordinates.copy(), pre_alive=pre_alive,
                        points=ora.points.copy())
        if auto_reset and done_ref.any():                          # what the kernel does in the step an env finishes
            idx = np.flatnonzero(done_ref)
            ring[idx, episodes[idx] % RING] = ora.total_reward[idx]
            last_ret[idx] = ora.total_reward[idx]
            episodes[idx] += 1
            ora.goals[idx] = 0.0
            ora.total_reward[idx] = 0.0
            ora.alive_u8[idx] = 1
            ora.points[idx] = px.sample_targets(seed, ids[idx], episodes[idx], k, radius).astype(np.float64)

    c = clean
    record_clean(table_name, n, k, T, auto_reset, tol, float(c.mean()), touches / (n * T), behind / (n * T))
    assert c.mean() >= min_clean, c.mean()
    # state after the rollout
    np.testing.assert_array_equal(eng.goals()[c], ora.goals[c].astype(np.float32))
    np.testing.assert_array_equal(eng.alives()[c], ora.alives[c])
    np.testing.assert_array_equal(eng.total_reward()[c], ora.total_reward[c].astype(np.float32))
    np.testing.assert_array_equal(eng.points()[c], ora.points[c].astype(np.float32))
    np.testing.assert_array_equal(eng.episodes()[c], episodes[c])
    np.testing.assert_array_equal(eng.finished()[c], episodes[c])
    if auto_reset:
        fin = c & (episodes > 0)
        np.testing.assert_array_equal(eng.last_return()[fin], last_ret[fin].astype(np.float32))
        got = eng.return_ring()
        for e in range(1, RING + 1):                               # slots an env has actually written
            sel = c & (episodes >= e)
            np.testing.assert_array_equal(got[sel, e - 1], ring[sel, e - 1].astype(np.float32))
    # outputs of the last step
    np.testing.assert_array_equal(eng.reward()[c], last["rew"][c])
    done_raw = eng.get(m.lib.F_DONE)
    np.testing.assert_array_equal(done_raw[c] != 0, last["done"][c])
    if auto_reset:
        assert set(np.unique(done_raw)) <= {0, 2}                  # finished in the last step = already re-armed
    else:
        assert set(np.unique(done_raw)) <= {0, 1}
    assert np.abs(eng.ee() - last["jc"][:, -1]).max() <= POS_TOL    # continuous: every env, guard band or not
    assert_obs_close(eng.obs()[c], last["obs"][c], last["jc"][c, -2], last["points"][c], last["pre_alive"][c])
    return dict(clean=float(c.mean()), finished=int((episodes > 0).sum()), max_episodes=int(episodes.max()))


# `floor` = the clean fraction MEASURED for the case (round 3, profiles/r03_fused_clean_fraction.jsonl; the inputs are
# seeded, so it is reproducible) minus two points: the comparison covers this share of the envs, stated, not guessed.
@pytest.mark.parametrize("table_name,T,auto_reset,tol,floor", [
    ("ref", 50, False, 8.0, 0.9075),      # BASELINE.json configs[2]: one full 50-step episode (test_multi.py:8); measured 0.9275
    ("ref", 50, True, 20.0, 0.9314),      # wide pickup box: many envs finish (several times) inside the launch; 0.9514
    ("dh7", 25, False, 8.0, 0.9536),      # configs[4]; 0.9736
    ("dh7", 25, True, 45.0, 0.9647),      # 0.9847
])
def test_fused_rollout_every_env_against_the_c_oracle_full_size(m, table_name, T, auto_reset, tol, floor):
    stats = run_fused_against_oracle(m, 1048576, 7, table_name, T, auto_reset, tol, seed=0xF00D + T, min_clean=floor)
    if auto_reset:
        assert stats["finished"] > 1000, stats                     # the re-arm path was really exercised


@pytest.mark.parametrize("n,table_name,auto_reset,tol,kernel,floor", [
    (65536, "ref", False, 8.0, "L=2", 0.9401),      # BASELINE.json configs[1]: rollout_split_kernel<Ref4Table, 2>; measured 0.9601
    (65536, "dh7", True, 45.0, "L=2", 0.9649),      # 0.9849
    (131072, "ref", True, 20.0, "pf=8", 0.9532),    # the per-GPU shard of 1 M arms over 8 GPUs: rollout_split_kernel<.., 2> (step kernel: one env per lane); 0.9732
    (131072, "dh7", False, 8.0, "pf=8", 0.9530),    # 0.9730
    (32768, "ref", True, 20.0, "L=4", 0.9542),      # rollout_split_kernel<.., 4>; 0.9742
])
def test_fused_rollout_every_env_against_the_c_oracle_at_each_dispatch_regime(m, n, table_name, auto_reset, tol, kernel, floor):
    probe = m.StepEngine(n, 7, dh_table=_tables(m)[table_name][0], radius=_tables(m)[table_name][1])
    # the regime, named by its step kernel (the fused rollout spreads an env over 4 lanes up to 32 768 arms and over 2 up to
    # 131 072; bit-identical to the one-lane kernel either way: test_rollout_kernel_variants_are_bit_identical)
    assert kernel in probe.step_kernel_name(), probe.step_kernel_name()
    probe.close()
    stats = run_fused_against_oracle(m, n, 7, table_name, 25, auto_reset, tol, seed=0xD15 + n, min_clean=floor)
    if auto_reset:
        assert stats["finished"] > 100, stats


@pytest.mark.parametrize("n,k,table_name,T,auto_reset,tol,floor", [
    (100003, 3, "ref", 40, True, 25.0, 0.9475),       # ragged size (prime), few targets: envs finish up to several times; measured 0.9675
    (65537, 7, "rt5", 12, False, 8.0, 0.9572),        # runtime-table kernel, one past a power of two; 0.9772
    (777, 32, "ref", 30, True, 30.0, 0.9247),         # K = 32: the 96 KB LDS tile of the fused kernel; 0.9447
])
def test_fused_rollout_against_the_c_oracle_ragged_and_generic(m, n, k, table_name, T, auto_reset, tol, floor):
    stats = run_fused_against_oracle(m, n, k, table_name, T, auto_reset, tol, seed=31 + n, min_clean=floor)
    if auto_reset and k <= 3:
        assert stats["max_episodes"] >= 2, stats


def test_ring_keeps_every_return_when_an_env_finishes_several_times_in_one_launch(m):
    """with one last_return slot an env that finishes twice in a launch loses a return; the ring
    keeps the last R.  Cross-checked against the launch-per-step path (step_random + reset_done)."""
    n, k, T = 20000, 1, 60
    a = m.StepEngine(n, k, pickup_tol=30.0, return_ring=8)
    b = m.StepEngine(n, k, pickup_tol=30.0, return_ring=8)
    for e in (a, b):
        e.reset_random(12, 7)                                      # full reset at episode 7: counts restart there
    returns = [[] for _ in range(n)]
    for t in range(T):
        a.step_random(12, t)
        done = np.flatnonzero(a.done())
        tot = a.total_reward()
        for i in done:
            returns[i].append(tot[i])
        a.reset_done(12)
    b.rollout_fused(T, 12, 0, auto_reset=True)
    fin = b.finished()
    np.testing.assert_array_equal(fin, [len(r) for r in returns])
    np.testing.assert_array_equal(b.episodes(), 7 + fin)
    assert fin.max() >= 3
    ring = b.return_ring()
    np.testing.assert_array_equal(ring, a.return_ring())
    for i in np.flatnonzero(fin > 0)[:2000]:
        for c, r in list(enumerate(returns[i]))[-8:]:
            assert ring[i, c % 8] == r
    # a reset_done after an auto-reset rollout must not re-arm anybody a second time (done == 2 = already re-armed)
    before = {f: b.get(getattr(m.lib, f)) for f in ("F_GOALS", "F_POINTS", "F_EPISODES", "F_TOTAL_REWARD", "F_LAST_RETURN",
                                                     "F_RETURN_RING", "F_ALIVE")}
    assert (b.get(m.lib.F_DONE) == 2).any()
    b.reset_done(12)
    for f, v in before.items():
        np.testing.assert_array_equal(b.get(getattr(m.lib, f)), v, err_msg=f)
    assert not b.get(m.lib.F_DONE).any() and not b.done_bits().any()


def test_config3_four_million_arms_one_handle_equals_eight_shards(m):
    """BASELINE.json configs[3]: 4 194 304 arms sharded over 8 GPUs = 8 x 524 288.  On the one GPU of this box: one
    handle owning all of them vs 8 handles with env_id_base = r * 524 288 (exactly what rank r creates) must agree
    bit for bit after 3 per-step launches + one fused segment; then every env of one more step against the C oracle."""
    from oracle import c_oracle
    n, k, world = 4194304, 7, 8
    shard = n // world
    whole = m.StepEngine(n, k)
    whole.reset_random(0x5EED, 0)
    whole.rollout(3, 0x5EED, 0)
    whole.rollout_fused(5, 0x5EED, 3)
    fields = ("F_GOALS", "F_ALIVE", "F_TOTAL_REWARD", "F_OBS", "F_REWARD", "F_DONE", "F_EE", "F_POINTS")
    ref = {f: whole.get(getattr(m.lib, f)) for f in fields}
    for r in range(world):
        part = m.StepEngine(shard, k, env_id_base=r * shard)
        part.reset_random(0x5EED, 0)
        part.rollout(3, 0x5EED, 0)
        part.rollout_fused(5, 0x5EED, 3)
        for f in fields:
            np.testing.assert_array_equal(part.get(getattr(m.lib, f)), ref[f][r * shard:(r + 1) * shard], err_msg=f"{f} rank {r}")
        part.close()
    del ref
    # every env of the 4 M batch against the oracle for one more step
    ora = c_oracle.COracle(n, k, threads=16)
    ora.goals[:] = whole.goals()
    ora.points[:] = whole.points()
    ora.alive_u8[:] = whole.get(m.lib.F_ALIVE)
    ora.total_reward[:] = whole.total_reward()
    pre_alive = ora.alives.copy()
    whole.step_random(0x5EED, 8)
    obs_ref, rew_ref, done_ref = ora.step(whole.goals().astype(np.float64))
    assert np.abs(whole.ee() - ora.joints_coordinates[:, -1]).max() <= POS_TOL
    pm = np.where(pre_alive, ora.pickup_margin, np.inf).min(axis=1)
    ok = ~((ora.ground_margin < GUARD) | (pm < GUARD))
    assert ok.mean() > 0.995
    np.testing.assert_array_equal(whole.reward()[ok], rew_ref[ok])
    np.testing.assert_array_equal(whole.done()[ok], done_ref[ok])
    np.testing.assert_array_equal(whole.alives()[ok], ora.alives[ok])
    np.testing.assert_array_equal(whole.total_reward()[ok], ora.total_reward[ok].astype(np.float32))
    assert_obs_close(whole.obs(), obs_ref, ora.joints_coordinates[:, -2], ora.points, pre_alive)
