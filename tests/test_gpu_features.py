"""GPU tests of the boundary features around the step path: stream ordering with torch, unusable actions, single-env
step / reset inside a batch (multienv.environment[i], manytor.py:82,118), the device-side sub-step trace (manytor.py:190)
and the C-ABI return gather (RCCL with one rank; a device copy without a communicator)."""
import numpy as np
import pytest

from parity_util import DIST_TOL, POS_TOL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def m():
    import manytor_amd
    if manytor_amd.device_count() < 1:
        pytest.fail("gpu tests need a visible MI355X and the in-tree libmanytor_hip.so")
    return manytor_amd


def test_one_hip_runtime_in_the_process(m):
    """torch bundles its own libamdhip64; the package loads that copy first so that torch and the engine share
    streams, events and allocations whatever the import order (manytor_amd/_lib.py)."""
    import torch  # noqa: F401
    with open("/proc/self/maps") as f:
        libs = {line.split()[-1] for line in f if "libamdhip64" in line}
    assert len(libs) == 1, libs


def test_torch_reads_are_ordered_after_queued_engine_work(m):
    """with use_torch_stream() the engine launches on torch's current stream -- also when that is
    the default stream (handle 0) -- so a torch read of a zero-copy view sees everything queued before it."""
    import torch
    n, k = 1048576, 7
    eng = m.StepEngine(n, k)
    eng.use_torch_stream()
    eng.reset_random(3, 0)
    eng.rollout(200, 3, 0)                                   # ~8 ms of queued kernels, no sync
    view = eng.device_tensor(m.lib.F_TOTAL_REWARD)
    snap = view.clone()                                      # torch op on the same stream: must run after the rollout
    eng.reset_random(3, 1)                                   # zeroes the returns; must run after the clone
    got = snap.cpu().numpy()
    ref = m.StepEngine(n, k)
    ref.reset_random(3, 0)
    ref.rollout(200, 3, 0)
    np.testing.assert_array_equal(got, ref.total_reward())
    assert np.abs(got).max() > 0
    np.testing.assert_array_equal(eng.total_reward(), np.zeros(n, dtype=np.float32))
    # and on a side stream
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        eng.use_torch_stream()
        eng.rollout(100, 3, 0)
        snap2 = eng.device_tensor(m.lib.F_TOTAL_REWARD).clone()
    side.synchronize()
    ref.reset_random(3, 1)
    ref.rollout(100, 3, 0)
    np.testing.assert_array_equal(snap2.cpu().numpy(), ref.total_reward())
    eng.set_stream(None)                                     # back to the private stream
    eng.rollout(3, 3, 100)
    eng.sync()


def test_unusable_actions_hold_the_pose_and_are_counted(m):
    """A NaN / inf / absurd action (a diverging policy) must not poison `goals` for good."""
    n, k = 512, 3
    eng = m.StepEngine(n, k)
    pts = np.random.RandomState(0).uniform(-30, 30, size=(n, k, 3)).astype(np.float32)
    pts[..., 2] = np.abs(pts[..., 2])
    eng.reset(pts)
    first = np.random.RandomState(1).randint(-90, 90, size=(n, 4)).astype(np.float32)
    eng.step(first)
    assert eng.bad_action_count() == 0
    ref = m.StepEngine(n, k)
    ref.reset(pts)
    ref.step(first)
    act = np.random.RandomState(2).randint(-90, 90, size=(n, 4)).astype(np.float32)
    bad = act.copy()
    bad[3, 1] = np.nan
    bad[77, 0] = np.inf
    bad[200, 3] = -np.inf
    bad[301, 2] = 1e30
    hold = act.copy()
    for i in (3, 77, 200, 301):
        hold[i] = first[i]                                   # expected behaviour: the env keeps its pose
    eng.step(bad)
    ref.step(hold)
    assert eng.bad_action_count() == 4
    for f in ("F_GOALS", "F_OBS", "F_REWARD", "F_DONE", "F_ALIVE", "F_EE", "F_TOTAL_REWARD"):
        np.testing.assert_array_equal(eng.get(getattr(m.lib, f)), ref.get(getattr(m.lib, f)), err_msg=f)
    assert np.isfinite(eng.goals()).all() and np.isfinite(eng.obs()).all()
    big_ok = act.copy()
    big_ok[5] = [720.0, -1080.0, 32768.0, -32768.0]          # large but usable: whole turns, same pose as 0
    eng.step(big_ok)
    assert eng.bad_action_count() == 4


def test_single_env_step_and_reset_inside_a_batch(m):
    """multienv.environment[i].step(a) / .reset() in the reference touch only env i (manytor.py:82,118)."""
    n, k = 300, 5
    rng = np.random.RandomState(4)
    pts = rng.uniform(-30, 30, size=(n, k, 3)).astype(np.float32)
    pts[..., 2] = np.abs(pts[..., 2])
    a, b = m.StepEngine(n, k, pickup_tol=25.0), m.StepEngine(n, k, pickup_tol=25.0)
    a.reset(pts)
    b.reset(pts)
    fields = ("F_GOALS", "F_OBS", "F_REWARD", "F_DONE", "F_ALIVE", "F_EE", "F_TOTAL_REWARD", "F_POINTS", "F_DONE_BITS")
    acts = rng.randint(-180, 180, size=(n, 4)).astype(np.float32)
    a.step(acts)                                             # batch step
    for i in range(n):                                       # the same, one env at a time
        obs, rew, done = b.env_step(i, acts[i])
        assert obs.shape == (3 * k,) and isinstance(rew, int) and isinstance(done, bool)
    for f in fields:
        np.testing.assert_array_equal(a.get(getattr(m.lib, f)), b.get(getattr(m.lib, f)), err_msg=f)
    # stepping one env leaves the others alone
    before = {f: b.get(getattr(m.lib, f)) for f in fields}
    obs, rew, done = b.env_step(129, [10, 20, 30, 40])
    after = {f: b.get(getattr(m.lib, f)) for f in fields}
    others = np.arange(n) != 129
    for f in fields[:-1]:
        np.testing.assert_array_equal(after[f][others], before[f][others], err_msg=f)
    np.testing.assert_array_equal(after["F_GOALS"][129], [10, 20, 30, 40])
    np.testing.assert_array_equal(obs, after["F_OBS"][129])
    bits = after["F_DONE_BITS"]
    unpacked = ((bits[:, None] >> np.arange(64, dtype=np.uint64)) & np.uint64(1)).astype(bool).ravel()[:n]
    np.testing.assert_array_equal(unpacked, after["F_DONE"].astype(bool))
    # reset of one env: given targets, and device-drawn ones
    from oracle import philox_ref as px
    newp = rng.uniform(0, 20, size=(k, 3)).astype(np.float32)
    b.env_reset(7, newp)
    np.testing.assert_array_equal(b.points()[7], newp)
    assert np.all(b.goals()[7] == 0) and b.total_reward()[7] == 0 and b.alives()[7].all()
    b.env_reset(8, None, seed=99, episode=5)
    np.testing.assert_array_equal(b.points()[8], px.sample_targets(99, np.array([8], dtype=np.uint64), 5, k, 51.3)[0])
    keep = np.ones(n, dtype=bool)
    keep[[7, 8]] = False
    np.testing.assert_array_equal(b.points()[keep], after["F_POINTS"][keep])
    np.testing.assert_array_equal(b.goals()[keep], after["F_GOALS"][keep])
    with pytest.raises(ValueError):
        b.env_step(n, [0, 0, 0, 0])


def test_multienv_environment_items_have_the_reference_methods(m, golden):
    """`multienv.environment[i]` is an Environment in the reference (manytor.py:82): step / reset / action /
    action_sample / is_done / get_observations work per env."""
    g = golden("f5_semantics_kat")
    pts = g["all_picked__points_in"]
    np.random.seed(0)
    me = m.Multienv((2, 2), len(pts))
    me.reset()
    e2 = me.environment[2]
    e2.points = pts
    np.testing.assert_allclose(e2.points, pts, atol=1e-5)
    assert e2.is_done() is False
    obs2, reward, done = e2.step([30, 45, -60, 90])
    assert (reward, done) == (int(g["all_picked__reward"][0]), bool(g["all_picked__done"][0]))
    np.testing.assert_allclose(obs2[0::3], g["all_picked__obs2"][0][0::3], atol=DIST_TOL)
    assert e2.total_reward == 1.0 and me.environment[1].total_reward == 0.0
    assert np.all(me.environment[1].goals == 0)              # the neighbours did not move
    a = e2.action_sample()
    assert len(a) == 4 and all(-180 <= int(v) < 180 for v in a)
    reward2, obs3 = e2.action([10, 10, 10, 10], None)
    assert reward2 == 0 and e2.total_reward == 1.0           # action() does not accumulate (manytor.py:258 is in step)
    e2.reset()
    assert e2.total_reward == 0.0 and e2.alives.all() and np.all(e2.goals == 0)
    assert e2.get_observations().shape == (3 * len(pts),)


def test_device_trace_buffer_matches_reference_substeps(m, golden):
    """MT_F_TRACE: the end effector at each of the 25 sub-step poses, for the whole batch, vs
    fixture F3 (the rows the reference appends to `trajectory`, manytor.py:190)."""
    g = golden("f3_substep_trace")
    n = len(g["prev"])
    eng = m.StepEngine(n, 1, trace=True)
    eng.reset(np.full((n, 1, 3), 1000.0, dtype=np.float32))
    eng.set(m.lib.F_GOALS, g["prev"].astype(np.float32))
    eng.step(g["action"])
    tr = eng.trace()
    assert tr.shape == (n, 25, 3)
    assert np.abs(tr - g["jc"][:, :, 3, :]).max() <= POS_TOL
    assert np.abs(tr[:, -1] - eng.ee()).max() <= 2e-5
    # the trace does not change the step's results
    plain = m.StepEngine(n, 1)
    plain.reset(np.full((n, 1, 3), 1000.0, dtype=np.float32))
    plain.set(m.lib.F_GOALS, g["prev"].astype(np.float32))
    plain.step(g["action"])
    for f in ("F_GOALS", "F_OBS", "F_REWARD", "F_EE", "F_TOTAL_REWARD"):
        np.testing.assert_array_equal(eng.get(getattr(m.lib, f)), plain.get(getattr(m.lib, f)), err_msg=f)
    with pytest.raises(ValueError):                           # a handle created without MT_FLAG_TRACE has no such field
        plain.trace()
    # random-action path and a fused request (falls back to per-step launches so that the trace stays complete)
    big = m.StepEngine(5000, 3, trace=True)
    big.reset_random(1, 0)
    prev = big.goals()
    big.rollout_fused(2, 1, 0)
    last_prev = big.goals().copy()
    big.step_random(1, 2)
    tr = big.trace()
    ref = m.route_trace(last_prev[:64], big.goals()[:64])[:, :, -1, :]
    assert np.abs(tr[:64] - ref).max() <= 2e-5
    assert prev.shape == last_prev.shape


def test_gather_returns_without_and_with_a_one_rank_rccl_communicator(m):
    """mt_gather_returns: device copy without a communicator; with mt_comm_init(world = 1 ... RCCL really loaded and
    a communicator really built) the same values.  Multi-rank behaviour is covered by the gloo world-2 CPU test of
    the host logic; an 8-GPU node is not available to this box."""
    import torch
    n = 70001
    eng = m.StepEngine(n, 7, return_ring=2)
    eng.reset_random(2, 0)
    eng.rollout(5, 2, 0)
    out = eng.gather_returns()
    eng.sync()
    ref = eng.total_reward()
    np.testing.assert_array_equal(out.cpu().numpy(), ref)
    assert eng.total_envs() == n
    uid = m.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    eng.comm_init(uid, 0, 1)
    assert eng.total_envs() == n
    out2 = torch.zeros(n, dtype=torch.float32, device="cuda")
    eng.gather_returns(out2)
    eng.gather_returns(out, field=m.lib.F_LAST_RETURN)
    eng.sync()
    np.testing.assert_array_equal(out2.cpu().numpy(), ref)
    np.testing.assert_array_equal(out.cpu().numpy(), eng.last_return())
    with pytest.raises(ValueError):
        eng.gather_returns(torch.zeros(n - 1, dtype=torch.float32, device="cuda"))
    # mt_reduce_returns: the summary a learner logs, through the one-rank communicator and (below) without one
    st = eng.return_stats()
    assert st["count"] == n and st["sum"] == float(ref.astype(np.float64).sum())
    assert st["min"] == float(ref.min()) and st["max"] == float(ref.max()) and st["done"] == int(eng.done().sum())
    assert eng.return_stats(field=m.lib.F_LAST_RETURN)["sum"] == float(eng.last_return().astype(np.float64).sum())
    # overlapped form, through the communicator: snapshot on the engine's stream, exchange on its side stream; the
    # reset and the steps queued right behind it must not leak into the result
    for rep in range(3):
        want = eng.total_reward()
        out3 = eng.gather_begin(out3 if rep else None)
        eng.reset_random(2, rep + 1)                         # zeroes the returns while the exchange may still run
        eng.rollout(4 + rep, 2, 0)
        ms = eng.gather_wait(host=True)
        assert ms is not None and ms >= 0.0
        np.testing.assert_array_equal(out3.cpu().numpy(), want)
        assert np.abs(want).max() > 0
    out4 = eng.gather_begin()                                # ... and sync() alone also covers a begun gather
    want = eng.total_reward()
    eng.reset_random(2, 9)
    eng.sync()
    np.testing.assert_array_equal(out4.cpu().numpy(), want)
    eng.comm_destroy()
    # without a communicator the overlapped form is a device copy on the side stream
    assert eng.return_stats() == st | {"sum": 0.0, "min": 0.0, "max": 0.0, "mean": 0.0, "done": 0}    # after reset_random(2, 9)
    out5 = eng.gather_begin()
    eng.rollout(3, 2, 0)
    eng.gather_wait(host=True)
    np.testing.assert_array_equal(out5.cpu().numpy(), np.zeros(n, dtype=np.float32))
    eng.close()


def test_bench_contingency_gather_through_a_torch_process_group(m):
    """bench.py's fallback when mt_comm_init is unavailable on a node (manytor_amd.distributed.attach_torch_gather):
    a one-rank nccl (= RCCL) process group of torch.distributed gathers straight from the arena view, ordered with the
    engine's launches on torch's stream."""
    import socket

    import torch
    import torch.distributed as dist

    from manytor_amd import distributed as D
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        n = 30011
        eng = m.StepEngine(n, 7)
        D.attach_torch_gather(eng, n, 0, 1)
        eng.reset_random(3, 0)
        eng.rollout(6, 3, 0)
        out = eng.gather_begin()
        want = eng.total_reward()
        eng.reset_random(3, 1)                               # same stream: ordered behind the collective
        assert eng.gather_wait(host=True) == 0.0
        torch.cuda.synchronize()
        np.testing.assert_array_equal(out.cpu().numpy(), want)
        assert np.abs(want).max() > 0 and eng.total_envs() == n
        with pytest.raises(ValueError):
            D.attach_torch_gather(m.StepEngine(100, 7), 300, 0, 2)       # shard size does not match the layout
    finally:
        dist.destroy_process_group()


def test_checkpoint_and_resume_is_bit_identical(m):
    """SURVEY 5 (checkpoint / resume): the reference's state is a handful of attributes (manytor.py:131-139);
    get_state() / set_state() carry the same on and off the device, also into a fresh engine."""
    n, k = 50000, 7
    a = m.StepEngine(n, k)
    a.reset_random(4, 0)
    a.rollout(9, 4, 0)
    snap = a.get_state()
    a.rollout(6, 4, 9)
    want = a.get_state()
    b = m.StepEngine(n, k)                                   # a fresh engine, never reset
    b.set_state(snap)
    b.rollout(6, 4, 9)
    got = b.get_state()
    for key in ("goals", "points", "alives", "total_reward", "obs", "reward", "done", "ee"):
        np.testing.assert_array_equal(got[key], want[key], err_msg=key)
    a.set_state(snap)                                        # and back in time on the same engine
    a.rollout(6, 4, 9)
    np.testing.assert_array_equal(a.obs(), want["obs"])


def test_checkpoint_restores_done_flags_episode_counters_and_the_ring(m):
    """a checkpoint taken while envs are finished (done = 1), re-armed in-kernel (done = 2) or several
    episodes in must resume reset_done() / finished() / return_ring() exactly as the original run would."""
    n, k = 20000, 1
    a = m.StepEngine(n, k, pickup_tol=30.0, return_ring=4)
    a.reset_random(12, 7)                                    # episode base 7
    a.rollout_fused(20, 12, 0, auto_reset=True)              # envs finish, some several times; some end with done == 2
    a.rollout(3, 12, 20)                                     # ... and some are finished-and-waiting (done == 1)
    snap = a.get_state()
    assert (snap["done"] == 1).any() and snap["episodes"].max() >= 9 and snap["episode0"] == 7

    def carry_on(e):
        e.reset_done(12)
        e.rollout(4, 12, 23)
        e.reset_done(12)
        e.rollout_fused(6, 12, 27, auto_reset=True)
        return {**e.get_state(), "finished": e.finished(), "done_bits": e.done_bits()}
    want = carry_on(a)
    b = m.StepEngine(n, k, pickup_tol=30.0, return_ring=4)   # fresh engine, never reset
    b.set_state(snap)
    np.testing.assert_array_equal(b.get(m.lib.F_DONE), snap["done"])
    bits = b.done_bits()
    unpacked = ((bits[:, None] >> np.arange(64, dtype=np.uint64)) & np.uint64(1)).astype(bool).ravel()[:n]
    np.testing.assert_array_equal(unpacked, snap["done"] != 0)           # ballot words rebuilt from the bytes
    got = carry_on(b)
    for key, v in want.items():
        np.testing.assert_array_equal(got[key], v, err_msg=key)
    assert want["finished"].max() >= 3


def test_mt_set_screens_non_finite_values(m):
    """the kernels are built with -ffinite-math-only, so NaN / inf must not reach the arena through
    attribute assignment either; joint angles beyond +-32768 degrees are refused like staged actions are."""
    e = m.StepEngine(8, 2)
    e.reset_random(1, 0)
    goals, pts = e.goals(), e.points()
    for field, good, bad_value in ((m.lib.F_GOALS, goals, np.nan), (m.lib.F_GOALS, goals, 1e6), (m.lib.F_POINTS, pts, np.inf),
                                   (m.lib.F_TOTAL_REWARD, e.total_reward(), -np.inf)):
        bad = good.copy()
        bad.flat[3] = bad_value
        with pytest.raises(ValueError):
            e.set(field, bad)
        np.testing.assert_array_equal(e.get(field), good)    # nothing was written
    with pytest.raises(ValueError):
        e.reset(np.full((8, 2, 3), np.nan, dtype=np.float32))


def test_single_env_reset_draws_fresh_targets_and_leaves_the_counters_alone(m):
    """environment[i].reset() with the device RNG must give NEW targets at every call (manytor.py:229 draws
    anew), must not make finished() count an episode, and must not collide with the next whole-batch reset."""
    me = m.Multienv((4, 4), 3, rng="device", seed=77)
    me.reset()
    e5 = me.environment[5]
    p0 = e5.points.copy()
    e5.reset()
    p1 = e5.points.copy()
    e5.reset()
    p2 = e5.points.copy()
    assert not np.array_equal(p0, p1) and not np.array_equal(p1, p2) and not np.array_equal(p0, p2)
    for p in (p1, p2):
        assert (p[:, 2] >= 0).all() and (np.linalg.norm(p, axis=1) <= 51.3 * (1 + 1e-6)).all()
    assert not me.engine.finished().any()                    # nobody finished an episode
    others = np.arange(16) != 5
    before = me.engine.points()
    me.reset()                                               # next whole-batch reset: env 5 gets the batch's episode key
    from oracle import philox_ref as px
    np.testing.assert_array_equal(me.engine.points(), px.sample_targets(77, np.arange(16, dtype=np.uint64), 1, 3, 51.3))
    assert not np.array_equal(me.engine.points()[others], before[others])


def test_environment_keeps_its_pose_mirror_when_the_kernel_rejects_an_action(m):
    env = m.Environment(3)
    np.random.seed(1)
    env.reset()
    env.step([10, 20, 30, 40])
    env.step([np.nan, 0, 0, 0])                              # rejected: the arm holds [10, 20, 30, 40]
    np.testing.assert_array_equal(env.goals, [10, 20, 30, 40])
    np.testing.assert_array_equal(env._pose[0], [10, 20, 30, 40])
    env.step([0, 0, 0, 0])
    tr = env.trajectory                                      # 1 seed row + 3 routes x 25 sub-steps
    assert tr.shape == (76, 3)
    np.testing.assert_allclose(tr[26:51], np.repeat(tr[25:26], 25, axis=0), atol=1e-5)   # the held step did not move


def test_writing_goals_through_the_raw_pointer_is_seen_by_the_sampled_step(m):
    """The sampled-action kernels look the sines / cosines of the pose a step STARTS from up in a whole-degree table while
    the host knows every angle to be a whole degree.  A caller who takes the raw MT_F_GOALS pointer (torch view) can write
    fractional angles behind the library's back: handing the pointer out must end that assumption -- the next sampled
    step has to start from exactly the pose that was written."""
    import torch
    from oracle import c_oracle
    n, k = 4096, 2
    e = m.StepEngine(n, k, debug_zmin=True)
    e.reset_random(5, 0)
    e.step_random(5, 0)                                       # whole-degree poses, table path
    g = e.device_tensor(m.lib.F_GOALS)                        # (D, N) view of the resident rows
    frac = torch.rand(g.shape, device=g.device) * 300.0 - 150.0
    e.sync()
    g.copy_(frac)
    torch.cuda.synchronize()
    prev = e.goals().astype(np.float64)
    np.testing.assert_array_equal(prev, frac.T.cpu().numpy().astype(np.float64))
    e.step_random(5, 1)
    ora = c_oracle.COracle(n, k, threads=4)
    ora.reset(e.points().astype(np.float64))
    ora.goals[:] = prev
    ora.step(e.goals().astype(np.float64))
    assert np.abs(e.zmin() - ora.zmin).max() <= 1e-4          # the route really started at the fractional pose
    assert np.abs(e.ee() - ora.joints_coordinates[:, -1]).max() <= 1e-4


@pytest.mark.parametrize("chains", [2, 3])
def test_chains_stay_forked_across_reset_and_rollout_and_join_where_they_must(m, monkeypatch, chains):
    """On its own stream a multi-chain handle does not join after mt_rollout / mt_reset_random: each env range is reset
    right behind its own last step and the next segment continues per chain.  Every other call folds the chains back
    first (MT_ENTER): the in-place gather of MT_F_LAST_RETURN begun after a reset must deliver the finished returns of
    EVERY range while the next episode is already running, getters and reset_done must see complete state, and the
    whole sequence must equal the single-chain engine bit for bit."""
    import torch
    n, k = 300003, 3
    monkeypatch.setenv("MT_CHAINS", "1")
    ref = m.StepEngine(n, k, pickup_tol=20.0)
    monkeypatch.setenv("MT_CHAINS", str(chains))
    eng = m.StepEngine(n, k, pickup_tol=20.0)
    assert f"{chains} chains" in eng.step_kernel_name()
    returns = {}
    for e in (ref, eng):
        e.reset_random(8, 0)
        e.rollout(7, 8, 0)                                   # forked from here on (eng)
        e.reset_random(8, 1)                                 # per chain, no join
        buf = e.gather_begin(field=m.lib.F_LAST_RETURN, snapshot=False)      # joins; exchange on the side stream
        e.rollout(5, 8, 0)                                   # next episode beside the exchange
        e.reset_random(8, 2)                                 # must wait for the exchange: it overwrites MT_F_LAST_RETURN
        e.rollout(4, 8, 0)
        e.gather_wait(host=True)
        returns[e] = buf.cpu().numpy().copy()
        e.reset_done(8)                                      # a whole-batch call right behind per-chain work
        e.rollout(3, 8, 4)
    np.testing.assert_array_equal(returns[eng], returns[ref])
    assert np.abs(returns[ref]).max() > 0                    # the 7-step episode's returns, not the zeros after the reset
    for f in ("F_GOALS", "F_ALIVE", "F_TOTAL_REWARD", "F_POINTS", "F_EPISODES", "F_LAST_RETURN", "F_OBS", "F_REWARD", "F_DONE",
              "F_EE", "F_DONE_BITS"):
        np.testing.assert_array_equal(eng.get(getattr(m.lib, f)), ref.get(getattr(m.lib, f)), err_msg=f)
    # sync() is a join point too: a torch read of a view after it sees every range
    eng.rollout(6, 8, 7)
    ref.rollout(6, 8, 7)
    eng.sync()
    view = eng.device_tensor(m.lib.F_TOTAL_REWARD).clone()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(view.cpu().numpy(), ref.total_reward())


def test_lap_timer_covers_every_chain(m, monkeypatch):
    """A lap that ends while the chains are forked ends when the LAST chain is done (one end event per stream), so a
    2-chain lap over T steps reads like the same work timed with a full join -- not like half of it."""
    n, k, T = 1048576, 7, 20
    monkeypatch.setenv("MT_CHAINS", "2")
    e = m.StepEngine(n, k)
    e.reset_random(1, 0)
    for _ in range(5):
        e.rollout(50, 1, 0)
    e.sync()
    e.lap_times()
    laps, joined = [], []
    for r in range(8):
        e.reset_random(1, r)
        e.sync()
        e.lap_begin()
        e.rollout(T, 1, 0)                                   # leaves the chains forked
        e.lap_end()
        laps.extend(e.lap_times())
        e.sync()
        e.timer_start()                                      # timer_start / timer_stop join
        e.rollout(T, 1, T)
        joined.append(e.timer_stop())
    lap, ref = np.median(laps), np.median(joined)
    assert 0.8 * ref <= lap <= 1.2 * ref, (lap, ref)
    assert lap * 1e3 / T > 25.0                              # us per step of 1 M arms: not half a step


def test_snapshot_gather_taken_per_chain_while_forked(m, monkeypatch):
    """mt_gather_returns_begin on a handle whose chains are forked snapshots every env range on its own chain (no join) and
    the chains stay forked for the per-chain reset queued next: the gathered returns must still be those of the finished
    episode for EVERY range, equal to what a single-chain engine gathers."""
    n, k = 400003, 2
    out = {}
    for chains in ("1", "2", "3"):
        monkeypatch.setenv("MT_CHAINS", chains)
        e = m.StepEngine(n, k)
        e.reset_random(3, 0)
        e.rollout(9, 3, 0)                                   # forked
        a = e.gather_begin()                                 # per-chain snapshot
        e.reset_random(3, 1)                                 # per chain, beside the exchange: must not leak zeros into `a`
        e.rollout(6, 3, 0)
        e.gather_wait()                                      # orders the handle's stream behind the first exchange
        b = e.gather_begin()                                 # second exchange: the snapshot buffer is reused behind the first
        e.reset_random(3, 2)
        e.gather_wait(host=True)
        e.sync()
        out[chains] = (a.cpu().numpy().copy(), b.cpu().numpy().copy(), e.total_reward())
    for chains in ("2", "3"):
        for x, y in zip(out[chains], out["1"]):
            np.testing.assert_array_equal(x, y)
    assert np.abs(out["1"][0]).max() > 0 and np.abs(out["1"][1]).max() > 0 and not out["1"][2].any()
