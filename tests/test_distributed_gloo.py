"""world_size-2 gloo rehearsal (CPU) of the N>1 path: shard ranges, shipping the RCCL unique id through the process
group, the gather's result layout, and bench.py's episode loop under the driver's own `--steps 20 --warmup 5`."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from manytor_amd import distributed as D
from oracle import philox_ref as px

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_cover_all_envs():
    for n, w in ((1048576, 8), (4194304, 8), (10, 3), (7, 8), (65536, 1)):
        spans = [D.shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == n
        for (b0, c0), (b1, _) in zip(spans, spans[1:]):
            assert b0 + c0 == b1
        assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    assert D.shard_range(4194304, 3, 8) == (3 * 524288, 524288)      # BASELINE.json configs[3]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    return D.init_process_group("gloo")


def _worker(rank, world, port, n_total, q):
    r, _, w = _init(rank, world, port)
    base, cnt = D.shard_range(n_total, r, w)
    ids = np.arange(base, base + cnt, dtype=np.uint64)
    # each rank derives its shard's inputs from GLOBAL env ids (what the device RNG does)
    local_actions = px.sample_actions(0x5EED, ids, 7, 4)
    local_returns = torch.from_numpy(local_actions.sum(axis=1).astype(np.float32))
    full = D.gloo_gather_returns(local_returns, n_total)
    # the control plane of D.connect: rank 0 makes the 128-byte id, everybody ends up with the same bytes
    uid = D.exchange_unique_id(lambda: bytes(range(128)), r)
    if r == 0:
        q.put((full.numpy(), uid))
    else:
        assert uid == bytes(range(128))
    dist.barrier()
    dist.destroy_process_group()


def _spawn(target, args, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port) + args + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=180)
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    return out


@pytest.mark.parametrize("n_total", [4096, 1001])
def test_gather_layout_world2_gloo(n_total):
    full, uid = _spawn(_worker, (n_total,))
    expect = px.sample_actions(0x5EED, np.arange(n_total, dtype=np.uint64), 7, 4).sum(axis=1).astype(np.float32)
    np.testing.assert_array_equal(full, expect)            # shard-invariant: same as one rank owning everything
    assert uid == bytes(range(128))


def _failing_id_worker(rank, world, port, q):
    r, _, w = _init(rank, world, port)

    def make_id():
        raise OSError("librccl.so: cannot open shared object file")
    try:
        D.exchange_unique_id(make_id, r)
        msg = None
    except RuntimeError as e:
        msg = str(e)
    # every rank left the SAME broadcast, so the next collective lines up (rank 0 used to raise before the
    # broadcast and its peers stayed blocked in it)
    flag = torch.tensor([0 if msg is None else 1], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.SUM)
    if r == 0:
        q.put((msg, int(flag.item())))
    else:
        assert msg is not None and "cannot open shared object" in msg
    dist.barrier()
    dist.destroy_process_group()


def test_unique_id_failure_on_rank0_raises_on_every_rank():
    msg, raised = _spawn(_failing_id_worker, ())
    assert raised == 2 and "rank 0 could not create the RCCL unique id" in msg and "cannot open shared object" in msg


def test_gather_single_process_is_identity():
    t = torch.arange(10, dtype=torch.float32)
    assert torch.equal(D.gloo_gather_returns(t, 10), t)


def _barrier_worker(rank, world, port, rounds, q):
    import time
    r, _, w = _init(rank, world, port)
    bar = D.make_host_barrier(r, w)
    assert bar is not None and not os.path.exists(bar.path)       # the name is gone once everybody has it mapped
    slots = bar._slots
    ok = True
    for it in range(1, rounds + 1):
        if (it + r) % 97 == 0:
            time.sleep(0.002)                                      # a straggler, a different rank every time
        slots[r, 1] = it                                           # "my work of round `it` is done"
        bar.wait()
        ok &= bool((slots[:, 1] >= it).all())                      # nobody left the barrier before everybody arrived
    t0 = time.perf_counter()
    for _ in range(200):
        bar.wait()
    us = (time.perf_counter() - t0) / 200 * 1e6
    dist.barrier()
    if r == 0:
        q.put((ok, us))
    else:
        assert ok
    bar.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_host_barrier_keeps_ranks_in_lockstep(world):
    ok, us = _spawn(_barrier_worker, (2000,), world=world)
    assert ok
    assert us < 5000                                               # microseconds per barrier (loose: shared CI cores)


def _barrier_unavailable_worker(rank, world, port, q):
    r, _, w = _init(rank, world, port)
    real = D.HostBarrier.__init__

    def broken(self, path, rank_, world_, create):
        if rank_ == 1:
            raise OSError("no shared /dev/shm on this rank")
        real(self, path, rank_, world_, create)
    D.HostBarrier.__init__ = broken
    bar = D.make_host_barrier(r, w)                                # one rank cannot attach => None on EVERY rank
    dist.barrier()
    if r == 0:
        q.put(bar is None)
    else:
        assert bar is None
    dist.destroy_process_group()


def test_host_barrier_falls_back_on_every_rank_if_one_cannot_attach():
    assert _spawn(_barrier_unavailable_worker, ()) is True
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("manytor_barrier_")]


class _FakeEngine:
    """Stand-in for StepEngine in bench.EpisodeLoop: returns = number of steps since the last reset + global env id /
    1e6, gathered over gloo.  Lets the CPU check count what the GPU run would launch."""

    tag_ids = True                                        # returns carry the global env id in their fraction

    def __init__(self, n_total, rank, world):
        self.n_total = n_total
        self.base, self.n = D.shard_range(n_total, rank, world)
        self.ret = torch.zeros(self.n)
        self.launches = self.gathers = self.resets = 0
        self.laps = []

    def reset_random(self, seed, episode):
        self.last = self.ret.clone()                      # the reset stores the finished return, then zeroes
        self.ret.zero_()
        self.resets += 1

    def rollout(self, steps, seed, step0):
        self.ret += steps
        self.launches += steps

    rollout_fused = rollout

    def gather_returns(self, out=None, field=None, snapshot=True):
        self.gathers += 1
        ids = torch.arange(self.base, self.base + self.n, dtype=torch.float32) if self.tag_ids else 0.0
        src = self.last if field is not None else self.ret       # MT_F_LAST_RETURN (what the reset kept) or the live returns
        return D.gloo_gather_returns(src + ids / 1e6, self.n_total)

    gather_begin = gather_returns

    def gather_wait(self, host=False):
        return 0.0 if host else None

    def lap_begin(self, what=None):
        self.laps.append(what)

    def lap_end(self, what=None):
        assert self.laps[-1] == what

    # the rest of what bench.measure / bench.TimedEngine ask of an engine
    def timer_start(self):
        self.t_start = self.launches

    def timer_stop_async(self):
        self.t_stop = self.launches

    def timer_read(self):
        return 0.01 * (self.t_stop - self.t_start)        # "device milliseconds" of the region: 10 us per step

    def lap_times(self):
        n, self.laps = len(self.laps), []
        return [0.05] * n                                 # "device milliseconds" per lap

    def sync(self):
        pass

    def total_reward(self):
        return self.ret.numpy()

    def total_envs(self):
        return self.n_total


def _bench_loop_worker(rank, world, port, steps, warmup, q):
    sys.path.insert(0, ROOT)
    import bench
    r, _, w = _init(rank, world, port)
    n_total = 1001
    eng = _FakeEngine(n_total, r, w)
    L = max(1, min(50, steps))                            # bench.py: >= 1 gather inside every timed region
    loop = bench.EpisodeLoop(eng, 0x5EED, L)
    loop.run(max(L, 200))                                 # pre-warm chunk
    loop.run(warmup)
    regions = []
    for _ in range(3):
        launches, gathers = loop.run(steps, time_kernels=True)
        regions.append((launches, gathers))
        dist.barrier()
    g = loop.gathered
    if r == 0:
        q.put((regions, g.numpy(), eng.launches, eng.gathers, eng.resets, L))
    dist.barrier()
    dist.destroy_process_group()


def test_episode_first_launch_stays_outside_the_step_laps():
    """Where mt_rollout absorbs the episode's reset into its first launch, bench.EpisodeLoop issues that launch (reset
    prologue + k steps) as a rollout call of its own outside the step laps: the region still runs EXACTLY `steps` steps, the
    laps hold the other steps only, and the calls are cut where one rollout call would cut its launches."""
    sys.path.insert(0, ROOT)
    import bench

    class Rec(_FakeEngine):
        def __init__(self):
            self.n_total, self.base, self.n = 64, 0, 64
            self.ret = torch.zeros(64)
            self.launches = self.gathers = self.resets = 0
            self.laps, self.calls = [], []

        def rollout(self, steps, seed, step0):
            self.calls.append((steps, step0, self.open))
            super().rollout(steps, seed, step0)

        def gather_returns(self, out=None, field=None, snapshot=True):
            self.gathers += 1
            return self.ret.clone()

        gather_begin = gather_returns

        def lap_begin(self, what=None):
            self.laps.append(what)
            self.open = True

        def lap_end(self, what=None):
            self.open = False

    for phase, want_calls in ((0, [(5, 0, False), (15, 5, True)]),
                              (10, [(10, 10, True), (5, 0, False), (5, 5, True)])):
        eng = Rec()
        eng.open = False
        loop = bench.EpisodeLoop(bench.TimedEngine(eng), 1, 20, steps_per_launch=5, absorbs_reset=True)
        loop.phase = phase
        loop.align()
        eng.calls.clear()
        before = eng.launches
        loop.eng.start_region()
        lapped, gathers = loop.run(20, time_kernels=True)
        assert eng.launches - before == 20 and gathers == 1
        assert [(s, s0 % 20, in_lap) for s, s0, in_lap in eng.calls] == want_calls
        assert lapped == 15 and loop.head_launches == 1 and loop.kernel_launches == 3
    # a region no longer than the first launch: lapped whole (nothing else to time)
    eng = Rec()
    eng.open = False
    loop = bench.EpisodeLoop(bench.TimedEngine(eng), 1, 5, steps_per_launch=5, absorbs_reset=True)
    loop.eng.start_region()
    assert loop.run(5, time_kernels=True) == (5, 1) and loop.head_launches == 0
    # an engine that does not absorb: one lapped call per segment, as ever
    eng = Rec()
    eng.open = False
    loop = bench.EpisodeLoop(bench.TimedEngine(eng), 1, 20, steps_per_launch=5)
    loop.eng.start_region()
    assert loop.run(20, time_kernels=True) == (20, 1) and [c[0] for c in eng.calls] == [20]


@pytest.mark.parametrize("steps,warmup", [(20, 5), (1000, 50), (7, 0)])
def test_bench_episode_loop_world2_gloo(steps, warmup):
    """The driver's invocation `--steps 20 --warmup 5` (and the default, and a tiny one): every timed region launches
    exactly `steps` steps and contains at least one gather; the gathered vector holds both ranks' shards."""
    regions, g, launches, gathers, resets, L = _spawn(_bench_loop_worker, (steps, warmup))
    for ln, gt in regions:
        assert ln == steps and gt >= 1
        assert gt in (steps // L, steps // L + 1)
    assert g.shape == (1001,)
    ids = np.arange(1001, dtype=np.float32) / np.float32(1e6)
    np.testing.assert_allclose(g - ids, np.full(1001, L), atol=1e-3)      # a full episode's return from every env
    assert launches == max(L, 200) + warmup + 3 * steps and resets == gathers + 1


def bench_lap_every():
    sys.path.insert(0, ROOT)
    import bench
    return bench.LAP_EVERY


def _measure_worker(rank, world, port, steps, warmup, phase, repeats, q):
    sys.path.insert(0, ROOT)
    import bench
    bench.EpisodeLoop.phase = phase                       # what main() sets from --episode-phase (N > 1: half an episode)
    r, _, w = _init(rank, world, port)
    bar = D.make_host_barrier(r, w)
    fab = bench.Fabric(r, w, dist, torch, "cpu", bar, None)
    out = {}
    for key, n_total in (("weak", 2 * 600), ("strong_1m", 1001), ("config3", 4003)):     # ragged strong shards
        eng = _FakeEngine(n_total, r, w)
        eng.tag_ids = False                               # bench.measure checks that returns are whole numbers
        res = bench.measure(fab, eng, n_total, steps, warmup, 50, 0x5EED, prewarm_s=0.02, min_timed_s=0.001, repeats=repeats)
        out[key] = (res, eng.launches, eng.gathers)
    if r == 0:
        q.put(out)
    dist.barrier()
    bar.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("phase,repeats", [(0, 0), (10, 0), (10, 17)])
def test_bench_measure_protocol_world2_gloo(phase, repeats):
    """bench.measure + bench.Fabric -- the protocol every leg of an N > 1 invocation goes through (headline, the 1 M-arm
    strong leg, configs[3]) -- at world size 2 over gloo with the shared-memory barrier: every region launches exactly
    `steps` steps and contains a gather, the figures every leg reports are there, and both ranks agree on the number of
    regions (rank 0's clock decides the time-based loops).  phase = 10 is the N > 1 default: a region starts half-way
    through an episode, so its one gather sits in the middle of it."""
    out = _spawn(_measure_worker, (20, 5, phase, repeats))
    for key, (res, launches, gathers) in out.items():
        for k in ("elapsed", "ms_per_step", "ms_per_step_min", "ms_per_step_max", "step_us", "launches", "gather_us",
                  "gathers_per_region", "repeats", "prewarm", "episode_len", "value", "value_device_timeline",
                  "region_device_ms", "device_ms_per_step", "steps_per_kernel_launch", "lapped_regions",
                  "laps_every_nth_region", "ms_per_step_lapped_regions"):
            assert k in res, (key, k)
        # the device timeline of a region: the stand-in reports 10 us per step between the start mark (after the opening
        # fence) and the end mark (before the closing one) -> exactly the region's 20 steps on both ranks
        assert res["region_device_ms"] == pytest.approx(0.2) and res["device_ms_per_step"] == pytest.approx(0.01)
        n_total = {"weak": 1200, "strong_1m": 1001, "config3": 4003}[key]
        assert res["value_device_timeline"] == pytest.approx(n_total * 20 / 0.2e-3)
        assert res["steps_per_kernel_launch"] == 1
        assert res["episode_len"] == 20 and res["gathers_per_region"] == 1 and res["repeats"] >= 5
        # the step laps ride in every region of a short run, in every 4th one of a long one (the stopwatch is not free);
        # every region runs its 20 steps either way
        if repeats:
            assert res["repeats"] == repeats and res["laps_every_nth_region"] == bench_lap_every() and res["lapped_regions"] == 5
        else:
            assert res["laps_every_nth_region"] == 1 and res["lapped_regions"] == res["repeats"]
        assert res["launches"] == 20 * res["lapped_regions"]
        assert launches == res["prewarm"] + 5 + phase + 20 * res["repeats"]
        assert res["ms_per_step_min"] <= res["ms_per_step"] <= res["ms_per_step_max"]
        # aligned after the warm-up, a 20-step region is ONE step segment = one 0.05 ms lap; started half-way through an
        # episode it is two segments around the episode end = two laps
        assert res["step_us"] == pytest.approx((50.0 if phase == 0 else 100.0) / 20)


# ---- bench.py starts its own ranks -------------------------------------------------------------------
def test_bench_launch_dry_run_prints_the_children():
    """`python bench.py --gpus 2 --steps 20 --warmup 5` with no launcher around it: the parent only plans / starts the
    ranks.  --launch-dry-run prints the children's argv and rank environment; nothing imports torch or touches a GPU."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
                          "--launch-dry-run"], capture_output=True, text=True, timeout=60, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    plan = json.loads(out.stdout)
    kids = plan["children"]
    assert len(kids) == 2
    ports = set()
    for r, kid in enumerate(kids):
        assert kid["argv"][0] == sys.executable and kid["argv"][1] == os.path.join(ROOT, "bench.py")
        assert kid["argv"][2:] == ["--gpus", "2", "--steps", "20", "--warmup", "5"]          # the dry-run flag does not travel
        e = kid["env"]
        assert e["RANK"] == str(r) and e["LOCAL_RANK"] == str(r) and e["WORLD_SIZE"] == "2"
        assert e["MASTER_ADDR"] == "127.0.0.1" and e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        ports.add(e["MASTER_PORT"])
    assert len(ports) == 1 and 1024 < int(ports.pop()) < 65536
    # under a launcher (WORLD_SIZE set) the script is a rank, never a launcher
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-dry-run"],
                         capture_output=True, text=True, timeout=60, env={**env, "WORLD_SIZE": "2", "RANK": "0"})
    assert out.returncode == 0 and json.loads(out.stdout)["children"] == []


_STUB = r"""
import json, os, sys, time
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
mode = sys.argv[sys.argv.index("--mode") + 1]
assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0 and os.environ["LOCAL_RANK"] == str(rank)
print(f"chatter of rank {rank}")                       # native libraries write to stdout too
if mode == "ok":
    if rank == 0:
        print(json.dumps({"metric": "m", "value": 1.0, "n_gpus": world}))
    sys.exit(0)
if mode == "rank1_fails":
    if rank == 1:
        sys.exit(3)
    time.sleep(60)                                     # the launcher has to end this one
if mode == "hang":
    time.sleep(60)
"""


@pytest.mark.parametrize("mode,rc", [("ok", 0), ("rank1_fails", 3), ("hang", 124)])
def test_bench_self_launch_relays_rank0_and_ends_the_group(mode, rc, tmp_path, capfd):
    """bench.self_launch with a stand-in rank script: rank 0's JSON line -- and only it -- reaches stdout, a failing rank or
    the timeout ends every other rank (each child leads its own process group) and sets the exit code."""
    import time
    sys.path.insert(0, ROOT)
    import bench
    stub = tmp_path / "rank_stub.py"
    stub.write_text(_STUB)
    t0 = time.monotonic()
    got = bench.self_launch(["--gpus", "3", "--mode", mode], 3, timeout_s=3.0 if mode == "hang" else 60.0, script=str(stub))
    took = time.monotonic() - t0
    out, err = capfd.readouterr()
    assert got == rc
    assert took < 30.0                                       # nobody waited for the sleeping ranks
    if mode == "ok":
        assert out.count("\n") == 1 and '"metric": "m"' in out and '"n_gpus": 3' in out
        for r in range(3):
            assert f"chatter of rank {r}" in err              # everything else went to stderr
    else:
        assert out == ""
        assert ("rank 1 exited with 3" in err) if mode == "rank1_fails" else ("timeout" in err)


def _ring_gather_worker(rank, world, port, q):
    """the gloo stand-in gathering a 2-D field (a return-ring row) used numpy without importing it."""
    sys.path.insert(0, ROOT)
    r, _, w = _init(rank, world, port)
    n_total = 11
    base, n = D.shard_range(n_total, r, w)

    class Eng:                                            # what attach_gloo_gather needs of a StepEngine
        device = "cpu"

        def get(self, field):
            ids = np.arange(base, base + n, dtype=np.float32)
            return ids if field != 13 else np.stack([ids, 100 + ids, 200 + ids], axis=1)   # MT_F_RETURN_RING: (n, R)

    eng = D.attach_gloo_gather(Eng(), n_total, r, w)
    out = torch.empty(n_total)
    eng.gather_returns(out, field=13, row=2)
    flat = torch.empty(n_total)
    eng.gather_returns(flat)
    if r == 0:
        q.put((out.numpy().copy(), flat.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_stand_in_gathers_a_return_ring_row():
    ring, flat = _spawn(_ring_gather_worker, ())
    np.testing.assert_array_equal(ring, 200 + np.arange(11, dtype=np.float32))
    np.testing.assert_array_equal(flat, np.arange(11, dtype=np.float32))


def test_bench_byte_models_and_traffic_lookup():
    """The byte accounting bench.py's roofline uses: SURVEY 8(d)'s model, what one launch per step really moves, and what a
    step moves when k steps share a launch (outputs every step, the state rows once per launch); the PMC traffic entry of a
    configuration only answers for a run that launches the same way."""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.algorithmic_bytes_per_env_step(4, 7) == 249 and bench.algorithmic_bytes_per_env_step(7, 7) == 285
    assert bench.actual_bytes_per_env_step(4, 7) == 233 and bench.actual_bytes_per_env_step(7, 7) == 257
    assert bench.moved_bytes_per_env_step(4, 7, 1) == 233 and bench.moved_bytes_per_env_step(7, 7, 1) == 257
    assert bench.moved_bytes_per_env_step(4, 7, 5) == pytest.approx(101 + 132 / 5)
    assert bench.moved_bytes_per_env_step(4, 7, 50) == pytest.approx(101 + 132 / 50)
    one = bench.load_traffic("d4_k7_n1048576")
    assert one is not None and 2.3e8 < one < 2.6e8
    assert bench.load_traffic("d4_k7_n1048576", 5) is None                  # measured with one launch per step
    five = bench.load_traffic("d4_k7_n131072", 5.0)
    assert five is not None and 0.95 < five / (bench.moved_bytes_per_env_step(4, 7, 5) * 5 * 131072) < 1.1
    assert bench.load_traffic("d4_k7_n131072") is None and bench.load_traffic("no_such_config") is None


def test_bench_lap_statistics_drop_a_booked_host_stall():
    """A lap is [begin event .. latest end event]; a host stall between a lap's last launch and the record of its end event is
    booked as device time (80 ms once per process was observed: profiles/r04_placement_history.txt).  The secondary figures of
    bench.py are step-weighted means over the laps WITHOUT the laps beyond three times the median lap."""
    sys.path.insert(0, ROOT)
    import bench
    laps = [80.799, 0.916, 0.954, 1.002, 0.994, 0.985, 0.972, 0.955, 0.949, 0.949, 0.929, 0.922]      # ms, 50 steps each: the observed case
    assert bench.robust_us_per_step(laps, [50] * 12) == pytest.approx(sum(laps[1:]) * 1e3 / 550)
    assert sum(laps) * 1e3 / 600 > 150                                                               # what the plain sum read
    assert bench.robust_us_per_step([1.0, 1.0, 1.0], [50, 50, 25]) == pytest.approx(3000 / 125)       # ragged laps: weighted
    assert bench.robust_us_per_step([1.0, 2.5, 1.2], [50, 50, 50]) == pytest.approx(4700 / 150)       # 2.5 x the median stays
    assert bench.robust_us_per_step([], []) == 0.0
