#!/usr/bin/env python3
"""Benchmark of the hot path: env-steps/sec of the batched manipulator step on MI355X.

    python bench.py                                   # 1 GPU, 1 048 576 arms, 4-DoF, K=7
    python bench.py --gpus N --steps K --warmup W     # N GPUs of one node: the script starts its own N ranks
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W          # ... or runs as the ranks torchrun started

One "step" = one Environment.step() (manytor.py:255-260: 25 interpolated sub-steps of DH forward
kinematics, ground flag, observation, pickup, reward, return) for EVERY env of the batch, with the
random action drawn in the same launch (manytor.py:215-217).  Inputs are synthetic and already
resident in HBM when the timed region starts.  Every `episode_len` steps the returns are gathered
over the ranks (mt_gather_returns: RCCL all-gather straight from the arena, the only collective) and
all envs are reset (test_multi.py:32-34).

Launching (`self_launch`): with --gpus N > 1 and no WORLD_SIZE in the environment the process is only a launcher -- before
anything touches torch or HIP it starts N fresh children of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR =
127.0.0.1 / a free MASTER_PORT), relays rank 0's one JSON line to its own stdout and everything else to stderr, returns the
worst exit code, and ends the whole group on the first failure or on --launch-timeout.  Under torchrun (WORLD_SIZE set)
nothing of that runs.

What N > 1 measures: the metric is "env-steps/sec (whole node), 1M parallel 4-DoF arms", so the headline of an N > 1 line is
that workload -- 1 048 576 arms SHARDED over the N GPUs ("scaling": "strong") -- and `secondary` holds the weak-scaling leg
(1 048 576 arms per GPU) and BASELINE.json configs[3] (4 194 304 arms sharded), each per-step and fused.  At N = 1 strong and
weak are the same job.  --envs-per-gpu / --envs-total pick another headline.

Timing protocol (robust to short --steps; `measure()`): a time-based pre-warm brings the GPU to its steady clock, then W
untimed warm-up steps, the running episode is ended so that a region starts an episode (N = 1) or half-way through one (N > 1:
the region's RCCL exchange then runs beside the steps behind the episode end; --episode-phase), then the region of EXACTLY K
steps -- bracketed by barrier + torch.cuda.synchronize() on both sides, max over ranks -- is run `repeats` times; `ms_per_step` /
`value` come from the MEDIAN region (min / max beside it).  Beside the wall clock every region is bracketed by HIP events on
the engine's streams (mt_timer_start / mt_timer_stop_async: from the idle device at the region's start to the end of its last
kernel, reset or exchange, no host wait in between): `value_device_timeline` is the same throughput on that clock, i.e. without
the host latency of the two fences, which at 8 GPUs x 131 072 arms is a third of a 20-step region.  The device time of a STEP
is measured with HIP-event laps around the step launches only (mt_rollout may run a step as two concurrent launches on two
streams that stay forked across calls; a lap begins with the first of them and ends when the last of them does; `roofline`
says so).  Where mt_rollout takes the episode's reset into its first launch (small shards), that launch -- another kernel --
is issued as a rollout call of its own outside the laps, inside the region and its device timeline (EpisodeLoop.head_steps).
The laps are a stopwatch inside the timed work (two event records per lap, 7 us): every LAP_EVERY-th region carries them, the
others run the same calls without; `value` is the median over all regions, `config.ms_per_step_lapped_regions` beside it.

Prints ONE JSON line on rank 0 (contract: see the task brief / DESIGN.md section "Measurement").
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "env-steps/sec (whole node), 1M parallel 4-DoF arms, random actions"
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
MIN_TIMED_S = 0.25           # the repeated timed regions together cover at least this much GPU work (5..300 regions)
LAP_EVERY = 4                   # bench.measure: every 4th timed region carries the HIP-event step laps (see there)
PREWARM_S = 0.3
F_LAST_RETURN = 12           # mt_field MT_F_LAST_RETURN (include/manytor_hip.h): the return an env had when it was last reset


def algorithmic_bytes_per_env_step(dof, k):
    """reads action 4D + goals 4D + points 12K + alive 4 + return 4; writes goals 4D +
    obs 12K + reward 4 + done 1 + alive 4 + return 4 + end effector 12."""
    return 12 * dof + 24 * k + 33


def actual_bytes_per_env_step(dof, k):
    """What mt_step_random really moves: the action is drawn in-kernel, so the 4D-byte action read of the
    SURVEY model does not happen (PMC traffic agrees: profiles/traffic.json)."""
    return algorithmic_bytes_per_env_step(dof, k) - 4 * dof


def moved_bytes_per_env_step(dof, k, steps_per_launch=1.0):
    """Bytes one env-step really moves when `steps_per_launch` consecutive steps share a launch (mt_rollout on small shards,
    mt_rollout_fused): every step writes its outputs (obs 12K + reward 4 + done 1 + end effector 12); the state -- goals 4D,
    targets 12K, alive mask 4, return 4 read; goals, alive mask, return written -- crosses once per LAUNCH.  One step per
    launch gives actual_bytes_per_env_step()."""
    return (12 * k + 17) + (8 * dof + 12 * k + 16) / float(steps_per_launch)


def usable_cores():
    """Host cores this process may really use: the affinity mask, capped by the cgroup CPU quota (the GPU box gives a
    one-GPU job a 16-core share of its 256 cores; oversubscribing a quota only adds throttling noise)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    why = "sched_getaffinity"
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            q = max(1, int(math.ceil(int(quota) / int(period))))
            if q < cores:
                cores, why = q, "cgroup cpu.max quota"
    except (OSError, ValueError):
        pass
    return cores, why


def numpy_multiprocess_baseline(dof, k, workers, budget_s=3.0, shard=16384, timeout_s=120.0):
    """vectorised numpy fanned out over `workers` processes (oracle/numpy_shard_worker.py, one per usable
    core, plain subprocesses).  All workers finish their start-up, are released together, and step for ~budget_s; the
    figure is total env-steps / the slowest worker's stepping time.  Returns (value or None, description)."""
    import subprocess
    procs = []
    try:
        for w in range(workers):
            procs.append(subprocess.Popen([sys.executable, "-m", "oracle.numpy_shard_worker", str(w), str(shard), str(dof), str(k),
                                           str(budget_s)], cwd=ROOT, stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                          stderr=subprocess.DEVNULL, text=True))
        deadline = time.monotonic() + timeout_s
        for p_ in procs:
            line = p_.stdout.readline()
            if line.strip() != "ready" or time.monotonic() > deadline:
                raise RuntimeError("a worker did not come up")
        for p_ in procs:
            p_.stdin.write("go\n")
            p_.stdin.flush()
        res = []
        for p_ in procs:
            out, _ = p_.communicate(timeout=max(1.0, deadline - time.monotonic()))
            res.append(json.loads(out.strip().splitlines()[-1]))
        slowest = max(r["seconds"] for r in res)
        total = sum(r["env_steps"] for r in res)
        return total / slowest, (f"{workers} processes x {shard} envs x {res[0]['env_steps'] // shard} steps (worker 0), oracle "
                                 f"BatchOracle fp64 (numpy), slowest worker {slowest:.1f} s")
    except Exception as e:                                   # noqa: BLE001 -- a baseline leg must not take the bench down
        return None, f"failed: {type(e).__name__}: {e}"
    finally:
        for p_ in procs:
            if p_.poll() is None:
                p_.kill()


def cpu_baseline(dof_table, k, budget_s=4.0, n=65536):
    """The CPU port of the step path (oracle/: parity-checked against the reference's fixtures), timed on this box's
    host cores on a bounded sample of the same workload (about 10 s in all).  Headline figure: the C restatement with
    OpenMP on every core this process may use; beside it the reference's own shape of CPU parallelism -- vectorised numpy,
    env shards fanned out over all usable cores with multiprocessing -- and the one-core vectorised-numpy
    and scalar (reference-shaped) Python figures.  Baseline, not the target."""
    from oracle import c_oracle
    from oracle import manytor_oracle as mo
    from oracle import philox_ref as px
    table = np.asarray(dof_table)
    dof = table.shape[0]
    threads, why = usable_cores()
    # (a) C + OpenMP
    nc = 1 << 20
    idc = np.arange(nc, dtype=np.uint64)
    cora = c_oracle.COracle(nc, k, table=table, threads=threads)
    radius = 51.3 if dof == 4 else 92.6
    # targets: the device stream's law (hemisphere rejection sampling) for 65 536 envs, tiled (the numpy restatement of
    # the rejection loop is slow at 1 M envs and the step's cost does not depend on which targets it sees)
    pts = np.tile(px.sample_targets(0x5EED, idc[:n], 0, k, radius).astype(np.float64), (nc // n, 1, 1))
    cora.reset(pts)
    cacts = [px.sample_actions(0x5EED, idc, t, dof).astype(np.float64) for t in range(4)]
    cora.step(cacts[0])
    t0 = time.perf_counter()
    csteps = 0
    while csteps < 1 or (time.perf_counter() - t0 < budget_s and csteps < 400):
        cora.step(cacts[csteps % 4])
        csteps += 1
    dt_c = time.perf_counter() - t0
    del cora, cacts
    # (b) vectorised numpy, one core
    ids = np.arange(n, dtype=np.uint64)
    ora = mo.BatchOracle(n, k, table=table)
    ora.reset(pts[:n])
    acts = [px.sample_actions(0x5EED, ids, t, dof).astype(np.float64) for t in range(8)]
    ora.step(acts[0])
    t0 = time.perf_counter()
    steps = 0
    while steps < 2 or (time.perf_counter() - t0 < budget_s / 2 and steps < 7):
        ora.step(acts[1 + steps])
        steps += 1
    dt = time.perf_counter() - t0
    # (b') the same, env shards over all usable cores, one process each
    mp_value, mp_sample = numpy_multiprocess_baseline(dof, k, threads, budget_s=budget_s * 0.75)
    # (c) the reference's own shape of the computation: one env at a time, three FK chains per sub-step (manytor.py:188)
    envs = [mo.ScalarEnv(k, table=table) for _ in range(16)]
    for e, p_ in zip(envs, pts[:16]):
        e.reset(points=p_)
    t1 = time.perf_counter()
    for t in range(6):
        for j, e in enumerate(envs):
            e.step(acts[t][j])
    dt_scalar = time.perf_counter() - t1
    return {
        "value": nc * csteps / dt_c, "unit": "env-steps/s", "cores": threads, "kind": "port",
        "sample": f"{nc} envs x {csteps} steps, C restatement of the reference with OpenMP (oracle/manytor_oracle.c), "
                  f"{dt_c:.1f} s on {threads} threads = every core this process may use ({why}; os.cpu_count() = "
                  f"{os.cpu_count()})",
        "numpy_multiprocess_value": mp_value, "numpy_multiprocess_cores": threads, "numpy_multiprocess_sample": mp_sample,
        "numpy_vectorised_value_1core": n * steps / dt,
        "numpy_vectorised_sample": f"{n} envs x {steps} steps, oracle BatchOracle fp64, {dt:.1f} s",
        "scalar_faithful_value_1core": 16 * 6 / dt_scalar,
        "scalar_faithful_sample": "16 envs x 6 steps, per-env Python loop with the reference's 75 FK chains per step "
                                  "(oracle ScalarEnv)",
    }


def achievable_bandwidth(m, dof, k, n_envs, dev, reps=0, attempts=4):
    """GB/s of the step's own access shape with no arithmetic (mt_stream_probe: the same rows read, rewritten and
    nt-written per env, same addressing, one env per lane) over `n_envs` envs: the yardstick SURVEY.md 8(d) asks for next to
    the 8 TB/s spec figure.  The BEST of `attempts` probes, each on freshly allocated buffers: a large arena streams 6 %
    faster or slower by where its pages landed, from one allocation to the next, and "achievable" is
    the faster mode -- the step's own 4 M-arm figure is the faster of its two passes as well.  None where the probe is not
    built for (dof, k)."""
    best = None
    for _ in range(attempts):
        try:
            us, nbytes = m.stream_probe(n_envs, dof, k, reps or max(20, min(400, int(2e8 // n_envs))), dev)
        except (m.ManytorError, ValueError):
            return None
        gbs = nbytes / (us * 1e-6) / 1e9
        best = gbs if best is None else max(best, gbs)
    return best


def load_traffic(workload_key, steps_per_launch=1.0):
    """HBM bytes per step launch from the committed PMC run (profiles/traffic.json), or None.  PMC counters need
    their own rocprofv3 passes, so this figure cannot be taken inside a bench run; `traffic_source` says so.  An entry
    measured with k steps per launch only answers for a run that launches the same way."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            entry = json.load(f).get(workload_key, {})
    except (OSError, ValueError):
        return None
    if round(float(entry.get("steps_per_launch", 1))) != round(float(steps_per_launch)):
        return None
    return entry.get("hbm_bytes_per_launch")


class EpisodeLoop:
    """The benchmark's control flow, the shape of test_multi.py:17-34: `episode_len` random-action steps, then gather
    the returns of all ranks and reset every env.  `engine` needs rollout / rollout_fused / reset_random /
    gather_begin / gather_wait / gather_returns / lap_begin / lap_end (manytor_amd.StepEngine; a stand-in in the CPU
    rehearsal test).  With `overlap` (default) the gather runs on the engine's side stream while the next episode's steps
    run beside it; its result is the previous episode's, as a learner would consume it, in one of two alternating
    buffers."""

    reset_first = False         # experiment switch (--reset-before-gather), see _episode_end

    def __init__(self, engine, seed, episode_len, fused=False, overlap=True, steps_per_launch=1, absorbs_reset=False):
        self.eng, self.seed, self.L, self.fused, self.overlap = engine, seed, int(episode_len), fused, overlap
        self.steps_per_launch = max(1, int(steps_per_launch))     # mt_rollout's k on small shards (engine.dispatch())
        # mt_rollout absorbs a deferred reset into its first launch (dispatch().rollout.absorbs_reset): the timed loop then
        # issues an episode's first launch -- reset prologue + `steps_per_launch` steps -- as a call of its own OUTSIDE the step
        # laps (inside the region and its device timeline like everything else), so that the laps hold step launches only
        # and agree with the kernel trace's average of the step kernel.  The launches are the ones one rollout call makes.
        self.head_steps = self.steps_per_launch if (absorbs_reset and not fused) else 0
        self.head_launches = 0      # episode-first launches issued outside the step laps (see head_steps)
        self.kernel_launches = 0    # kernel launches (per env range) behind the timed steps: ceil(segment / k), 1 per fused segment
        self.lap_steps = []         # steps inside every timed step lap, in lap order
        self.step = 0
        self.episode = 0
        self.gathers = 0
        self.gathered = None
        self._bufs = [None, None]
        self.eng.reset_random(seed, 0)

    phase = 0                   # steps of the running episode already done when a timed region starts (--episode-phase)

    def align(self):
        """End the running episode now (gather + reset, as at a regular episode end), so that the next step is the first of
        an episode, then run `phase` steps of it.  bench.measure() calls it between the untimed warm-up steps and the first
        timed region.  phase = 0: a region of K = episode_len steps is ONE rollout segment followed by its gather and reset
        (the exchange then runs beside the reset only, and the fence waits for it).  phase = p: every region is the last
        L - p steps of an episode, its gather and reset, and the first p steps of the next one beside the exchange."""
        if self.step % self.L:
            self.step += self.L - self.step % self.L
            self._episode_end(False)
        if self.phase % self.L:
            self.run(self.phase % self.L)

    def _episode_end(self, time_kernels):
        """test_multi.py:32-34: read every env's return, then reset every env.  Overlapped form: the returns are
        snapshotted (per chain while mt_rollout's chains are forked, each range behind its own last step), the exchange
        runs on the engine's side stream, and the reset -- per chain as well -- and the next episode run beside it.
        `reset_first` is the measured alternative: reset first (it stores each env's finished return in
        MT_F_LAST_RETURN), then gather that row in place, no snapshot copy (tools/ab_episode_end.sh: a tie)."""
        self.episode += 1
        if self.reset_first:
            self.eng.reset_random(self.seed, self.episode)
            if self.overlap:
                b = self.gathers % 2
                self._bufs[b] = self.gathered = self.eng.gather_begin(self._bufs[b], field=F_LAST_RETURN, snapshot=False)
            else:
                self.gathered = self.eng.gather_returns(self.gathered, field=F_LAST_RETURN)
            self.gathers += 1
            return
        if self.overlap:
            self.eng.gather_wait()                   # stream order only: the previous exchange finished long ago
            b = self.gathers % 2
            self._bufs[b] = self.gathered = self.eng.gather_begin(self._bufs[b])
        else:
            if time_kernels:
                self.eng.lap_begin("gather")
            self.gathered = self.eng.gather_returns(self.gathered)     # RCCL all-gather (device copy at N = 1)
            if time_kernels:
                self.eng.lap_end("gather")
        self.gathers += 1
        self.eng.reset_random(self.seed, self.episode)

    def run(self, count, time_kernels=False, laps=True):
        """`count` env steps.  Returns (step launches timed, gathers done) of this call.  time_kernels: the calls are cut as a
        timed region cuts them (an episode's first launch by itself); laps = False: the same calls WITHOUT the HIP-event laps
        around them (bench.measure laps every LAP_EVERY-th region only: the stopwatch costs 7 us per lap, tools/lap_cost.py)."""
        done = launches = gathers = 0
        lap = time_kernels and laps
        while done < count:
            seg = min(count - done, self.L - self.step % self.L)
            if time_kernels and self.head_steps and self.step % self.L == 0 and seg > self.head_steps:
                seg = self.head_steps
                self.eng.rollout(seg, self.seed, self.step)          # the episode's first launch: reset + steps, unlapped
                self.step += seg
                done += seg
                self.head_launches += 1
                if self.step % self.L == 0:
                    self._episode_end(lap)
                    gathers += 1
                continue
            if lap:
                self.eng.lap_begin("step")      # HIP events on the engine's stream, no host synchronisation
            if self.fused:
                self.eng.rollout_fused(seg, self.seed, self.step)
            else:
                self.eng.rollout(seg, self.seed, self.step)
            if lap:
                self.eng.lap_end("step")
                launches += seg
                self.lap_steps.append(seg)
                self.kernel_launches += 1 if self.fused else -(-seg // self.steps_per_launch)
            self.step += seg
            done += seg
            if self.step % self.L == 0:
                self._episode_end(lap)
                gathers += 1
        return launches, gathers


class TimedEngine:
    """StepEngine plus two named HIP-event lap timers (step launches, gathers) for EpisodeLoop."""

    def __init__(self, eng):
        self.e = eng
        self.ms = {"step": 0.0, "gather": 0.0}
        self._open = None

    def __getattr__(self, name):
        return getattr(self.e, name)

    def lap_begin(self, what):
        self._open = what
        self.e.lap_begin()

    def lap_end(self, what):
        self.e.lap_end()
        # one synchronisation-free event pair per lap; totals are read per kind at region end (collect)
        self._kinds.append(what)

    _kinds = None

    def start_region(self):
        self._kinds = []

    def collect(self):
        """Per-kind device milliseconds of the laps since start_region (synchronises once): the sums, and under
        "step_laps" every step lap by itself.  A lap is [begin event .. latest end event]: an event recorded on an idle
        stream completes at once, so a HOST stall between a lap's last launch and its end-event record (a first-use
        allocation inside the HIP runtime: 80 ms observed, once per process, in whichever lap was the unlucky one) reads as
        device time of that lap.  Callers therefore take the MEDIAN over laps / regions, never the plain sum."""
        ms = self.e.lap_times()
        out = {"step": 0.0, "gather": 0.0, "step_laps": []}
        for kind, v in zip(self._kinds, ms):
            out[kind] += v
            if kind == "step":
                out["step_laps"].append(v)
        self._kinds = []
        return out


def robust_us_per_step(lap_ms, lap_steps):
    """us per step from laps of known step counts: the step-weighted mean over the laps (so that the mix of early-episode
    and late-episode steps stays what it is), without the laps whose per-step time exceeds three times the median lap's --
    a host stall booked as device time (TimedEngine.collect) reads 10-100 x, nothing the device does reads 3 x."""
    per = [ms * 1e3 / max(1, st) for ms, st in zip(lap_ms, lap_steps)]
    if not per:
        return 0.0
    med = sorted(per)[len(per) // 2]
    keep = [(ms, st) for ms, st, p_ in zip(lap_ms, lap_steps, per) if p_ <= 3.0 * med]
    return sum(ms for ms, _ in keep) * 1e3 / max(1, sum(st for _, st in keep))


def time_step_launches(m, n, table, radius, k, dev, seed, fused=False, steps=600, episode_len=50, want_spl=False, rollout_k=None):
    """us per step of a secondary configuration: the same episode loop as the headline (reset every `episode_len`
    steps, so the alive masks stay those of real episodes), pre-warmed, HIP events around the step launches only.
    rollout_k = 1 forces mt_rollout's one-launch-per-step form (MT_ROLLOUT_K, read by mt_create) where the default on this
    batch size is k steps per launch.  -> (us, kernel name[, steps per kernel launch])."""
    keep = os.environ.get("MT_ROLLOUT_K")
    if rollout_k is not None:
        os.environ["MT_ROLLOUT_K"] = str(rollout_k)
    try:
        raw = m.StepEngine(n, k, dh_table=table, radius=radius, device=dev)
    finally:
        if rollout_k is not None:
            if keep is None:
                os.environ.pop("MT_ROLLOUT_K", None)
            else:
                os.environ["MT_ROLLOUT_K"] = keep
    e = TimedEngine(raw)
    loop = EpisodeLoop(e, seed, episode_len, fused=fused, overlap=False, steps_per_launch=rollout_steps_per_launch(raw),
                       absorbs_reset=rollout_absorbs_reset(raw))
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:
        loop.run(4 * episode_len)
        e.sync()
    e.start_region()
    loop.lap_steps = []
    launches, _ = loop.run(steps, time_kernels=True)
    us = robust_us_per_step(e.collect()["step_laps"], loop.lap_steps)      # mean over the episode laps, stalled laps dropped
    name = "rollout_kernel (fused)" if fused else e.step_kernel_name()
    spl = launches / max(1, loop.kernel_launches)
    e.close()
    return (us, name, spl) if want_spl else (us, name)


def time_loaded_action_steps(m, n, table, radius, k, dev, seed, steps=600, episode_len=50, chunk=10):
    """us per mt_step when the actions come from HBM, as with a policy in the loop (step_kernel<SAMPLE = false>: exactly
    SURVEY 8(d)'s byte model incl. the 4D-byte action read).  The actions of every step are written by mt_sample_actions (a
    separate small launch), so the arms move as in the headline.  On a multi-chain handle both calls are issued per chain
    (each half of the env range on its own stream, left forked from call to call).  What is TIMED is the (sample, step)
    pair: HIP-event laps around chunks of `chunk` pairs (a lap around every single launch would add the cost of its two
    event records to a 40 us kernel).  The step's own time is the pair minus the sampler's cheapest form -- ONE launch over
    the whole batch, timed alone on a caller's stream, where the per-step calls are single launches: a lower bound of what
    the sampler costs inside the pair, hence an upper bound of the step (subtracting the per-chain sampler timed alone read
    6.5-10 us at 1 M arms from run to run -- two tiny launches and their fork -- and flattered the step by as much).
    -> (us per step = pair - single-launch sample, us per sample launch, us per pair as timed, kernel name)."""
    import torch
    e = m.StepEngine(n, k, dh_table=table, radius=radius, device=dev)
    t = 0
    seg_steps = []

    def pairs(count, timed):
        nonlocal t
        done = 0
        while done < count:
            if t % episode_len == 0:
                e.reset_random(seed, t // episode_len)
            seg = min(chunk, count - done, episode_len - t % episode_len)
            if timed:
                e.lap_begin()
            for _ in range(seg):
                e.sample_actions(seed, t)
                e.step()
                t += 1
            if timed:
                e.lap_end()
                seg_steps.append(seg)
            done += seg
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:
        pairs(200, False)
        e.sync()
    e.lap_times()
    pairs(steps, True)
    us_pair = robust_us_per_step(e.lap_times(), seg_steps)                 # mean over the chunk laps, stalled laps dropped
    d = e.dispatch()["chains"]
    name = e.step_kernel_name().split(" [mt_rollout")[0].replace("step_kernel<", "step_kernel<SAMPLE=false, ") + \
        (f" [mt_step: {d['count']} chains of {d['span']} envs]" if d["count"] > 1 else "")
    torch.cuda.synchronize()
    e.use_torch_stream()                                                   # per-step calls are single launches here
    for rep in range(steps // chunk):
        e.lap_begin()
        for j in range(chunk):
            e.sample_actions(seed, rep * chunk + j)
        e.lap_end()
    us_sample = robust_us_per_step(e.lap_times(), [chunk] * (steps // chunk))
    e.close()
    return us_pair - us_sample, us_sample, us_pair, name


class Fabric:
    """What a measurement needs from the process group: rank / world, a fence (own device drained, every rank arrived,
    device idle), "rank 0 decides" and max-over-ranks.  One rank: all trivial."""

    def __init__(self, rank=0, world=1, dist=None, torch=None, ctrl_dev="cpu", host_barrier=None, device_sync=None):
        self.rank, self.world, self.dist, self.torch, self.ctrl_dev = rank, world, dist, torch, ctrl_dev
        self.host_barrier, self.device_sync = host_barrier, device_sync or (lambda: None)

    def fence(self, raw):
        """torch.cuda.synchronize() + barrier.  The device is drained FIRST -- every stream of this rank, incl. an exchange
        still running on the engine's side stream -- so that the barrier is passed only when every rank's device work is
        complete (and a collective barrier never shares the device with an exchange of the other communicator); behind the
        barrier nothing is outstanding anywhere, so no second synchronisation follows (it cost 13 us per fence doing
        nothing: profiles/r03_region_host_latency.txt -- a tenth of a 20-step region of a 131 072-arm shard)."""
        self.device_sync()
        if self.world > 1:
            if self.host_barrier is not None:
                self.host_barrier.wait()
            else:
                self.dist.barrier()

    def rank0_says(self, flag):
        """Time-based loops contain collectives (the gathers): every rank must run the same number of them, so rank 0's
        clock decides for everybody."""
        if self.world == 1:
            return flag
        t = self.torch.tensor([1 if flag else 0], dtype=self.torch.int32, device=self.ctrl_dev)
        self.dist.broadcast(t, src=0)
        return bool(int(t.item()))

    def max_over_ranks(self, values):
        if self.world == 1:
            return np.asarray(values, dtype=np.float64)
        t = self.torch.tensor(np.asarray(values, dtype=np.float64), dtype=self.torch.float64, device=self.ctrl_dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return t.cpu().numpy()

    def any_rank(self, flag):
        if self.world == 1:
            return bool(flag)
        t = self.torch.tensor([1 if flag else 0], dtype=self.torch.int32, device=self.ctrl_dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return bool(int(t.item()))


def fraction_or_none(x):
    """A fraction of the HBM peak on SURVEY 8(d)'s MODEL bytes (12 D + 24 K + 33 per env-step), or None where the launches
    move so much less than the model says -- no action read (the action is drawn in registers), the state rows once per k
    steps -- that model bytes over time would exceed the peak: not a fraction of anything then (the GB/s figure beside it
    stays; `frac` / `frac_of_hbm_peak` on the bytes really moved is the roofline figure)."""
    return x if x <= 1.0 else None


def rollout_absorbs_reset(raw):
    """Does mt_rollout on this engine take a deferred mt_reset_random into its first launch?  (mt_describe_dispatch)"""
    try:
        return bool(raw.dispatch()["rollout"]["absorbs_reset"])
    except (AttributeError, KeyError):
        return False


def rollout_steps_per_launch(raw):
    """k of mt_rollout on this engine (1 = one launch per step), from mt_describe_dispatch."""
    try:
        return int(raw.dispatch()["rollout"]["steps_per_launch"])
    except (AttributeError, KeyError):
        return 1


def measure(fab, raw, n_total, steps, warmup, episode_len, seed, fused=False, overlap=True, repeats=0, prewarm_s=PREWARM_S,
            min_timed_s=MIN_TIMED_S):
    """The timing protocol of the contract on one engine (this rank's shard; gather already attached): time-based
    pre-warm, `warmup` untimed steps, then `repeats` regions of EXACTLY `steps` steps, each bracketed by fence() on both
    sides; the region time is the max over ranks, the figure the median region.  Every region is also bracketed by HIP
    events on the engine's streams (device idle after the opening fence -> end of the region's last kernel / reset /
    exchange): `region_device_ms`, max over ranks, median.  Returns a dict of raw numbers."""
    eng = TimedEngine(raw)
    L = max(1, min(episode_len, steps))                         # >= 1 gather inside every timed region
    loop = EpisodeLoop(eng, seed, L, fused=fused, overlap=overlap, steps_per_launch=rollout_steps_per_launch(raw),
                       absorbs_reset=rollout_absorbs_reset(raw))
    prewarm = 0
    fab.fence(raw)
    t0 = time.perf_counter()
    while fab.rank0_says(time.perf_counter() - t0 < prewarm_s):
        loop.run(max(L, 200))
        prewarm += max(L, 200)
        raw.sync()
    est_s = float(fab.max_over_ranks([(time.perf_counter() - t0) / max(1, prewarm) * steps])[0])   # one region, estimated
    reps = repeats or int(min(300, max(5, math.ceil(min_timed_s / max(est_s, 1e-6)))))
    loop.run(warmup)                                            # the W untimed warm-up steps of the contract
    loop.align()                                                # the first timed region starts an episode
    fab.fence(raw)
    regions, device_ms, step_us_regions, gather_ms, launches, gathers = [], [], [], 0.0, 0, 0
    # The step laps are a stopwatch INSIDE the timed work: an event record in front of and behind the launches of every lap,
    # 7 us per lap (tools/lap_cost.py) -- 1 % of a 20-step region of 1 M arms, 10 % of one of a 131 072-arm shard (two laps).
    # Every LAP_EVERY-th region carries them (same calls, same launches in all regions), the others run without.
    lap_every = LAP_EVERY if reps >= 3 * LAP_EVERY else 1
    lapped = []
    for rep in range(reps):
        with_laps = rep % lap_every == 0
        eng.start_region()
        fab.fence(raw)
        raw.timer_start()                                       # start mark on the idle device (ahead of the host clock)
        t0 = time.perf_counter()
        ln, gt = loop.run(steps, time_kernels=True, laps=with_laps)
        raw.timer_stop_async()                                  # end marks behind everything queued; no join, no host wait
        fab.fence(raw)
        regions.append(time.perf_counter() - t0)
        device_ms.append(raw.timer_read())
        ms = eng.collect()
        lapped.append(with_laps)
        if with_laps:
            step_us_regions.append(ms["step"] * 1e3 / max(1, ln))   # device time per step of THIS region (sum of its step laps)
        # overlapped: device time of the region's last exchange on the side stream; in line: HIP-event laps around it
        gather_ms += (raw.gather_wait(host=True) * gt if gt else 0.0) if loop.overlap else ms["gather"]
        launches += ln
        gathers += gt
    regions = fab.max_over_ranks(regions)                       # a region takes as long as its slowest rank
    device_ms = fab.max_over_ranks(device_ms)
    # sanity on what was computed (not timed): returns are small integers, every rank's shard arrived
    tr = raw.total_reward()
    assert np.isfinite(tr).all() and np.all(tr == np.round(tr))
    assert loop.gathered is not None and loop.gathered.numel() == n_total
    g = loop.gathered.cpu().numpy()
    assert np.all(g == np.round(g)) and np.abs(g).max() <= L    # returns of an L-step episode
    elapsed = float(np.median(regions))
    lap_mask = np.asarray(lapped, dtype=bool)
    elapsed_lapped = float(np.median(np.asarray(regions)[lap_mask]))     # the regions that carried the stopwatch
    dev_s = float(np.median(device_ms)) * 1e-3
    step_us = float(np.median(step_us_regions))                 # median over regions: one stalled lap does not move it
    return {"elapsed": elapsed, "ms_per_step": elapsed / steps * 1e3, "ms_per_step_min": float(regions.min()) / steps * 1e3,
            "ms_per_step_max": float(regions.max()) / steps * 1e3, "step_us": step_us,
            "step_us_max_region": float(np.max(step_us_regions)),
            "launches": launches, "gather_us": gather_ms * 1e3 / max(1, gathers), "gathers_per_region": gathers // reps,
            "repeats": reps, "prewarm": prewarm, "episode_len": L, "value": n_total * steps / elapsed,
            "region_device_ms": dev_s * 1e3, "device_ms_per_step": dev_s * 1e3 / steps,
            "value_device_timeline": n_total * steps / max(dev_s, 1e-12),
            "steps_per_kernel_launch": launches / max(1, loop.kernel_launches),
            "episode_first_launches_outside_laps": loop.head_launches, "episode_first_launch_steps": loop.head_steps,
            "lapped_regions": int(lap_mask.sum()), "laps_every_nth_region": lap_every,
            "ms_per_step_lapped_regions": elapsed_lapped / steps * 1e3}


def self_launch(argv, gpus, dry_run=False, timeout_s=1800.0, script=None):
    """--gpus N > 1 without a launcher: start N fresh children of this script, one rank per GPU (what `python -m
    torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` would export), relay rank 0's JSON line to
    stdout and everything else to stderr, and return the worst exit code.  Called BEFORE anything imports torch or touches
    HIP (a process that has initialised the GPU must not start replacing or duplicating itself); children are plain
    subprocess.Popen of a new interpreter -- no exec of this process, no fork of GPU state.  The first failing rank, or the
    timeout, ends the whole group."""
    import signal
    import socket
    import subprocess
    with socket.socket() as sock:                               # a free rendezvous port
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    script = os.path.abspath(script or __file__)
    plans = []
    for r in range(gpus):
        env = {"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(gpus), "LOCAL_WORLD_SIZE": str(gpus),
               "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY":
               os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")}
        plans.append({"argv": [sys.executable, script] + [a for a in argv if a != "--launch-dry-run"], "env": env})
    if dry_run:
        print(json.dumps({"launcher": "bench.py self_launch", "children": plans}))
        return 0
    procs, out0, reader = [], [], None
    worst, why = 0, None
    try:
        for r, pl in enumerate(plans):
            procs.append(subprocess.Popen(pl["argv"], env={**os.environ, **pl["env"]}, cwd=os.getcwd(),
                                          stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr,
                                          start_new_session=True))
        import threading
        reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
        reader.start()
        deadline = time.monotonic() + timeout_s
        live = set(range(gpus))
        while live and why is None:
            for r in sorted(live):
                rc = procs[r].poll()
                if rc is None:
                    continue
                live.discard(r)
                if rc != 0:
                    worst, why = (rc if rc > 0 else 128 - rc), f"rank {r} exited with {rc}"
                    break
            if why is None and live and time.monotonic() > deadline:
                worst, why = 124, f"timeout after {timeout_s:.0f} s"
            if live and why is None:
                time.sleep(0.05)
        if why is not None:
            print(f"[bench launcher] {why}: ending the other ranks", file=sys.stderr)
    finally:
        # whoever is still alive here (a failure, the timeout, an interrupt of the launcher) goes, group by group: each
        # child leads its own session / process group.  After a normal end nobody is.
        for sig, wait_s in ((signal.SIGTERM, 10.0), (signal.SIGKILL, 5.0)):
            alive = [q for q in procs if q.poll() is None]
            if not alive:
                break
            for q in alive:
                try:
                    os.killpg(q.pid, sig)
                except (ProcessLookupError, PermissionError):
                    pass
            t_end = time.monotonic() + wait_s
            while time.monotonic() < t_end and any(q.poll() is None for q in alive):
                time.sleep(0.05)
    if reader is not None:
        reader.join(timeout=10.0)
    text = (out0[0] if out0 else b"").decode(errors="replace")
    line = None
    for ln in text.splitlines():
        try:
            if isinstance(json.loads(ln), dict) and "metric" in ln:
                line = ln
                continue
        except ValueError:
            pass
        print(ln, file=sys.stderr)
    if line is not None:
        sys.stdout.write(line + "\n")
        sys.stdout.flush()
    elif worst == 0:
        print("[bench launcher] rank 0 printed no JSON line", file=sys.stderr)
        worst = 1
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--envs-per-gpu", type=int, default=0,
                    help="weak scaling: arms owned by every GPU, as the headline (the default headline is 1 048 576 arms in "
                         "all: on one GPU that is the same job; with N > 1 the weak leg is in `secondary`)")
    ap.add_argument("--envs-total", type=int, default=0,
                    help="strong scaling: total arms, sharded over the GPUs, as the headline (default 1 048 576 = the metric's "
                         "workload; --gpus 8 --envs-total 4194304 = BASELINE.json configs[3], which an N > 1 run also "
                         "reports as a secondary leg)")
    ap.add_argument("--dof", type=int, default=4, choices=(4, 7))
    ap.add_argument("--targets", type=int, default=7)
    ap.add_argument("--episode-len", type=int, default=50)       # test_multi.py:8
    ap.add_argument("--repeats", type=int, default=0, help="timed regions (0 = enough for >= 250 ms of GPU work, 5..300)")
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EED)
    ap.add_argument("--sync-gather", action="store_true",
                    help="run the return gather in line on the engine's stream instead of overlapped on its side stream")
    ap.add_argument("--reset-before-gather", action="store_true",
                    help="experiment: at an episode end reset first, then gather MT_F_LAST_RETURN in place (no snapshot copy) "
                         "instead of snapshot + gather, then reset")
    ap.add_argument("--episode-phase", type=int, default=-1,
                    help="steps of the running episode already done when a timed region starts (0: every region is one "
                         "whole segment and ends with its gather + reset; p: the episode ends L - p steps into the region "
                         "and the exchange runs beside the p steps behind it).  Default: 0 on one GPU (nothing to hide: "
                         "40.9 vs 42.0 us per step, profiles/r03_ab_episode_phase.txt), half an episode with N > 1 (the "
                         "RCCL all-gather of a region is then hidden behind steps instead of waited for at the fence)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="torch.distributed backend of the CONTROL plane for N > 1 (barrier, max over ranks, shipping "
                         "the RCCL unique id); the return gather itself always goes through mt_gather_returns = RCCL, "
                         "unless --rehearsal")
    ap.add_argument("--rehearsal", action="store_true",
                    help="1-GPU box only: every rank on GPU 0, control AND gather over gloo (RCCL refuses two ranks on "
                         "one device); checks the N > 1 control flow, not a scaling number")
    ap.add_argument("--single-device", action="store_true",
                    help="testing only: every rank on GPU 0 (with MT_RCCL_LIB pointing at the stand-in transport of "
                         "tests/fake_rccl this runs the whole N > 1 product path on a one-GPU box)")
    ap.add_argument("--launch-dry-run", action="store_true",
                    help="print what the launcher would start for --gpus N (argv and rank environment of every child) and exit")
    ap.add_argument("--launch-timeout", type=float, default=1800.0, help="seconds the self-launched ranks get in all")
    ap.add_argument("--hw-trig", action="store_true")
    ap.add_argument("--dh-in-lds", action="store_true")
    ap.add_argument("--direct-trig", action="store_true")
    ap.add_argument("--no-specialize", action="store_true")
    ap.add_argument("--fused", action="store_true",
                    help="secondary mode: each episode segment is ONE mt_rollout_fused launch (state kept on chip "
                         "between steps); the roofline object then describes that kernel")
    ap.add_argument("--ablate", type=int, default=0, help="diagnostic builds (results invalid): 1 skip interior "
                    "sub-steps, 2 also skip the observation math")
    args = ap.parse_args()

    # --gpus N without a launcher: this process only starts the ranks (nothing above this line has touched torch or HIP)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(sys.argv[1:], args.gpus, args.launch_dry_run, args.launch_timeout))
    if args.launch_dry_run:
        print(json.dumps({"launcher": "none (one rank, or WORLD_SIZE already set by a launcher)", "children": []}))
        return

    # The contract is ONE JSON line on stdout.  Native libraries loaded below write there too (RCCL prints a version
    # banner when its first multi-rank communicator is built, gloo its connection report), so fd 1 is pointed at
    # stderr for the rest of the run and the line goes to the original stdout at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    # multi-process GPU work on this pool needs dmabuf IPC (RCCL's hipIpcGetMemHandle fails with the legacy mode); the
    # launcher normally exports it already
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import torch
    import torch.distributed as dist

    import manytor_amd as m
    from manytor_amd import distributed as D

    EpisodeLoop.reset_first = args.reset_before_gather
    EpisodeLoop.phase = args.episode_phase if args.episode_phase >= 0 else (0 if args.gpus == 1 else min(args.episode_len, args.steps) // 2)
    rank, local_rank, world = D.env_from_torchrun()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher started a different number of ranks")
    if not torch.cuda.is_available() or m.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: the step path has no CPU fallback")
    # one process per GPU: LOCAL_RANK names the device, unless the launcher already narrowed every process's view to
    # its own card (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES per rank), in which case there is only device 0
    dev = 0 if (args.rehearsal or args.single_device) else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev)
    backend = "gloo" if args.rehearsal else args.backend
    if world > 1:
        if backend == "nccl":
            D.init_process_group("nccl", device=dev)
        else:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    ctrl_dev = "cuda" if backend == "nccl" else "cpu"
    # Barrier of the timed regions at N > 1: a shared-memory barrier between the ranks of the node (microseconds); the
    # process group's collective barrier (tens of microseconds inside a 0.9 ms region) only if that cannot be set up.
    host_barrier = D.make_host_barrier(rank, world) if world > 1 else None
    fab = Fabric(rank, world, dist if world > 1 else None, torch, ctrl_dev, host_barrier, torch.cuda.synchronize)

    table = m.REF_DH_TABLE if args.dof == 4 else m.DH7_TABLE
    radius = 51.3 if args.dof == 4 else 92.6
    bpe = algorithmic_bytes_per_env_step(args.dof, args.targets)
    bpe_actual = actual_bytes_per_env_step(args.dof, args.targets)

    def make_engine(n_total, strong):
        """This rank's shard of `n_total` envs (strong) or its own n_total / world (weak), with its gather attached:
        mt_comm_init over RCCL, or -- agreed on by all ranks -- the stand-ins.  -> (engine, n_local, how the gather runs)."""
        if strong:
            base, n_local = D.shard_range(n_total, rank, world)
        else:
            n_local = n_total // world
            base = rank * n_local
        raw = m.StepEngine(n_local, args.targets, dh_table=table, radius=radius, device=dev, env_id_base=base,
                           hw_trig=args.hw_trig, dh_in_lds=args.dh_in_lds, direct_trig=args.direct_trig,
                           specialize=not args.no_specialize, ablate=args.ablate)
        # The engine keeps its own stream: step launches, the RCCL gather and the resets are all enqueued on it through the
        # C ABI, in program order; torch only brackets the timed regions (barrier + torch.cuda.synchronize()).
        fallback = None
        if world > 1 and not args.rehearsal:
            err = None
            try:
                D.connect(raw, rank, world)                     # mt_comm_unique_id -> store -> mt_comm_init (RCCL)
            except Exception as e:                              # noqa: BLE001 -- reported below, on every rank
                err = e
            # Contingency (never taken where mt_comm_init works): if ANY rank could not join the communicator, all ranks
            # drop it and gather through the torch.distributed process group instead, and the JSON line says so.
            if fab.any_rank(err is not None):
                print(f"[bench] rank {rank}: mt_comm_init path unavailable ({err}); gathering through torch.distributed "
                      f"({backend}) instead", file=sys.stderr)
                if err is None:
                    raw.comm_destroy()
                if backend == "nccl":
                    D.attach_torch_gather(raw, n_total, rank, world)
                else:
                    D.attach_gloo_gather(raw, n_total, rank, world)
                fallback = f"mt_comm_init failed on at least one rank ({err if err is not None else 'another rank'})"
        elif world > 1:
            D.attach_gloo_gather(raw, n_total, rank, world)     # rehearsal stand-in for the RCCL gather
        return raw, n_local, fallback

    def describe_collective(fallback):
        if world == 1:
            return "none (1 GPU: mt_gather_returns is a device copy)"
        if args.rehearsal:
            return "gloo all-gather (REHEARSAL on one device, not RCCL)"
        if fallback:
            return f"FALLBACK: torch.distributed ({backend}) all-gather of the returns per episode, in line; {fallback}"
        if os.environ.get("MT_RCCL_LIB", "").endswith("libfake_rccl.so"):
            return ("mt_gather_returns (C ABI) over the shared-memory STAND-IN for librccl (tests/fake_rccl): a rehearsal of "
                    "the N > 1 product path on one GPU, not a scaling number")
        return "RCCL all-gather of returns per episode through mt_gather_returns (C ABI), straight from the arena"

    def gather_mode(fallback):
        if fallback or args.rehearsal:
            return "in line (the stand-in gathers do not overlap anything)"
        if args.reset_before_gather and not args.sync_gather:
            return ("overlapped: the reset stores the finished returns (MT_F_LAST_RETURN), the exchange reads them in place on "
                    "the engine's side stream (mt_gather_returns_begin_inplace) beside the next episode's steps")
        return ("in line on the engine's stream" if args.sync_gather else
                "overlapped: snapshot (per chain) + exchange on the engine's side stream (mt_gather_returns_begin), beside "
                "the reset and the next episode's steps")

    def regime_of(n_local, bytes_per_env):
        mb = bytes_per_env * n_local / 1e6
        if mb < 40:
            return ("on-die: the %.0f MB a step touches fit the L2s (8 x 4 MB) and the 256 MiB Infinity Cache, the step is bound "
                    "by the kernel boundary and the dependent arithmetic of a few waves per SIMD, not by HBM" % mb)
        if mb < 400:
            return ("HBM + Infinity Cache: the %.0f MB a step touches are within (or about) the size of the 256 MiB MALL, so part "
                    "of the re-read state is served on-die; the pure-HBM point is the 4 194 304-arm row" % mb)
        return "HBM (working set of %.0f MB per step, beyond the Infinity Cache)" % mb

    def leg_record(r, f, n_total, n_local, strong, label, name, fallback):
        """What every leg of an N > 1 line reports (and the N > 1 headline inside `secondary.headline_detail`)."""
        moved = moved_bytes_per_env_step(args.dof, args.targets, r["steps_per_kernel_launch"])
        rec = {"what": label, "scaling": "strong" if strong else "weak", "envs_total": n_total, "envs_on_rank0": n_local,
               "n_gpus": world, "value": r["value"], "value_device_timeline": r["value_device_timeline"], "unit": "env-steps/s",
               "ms_per_step": r["ms_per_step"], "ms_per_step_min": r["ms_per_step_min"], "ms_per_step_max": r["ms_per_step_max"],
               "device_ms_per_step": r["device_ms_per_step"],
               "avg_kernel_us": r["step_us"], "kernel": name, "steps_per_kernel_launch": r["steps_per_kernel_launch"],
               "gather_us": r["gather_us"], "gathers_in_timed_region": r["gathers_per_region"], "repeats": r["repeats"],
               "ms_per_step_lapped_regions": r["ms_per_step_lapped_regions"], "lapped_regions": r["lapped_regions"],
               "episode_len": r["episode_len"], "episode_phase": EpisodeLoop.phase % r["episode_len"],
               "collective": describe_collective(fallback), "gather_mode": gather_mode(fallback),
               # per-GPU fraction of the HBM peak while a step is on the device, on the bytes the launches really move
               "bytes_per_env_step": moved,
               "frac_of_hbm_peak_per_gpu": moved * n_local / (r["step_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
               "regime": regime_of(n_local, moved)}
        if f is not None:
            rec["fused"] = {"value": f["value"], "value_device_timeline": f["value_device_timeline"], "ms_per_step": f["ms_per_step"],
                            "device_ms_per_step": f["device_ms_per_step"], "avg_kernel_us": f["step_us"],
                            "gather_us": f["gather_us"], "gathers_in_timed_region": f["gathers_per_region"],
                            "kernel": "rollout_kernel (mt_rollout_fused: one launch per episode segment)", "repeats": f["repeats"]}
            rec["fused_us_per_step"] = f["step_us"]
        return rec

    def secondary_leg(n_total, strong, label):
        """A further configuration on the same ranks: fresh engines + ONE communicator of their own, the same episode loop,
        real gather and fences as the headline (shorter pre-warm), first through mt_rollout, then -- same engines, same
        communicator -- as fused segments."""
        raw, n_local, fallback = make_engine(n_total, strong)
        kw = dict(overlap=not args.sync_gather, prewarm_s=0.15, min_timed_s=0.03)
        r = measure(fab, raw, n_total, args.steps, args.warmup, args.episode_len, args.seed, fused=False, **kw)
        f = measure(fab, raw, n_total, args.steps, args.warmup, args.episode_len, args.seed, fused=True, **kw)
        name = raw.step_kernel_name()
        raw.close()
        return leg_record(r, f, n_total, n_local, strong, label, name, fallback)

    # ---- the headline: the metric's workload -- 1 048 576 arms in all -- unless --envs-total / --envs-per-gpu ----------
    if args.envs_total > 0:
        strong, n_total = True, args.envs_total
    elif args.envs_per_gpu > 0:
        strong, n_total = False, args.envs_per_gpu * world
    else:
        strong, n_total = world > 1, 1048576     # one GPU: strong and weak are the same job (reported as "weak", as before)
    raw, n_local, gather_fallback = make_engine(n_total, strong)
    r = measure(fab, raw, n_total, args.steps, args.warmup, args.episode_len, args.seed, fused=args.fused,
                overlap=not args.sync_gather, repeats=args.repeats)
    if args.ablate:
        print("ABLATION BUILD: timings only, outputs are not the reference's", file=sys.stderr)
    L = r["episode_len"]
    kernel_name = raw.step_kernel_name()
    dispatch = raw.dispatch()
    raw.close()

    out = None
    if rank == 0:
        step_s = r["step_us"] * 1e-6                            # device time of one step of this rank's envs (HIP events)
        # mt_rollout may run a step as several concurrent launches on env ranges (the dispatch says so)
        chains = 1 if args.fused else int(dispatch["rollout"]["chains"])
        envs_per_launch = n_local if chains == 1 else int(dispatch["chains"]["span"])
        spl = r["steps_per_kernel_launch"]                      # 1 = one launch per step; L for --fused; k on small shards
        moved = moved_bytes_per_env_step(args.dof, args.targets, spl)
        achieved = moved * n_local / step_s / 1e9
        achieved_model = bpe * n_local / step_s / 1e9
        trig = 2 if args.hw_trig else (1 if args.direct_trig else 0)
        static = not (args.no_specialize or args.dh_in_lds)
        table_name = ("Ref4Table" if args.dof == 4 else "Dh7Table") if static else f"RtTable<{args.dof}>"
        variant = "+".join(v for v, on in (("recurrence", trig == 0), ("direct_trig", trig == 1), ("hw_trig", trig == 2),
                                           ("static_table", static), ("runtime_table", not static),
                                           ("dh_in_lds", args.dh_in_lds)) if on)
        workload = f"{n_total} arms in all = {n_local} arms/GPU x {world} GPU, {args.dof}-DoF DH chain, K={args.targets} targets, " \
                   f"25 sub-steps, random integer-degree actions drawn in-kernel, episode {L} steps"
        # the streaming yardstick of the step's own access shape: pure HBM (4 M envs, ~1 GB per pass) and at this size
        ach_hbm = achievable_bandwidth(m, args.dof, args.targets, 4194304, dev)
        ach_same = achievable_bandwidth(m, args.dof, args.targets, max(256, n_local), dev)
        out = {
            "metric": METRIC, "value": r["value"], "unit": "env-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_step"],
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "value_device_timeline": r["value_device_timeline"], "device_ms_per_step": r["device_ms_per_step"],
            "config": {"workload": workload, "envs_per_gpu": n_local, "envs_total": n_total, "dof": args.dof,
                       "targets": args.targets, "substeps": 25, "episode_len": L, "episode_phase": EpisodeLoop.phase % L,
                       "collective": describe_collective(gather_fallback),
                       "kernel_variant": variant, "prewarm_launches": r["prewarm"], "repeats": r["repeats"],
                       "laps_every_nth_region": r["laps_every_nth_region"], "lapped_regions": r["lapped_regions"],
                       "ms_per_step_lapped_regions": r["ms_per_step_lapped_regions"],
                       "laps_note": "the HIP-event laps that time the step launches (roofline.avg_kernel_us) are a stopwatch inside "
                                    "the timed work: an event record in front of and behind each lap's launches, 7 us per lap "
                                    "(tools/lap_cost.py).  Every laps_every_nth_region-th region carries them, the others run the "
                                    "same calls and launches without; ms_per_step / value are the median over ALL regions, "
                                    "ms_per_step_lapped_regions the median of those with the stopwatch",
                       "gathers_in_timed_region": r["gathers_per_region"], "gather_mode": gather_mode(gather_fallback),
                       "launcher": os.environ.get("TORCHELASTIC_RUN_ID") and "torch.distributed.run" or
                                   ("bench.py self_launch (one fresh child process per rank)" if world > 1 else "none (one process)"),
                       "timing": "median over `repeats` regions of exactly `steps` steps, each bracketed by "
                                 "torch.cuda.synchronize() + barrier (device drained, then every rank arrived), max over ranks; value_device_timeline = the same regions on the "
                                 "device's clock (HIP events from the idle device at the region's start to the end of its "
                                 "last kernel / reset / exchange, max over ranks, median)",
                       "barrier": "none (one rank)" if world == 1 else
                                  ("device drained, then a shared-memory barrier between the node's ranks "
                                   "(manytor_amd.distributed.HostBarrier)" if host_barrier is not None else
                                   "device drained, then torch.distributed barrier"),
                       "dispatch": {k: dispatch[k] for k in ("step", "chains", "rollout", "fused", "reset", "overrides")}},
            "ms_per_step_min": r["ms_per_step_min"], "ms_per_step_max": r["ms_per_step_max"],
            "gather_us": r["gather_us"],
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": load_traffic(f"d{args.dof}_k{args.targets}_n{n_local}", spl),
                "traffic_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, committed under "
                                  "profiles/ (traffic.json); not re-measured in this run.  Per STEP where a step is one or two "
                                  "launches (bytes_per_step is its counterpart), per LAUNCH of steps_per_kernel_launch steps "
                                  "where mt_rollout runs several steps per launch (counterpart: bytes_per_launch)",
                "kernel": (f"rollout_kernel<{table_name}> ({L} steps per launch; us per step quoted)" if args.fused else
                           kernel_name + " (action drawn in-kernel)"),
                "bytes_per_env_step": moved, "avg_kernel_us": r["step_us"], "avg_kernel_us_max_region": r["step_us_max_region"],
                "steps_timed": r["launches"],
                "episode_first_launches_outside_laps": r["episode_first_launches_outside_laps"],
                "episode_first_launch_steps": r["episode_first_launch_steps"],
                "steps_per_kernel_launch": spl,
                "launches_per_step": chains, "envs_per_launch": envs_per_launch,
                "bytes_per_launch": moved * spl * envs_per_launch, "bytes_per_step": moved * n_local,
                "bytes_per_env_step_survey_model": bpe, "achieved_survey_model": achieved_model,
                "frac_survey_model": fraction_or_none(achieved_model / HBM_PEAK_GBS),
                "achievable_gbs": ach_hbm,
                "achievable_note": "mt_stream_probe: the memory operations of one step (same rows read / rewritten / nt-written "
                                   "per env, same addressing, one env per lane) with no arithmetic, over 4 194 304 envs "
                                   "(0.98 GB per pass: pure HBM) -- what this access shape reaches on this box; "
                                   "achievable_gbs_same_size is the same probe over envs_per_gpu envs (one launch)",
                "achievable_gbs_same_size": ach_same,
                "frac_of_achievable_same_size": (achieved / ach_same) if ach_same else None,
                "regime": regime_of(n_local, moved),
                "note": "achieved / frac use the bytes the timed launches really move per env-step: SURVEY 8(d)'s 12 D + 24 K + "
                        "33 minus the 4 D-byte action read that does not happen (the action is drawn in registers and IS the "
                        "new goals), and -- when mt_rollout runs k steps per launch (steps_per_kernel_launch > 1: small shards) "
                        "-- the state rows once per launch instead of once per step; *_survey_model are the same time with "
                        "SURVEY's own 249-byte figure.  avg_kernel_us = device time of ONE STEP of this rank's envs by HIP "
                        "events around the step launches, fork and join of the chains included; where mt_rollout absorbs the "
                        "episode's reset into its first launch (config.dispatch.rollout.absorbs_reset), that launch -- reset "
                        "prologue + episode_first_launch_steps steps, another kernel -- is issued outside these laps (inside "
                        "the region and its device timeline).  With launches_per_step = 2 a "
                        "step is two CONCURRENT launches of envs_per_launch envs on two streams: rocprofv3's per-launch "
                        "average is then the duration of each of two overlapping kernels, not half a step; the union of "
                        "their intervals per step (tools/trace_summary.py --union) corresponds to avg_kernel_us",
            },
        }
    secondary = world == 1 and not args.fused and not args.ablate and not args.no_secondary
    if secondary and rank == 0:
        # informational, not the headline: the same episodes as ONE launch each
        us, _ = time_step_launches(m, n_local, table, radius, args.targets, dev, args.seed, fused=True, steps=1000)
        out["secondary"] = {"fused_rollout": {
            "env_steps_per_s": n_local / (us * 1e-6), "us_per_step": us, "steps_per_launch": 50,
            "note": "mt_rollout_fused: state stays in registers/LDS between steps, bit-identical results; "
                    "arithmetic-bound, so the per-step byte model does not apply"}}
        if dispatch["rollout"]["steps_per_launch"] == 1 and dispatch["fused"]["usable"]:
            # what the k-steps-per-launch form of small shards (kPolicy.multi_step_k) would give at THIS size: not the
            # default here -- the headline stays the one-launch-per-step, HBM-bound path SURVEY 8(d)'s byte model describes
            kk = int(dispatch["policy"]["multi_step_k"])
            us_k, name_k, spl_k = time_step_launches(m, n_local, table, radius, args.targets, dev, args.seed, steps=600,
                                                     want_spl=True, rollout_k=kk)
            mv = moved_bytes_per_env_step(args.dof, args.targets, spl_k)
            out["secondary"]["multi_step_rollout"] = {
                "steps_per_launch": kk, "us_per_step": us_k, "env_steps_per_s": n_local / (us_k * 1e-6), "kernel": name_k,
                "bytes_per_env_step": mv, "frac_of_hbm_peak": mv * n_local / (us_k * 1e-6) / 1e9 / HBM_PEAK_GBS,
                "note": f"MT_ROLLOUT_K={kk}: mt_rollout with {kk} steps per launch through the rollout kernels at the headline's "
                        "size (the default only up to 262 144 envs): the state rows cross HBM once per launch, the step becomes "
                        "arithmetic-bound; same bits"}
        us, us_sample, us_pair, lname = time_loaded_action_steps(m, n_local, table, radius, args.targets, dev, args.seed)
        out["secondary"]["loaded_action_step"] = {
            "us_per_step": us, "sample_actions_us": us_sample, "pair_us": us_pair, "kernel": lname,
            "env_steps_per_s": n_local / (us * 1e-6), "bytes_per_env_step": bpe,
            "frac_of_hbm_peak": bpe * n_local / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "frac_of_hbm_peak_pair": (bpe + 4 * args.dof) * n_local / (us_pair * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "note": "mt_step with the actions read from HBM (policy-in-the-loop shape, step_kernel<SAMPLE = false>): "
                    "the kernel that moves exactly the SURVEY 8(d) bytes, issued per chain like mt_rollout's steps; the "
                    "actions are written by a separate mt_sample_actions launch (per chain as well).  TIMED is the (sample, "
                    "step) pair (pair_us; frac_of_hbm_peak_pair = against the bytes of both launches, SURVEY's 249 + the 4 "
                    "D-byte action write); us_per_step = pair_us minus the sampler's cheapest form (sample_actions_us: one "
                    "launch over the batch, timed alone), i.e. an upper bound of the step's own time"}
        # BASELINE.json's other single-GPU configurations and the pure-HBM point, same kernel path, short runs.  Every
        # configuration is measured on two fresh engines, once in this order and once more in the reverse order at the end
        # (us_per_step = the faster pass, both are listed): the 4 M-arm arena reads 171 or 193 us by where its pages landed
        # (DESIGN.md section 5), and a second look at every figure is cheap.
        others = (("configs[1]: 65536 arms, 4-DoF", 65536, m.REF_DH_TABLE, 51.3),
                  ("131072 arms, 4-DoF (1 M arms over 8 GPUs, per-GPU shard)", 131072, m.REF_DH_TABLE, 51.3),
                  ("524288 arms, 4-DoF (configs[3] over 8 GPUs, per-GPU shard)", 524288, m.REF_DH_TABLE, 51.3),
                  ("configs[4]: 1048576 arms, 7-DoF table", 1048576, m.DH7_TABLE, 92.6),
                  ("4194304 arms, 4-DoF (1 GB working set: pure HBM regime)", 4194304, m.REF_DH_TABLE, 51.3))
        passes = {}
        for order in (others, others[::-1]):
            for label, n2, tbl, rad in order:
                passes.setdefault(label, []).append(time_step_launches(m, n2, tbl, rad, args.targets, dev, args.seed,
                                                                       steps=300 if n2 > (1 << 21) else 600, want_spl=True))
        out["secondary"]["other_configs"] = {}
        for label, n2, tbl, rad in others:
            us, kname, spl2 = min(passes[label], key=lambda p_: p_[0])
            b2 = algorithmic_bytes_per_env_step(len(tbl), args.targets)
            b2a = moved_bytes_per_env_step(len(tbl), args.targets, spl2)
            gbs = b2a * n2 / (us * 1e-6) / 1e9
            rec = {"us_per_step": us, "us_per_step_passes": [p_[0] for p_ in passes[label]], "kernel": kname,
                   "env_steps_per_s": n2 / (us * 1e-6), "steps_per_kernel_launch": spl2,
                   "bytes_per_env_step": b2a, "frac_of_hbm_peak": gbs / HBM_PEAK_GBS,
                   "bytes_per_env_step_survey_model": b2,
                   "survey_model_gbs": b2 * n2 / (us * 1e-6) / 1e9,
                   "frac_survey_model": fraction_or_none(b2 * n2 / (us * 1e-6) / 1e9 / HBM_PEAK_GBS), "regime": regime_of(n2, b2a)}
            if spl2 > 1:    # the pure one-launch-per-step figure beside the k-steps-per-launch default (MT_ROLLOUT_K=1)
                us1, kname1 = time_step_launches(m, n2, tbl, rad, args.targets, dev, args.seed, steps=600, rollout_k=1)
                rec["one_launch_per_step"] = {"us_per_step": us1, "kernel": kname1, "env_steps_per_s": n2 / (us1 * 1e-6),
                                              "bytes_per_env_step": actual_bytes_per_env_step(len(tbl), args.targets)}
            if n2 == 4194304 and len(tbl) == args.dof and out["roofline"]["achievable_gbs"]:
                rec["achievable_gbs"] = out["roofline"]["achievable_gbs"]
                rec["frac_of_achievable"] = gbs / out["roofline"]["achievable_gbs"]
            out["secondary"]["other_configs"][label] = rec
    if world > 1 and not args.fused and not args.ablate and not args.no_secondary:
        # One N > 1 invocation yields every figure BASELINE.json's north_star names: the headline is the metric's 1 M arms
        # sharded over these ranks; here the weak leg (1 M arms PER GPU) and configs[3] (4 194 304 arms sharded), each
        # through mt_rollout and as fused segments.  Collective code: every rank runs every leg.
        legs = {}
        for key, nt, st, label in (("strong_1m", 1048576, True, "north_star: 1 048 576 arms sharded over the GPUs (the 1 -> N curve)"),
                                   ("weak_1m_per_gpu", 1048576 * world, False, "weak scaling: 1 048 576 arms per GPU"),
                                   ("config3", 4194304, True, "BASELINE.json configs[3]: 4 194 304 arms sharded over the GPUs")):
            if nt == n_total and (st == strong or nt % world == 0):
                continue                                        # that IS the headline of this invocation
            legs[key] = secondary_leg(nt, st, label)
        if rank == 0:
            out["secondary"] = legs
            out["secondary"]["note"] = ("`value` of the JSON line is the leg named in config.workload (default: the metric's "
                                        "1 048 576 arms sharded over the GPUs, scaling = strong); each leg here states its own "
                                        "scaling.  value = wall clock of the fenced regions; value_device_timeline = the same "
                                        "regions on the device's clock (what the GPUs sustain once the host's fences are out "
                                        "of the picture: a learner that does not fence every 20 steps sees this one)")
    if not args.no_cpu_baseline:
        # rank 0 times the CPU ports (every host core it may use); its peers wait asleep, not spinning on those cores
        if rank == 0:
            out["cpu_baseline"] = cpu_baseline(table, args.targets)
        if world > 1:
            if host_barrier is not None:
                host_barrier.wait(timeout_s=1800.0, sleep_s=0.02)
            else:
                dist.barrier()
    if rank == 0:
        os.write(json_fd, (json.dumps(out) + "\n").encode())

    if host_barrier is not None:
        host_barrier.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
