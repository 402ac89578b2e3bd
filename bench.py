#!/usr/bin/env python3
"""Benchmark of the hot path: env-steps/sec of the batched manipulator step on MI355X.

    python bench.py                                   # 1 GPU, 1 048 576 arms, 4-DoF, K=7
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one Environment.step() (manytor.py:255-260: 25 interpolated sub-steps of DH forward
kinematics, ground flag, observation, pickup, reward, return) for EVERY env of the batch, with the
random action drawn in the same launch (manytor.py:215-217).  Inputs are synthetic and already
resident in HBM when the timed region starts.  Every `episode_len` steps the returns are gathered
over the ranks (RCCL all-gather, the only collective) and all envs are reset (test_multi.py:32-34).

Prints ONE JSON line on rank 0 (contract: see the task brief / DESIGN.md section "Measurement").
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "env-steps/sec (whole node), 1M parallel 4-DoF arms, random actions"
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def algorithmic_bytes_per_env_step(dof, k):
    """reads action 4D + goals 4D + points 12K + alive 4 + return 4; writes goals 4D +
    obs 12K + reward 4 + done 1 + alive 4 + return 4 + end effector 12."""
    return 12 * dof + 24 * k + 33


def cpu_baseline(dof_table, k, budget_s=12.0, n=65536, threads=16):
    """The CPU port of the step path (oracle/: parity-checked against the reference's fixtures), timed on this box's
    host cores on a bounded sample of the same workload.  Headline figure: the C restatement with OpenMP on the
    box's CPU share; beside it the vectorised-numpy and the scalar (reference-shaped) Python figures on one core.
    Baseline, not the target."""
    from oracle import c_oracle
    from oracle import manytor_oracle as mo
    from oracle import philox_ref as px
    table = np.asarray(dof_table)
    dof = table.shape[0]
    threads = max(1, min(threads, os.cpu_count() or 1))
    # (a) C + OpenMP
    nc = 1 << 20
    idc = np.arange(nc, dtype=np.uint64)
    cora = c_oracle.COracle(nc, k, table=table, threads=threads)
    cora.reset(px.sample_targets(0x5EED, idc, 0, k, 51.3).astype(np.float64))
    cacts = [px.sample_actions(0x5EED, idc, t, dof).astype(np.float64) for t in range(4)]
    cora.step(cacts[0])
    t0 = time.perf_counter()
    csteps = 0
    while csteps < 1 or (time.perf_counter() - t0 < budget_s and csteps < 200):
        cora.step(cacts[csteps % 4])
        csteps += 1
    dt_c = time.perf_counter() - t0
    # (b) vectorised numpy, one core
    ids = np.arange(n, dtype=np.uint64)
    ora = mo.BatchOracle(n, k, table=table)
    ora.reset(px.sample_targets(0x5EED, ids, 0, k, 51.3).astype(np.float64))
    acts = [px.sample_actions(0x5EED, ids, t, dof).astype(np.float64) for t in range(16)]
    ora.step(acts[0])
    t0 = time.perf_counter()
    steps = 0
    while steps < 2 or (time.perf_counter() - t0 < budget_s / 3 and steps < 15):
        ora.step(acts[1 + steps])
        steps += 1
    dt = time.perf_counter() - t0
    # (c) the reference's own shape of the computation: one env at a time, three FK chains per sub-step (manytor.py:188)
    envs = [mo.ScalarEnv(k, table=table) for _ in range(16)]
    pts = px.sample_targets(0x5EED, ids[:16], 0, k, 51.3)
    for e, p in zip(envs, pts):
        e.reset(points=p)
    t1 = time.perf_counter()
    for t in range(12):
        for j, e in enumerate(envs):
            e.step(acts[t][j])
    dt_scalar = time.perf_counter() - t1
    return {
        "value": nc * csteps / dt_c, "unit": "env-steps/s", "cores": threads, "kind": "port",
        "sample": f"{nc} envs x {csteps} steps, C restatement of the reference with OpenMP (oracle/manytor_oracle.c), "
                  f"{dt_c:.1f} s on {threads} of {os.cpu_count()} host cores",
        "numpy_vectorised_value_1core": n * steps / dt,
        "numpy_vectorised_sample": f"{n} envs x {steps} steps, oracle BatchOracle fp64, {dt:.1f} s",
        "scalar_faithful_value_1core": 16 * 12 / dt_scalar,
        "scalar_faithful_sample": "16 envs x 12 steps, per-env Python loop with the reference's 75 FK chains per step "
                                  "(oracle ScalarEnv)",
    }


def measured_copy_bandwidth(torch, nbytes=1 << 30, reps=20):
    """Device-to-device copy rate on this GPU (read + write bytes per second): the practical HBM ceiling that
    SURVEY.md 8(d) asks to quote next to the 8 TB/s spec figure."""
    src = torch.empty(nbytes // 4, dtype=torch.float32, device="cuda").normal_()
    dst = torch.empty_like(src)
    for _ in range(3):
        dst.copy_(src)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(reps):
        dst.copy_(src)
    ev1.record()
    torch.cuda.synchronize()
    return 2.0 * nbytes * reps / (ev0.elapsed_time(ev1) * 1e-3) / 1e9


def load_traffic(workload_key):
    """HBM bytes per step launch from the committed PMC run (profiles/), or None."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(workload_key, {}).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--envs-per-gpu", type=int, default=1048576)
    ap.add_argument("--dof", type=int, default=4, choices=(4, 7))
    ap.add_argument("--targets", type=int, default=7)
    ap.add_argument("--episode-len", type=int, default=50)       # test_multi.py:8
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EED)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="collective backend for N > 1: nccl = RCCL over xGMI (default); gloo = CPU rehearsal")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal only: put every rank on GPU 0 (use with --backend gloo on a 1-GPU box)")
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

    import torch
    import torch.distributed as dist

    import manytor_amd as m
    from manytor_amd import distributed as D

    rank, local_rank, world = D.env_from_torchrun()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available() or m.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: the step path has no CPU fallback")
    dev = 0 if args.single_device else local_rank
    torch.cuda.set_device(dev)
    if world > 1:
        if args.backend == "nccl":
            D.init_process_group("nccl")
        else:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    n_local = args.envs_per_gpu
    n_total = n_local * world                                   # weak scaling: per-GPU work fixed
    base = rank * n_local
    table = m.REF_DH_TABLE if args.dof == 4 else m.DH7_TABLE
    radius = 51.3 if args.dof == 4 else 92.6
    eng = m.StepEngine(n_local, args.targets, dh_table=table, radius=radius, device=dev, env_id_base=base,
                       hw_trig=args.hw_trig, dh_in_lds=args.dh_in_lds, direct_trig=args.direct_trig,
                       specialize=not args.no_specialize, ablate=args.ablate)
    eng.use_torch_stream()                                      # engine launches and torch/RCCL share one ordering
    returns = eng.device_tensor(m.lib.F_TOTAL_REWARD)
    L = args.episode_len

    state = {"step": 0, "episode": 0, "gathered": None}
    eng.reset_random(args.seed, 0)

    def run_steps(count, kernel_ms=None):
        """`count` env steps; episode boundary every L steps (gather returns over ranks, reset all)."""
        done = 0
        while done < count:
            seg = min(count - done, L - state["step"] % L)
            if kernel_ms is not None:
                eng.lap_begin()             # HIP events on the engine's stream, no host synchronisation
            if args.fused:
                eng.rollout_fused(seg, args.seed, state["step"])
            else:
                eng.rollout(seg, args.seed, state["step"])
            if kernel_ms is not None:
                eng.lap_end()
                kernel_ms.append(seg)
            state["step"] += seg
            done += seg
            if state["step"] % L == 0:
                src = returns if args.backend == "nccl" or world == 1 else returns.cpu()
                state["gathered"] = D.gather_returns(src, n_total)          # RCCL all-gather (identity at N=1)
                state["episode"] += 1
                eng.reset_random(args.seed, state["episode"])

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(args.warmup)
    D.gather_returns(returns if args.backend == "nccl" or world == 1 else returns.cpu(), n_total)   # warm-up, untimed
    fence()
    kernel_ms = []
    t0 = time.perf_counter()
    run_steps(args.steps, kernel_ms)
    fence()
    elapsed = time.perf_counter() - t0
    laps_ms, _ = eng.laps_total()                     # device time of the step launches inside the timed region
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # sanity on what was computed (not timed): returns are small integers, something happened
    tr = eng.total_reward()
    assert np.isfinite(tr).all() and np.all(tr == np.round(tr))
    if args.ablate:
        print("ABLATION BUILD: timings only, outputs are not the reference's", file=sys.stderr)
    if state["gathered"] is not None:
        assert state["gathered"].numel() == n_total

    if rank == 0:
        bpe = algorithmic_bytes_per_env_step(args.dof, args.targets)
        launches = sum(kernel_ms)                     # env-steps per env; = launches of step_kernel unless --fused
        avg_kernel_s = laps_ms / launches / 1e3
        achieved = bpe * n_local / avg_kernel_s / 1e9
        trig = 2 if args.hw_trig else (1 if args.direct_trig else 0)
        static = not (args.no_specialize or args.dh_in_lds)
        table_name = ("Ref4Table" if args.dof == 4 else "Dh7Table") if static else f"RtTable<{args.dof}>"
        variant = "+".join(v for v, on in (("recurrence", trig == 0), ("direct_trig", trig == 1), ("hw_trig", trig == 2),
                                           ("static_table", static), ("runtime_table", not static),
                                           ("dh_in_lds", args.dh_in_lds)) if on)
        workload = f"{n_local} arms/GPU x {world} GPU, {args.dof}-DoF DH chain, K={args.targets} targets, " \
                   f"25 sub-steps, random integer-degree actions drawn in-kernel, episode {L} steps"
        out = {
            "metric": METRIC, "value": n_total * args.steps / elapsed, "unit": "env-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "envs_per_gpu": n_local, "envs_total": n_total, "dof": args.dof,
                       "targets": args.targets, "substeps": 25, "episode_len": L,
                       "collective": (f"{'RCCL' if args.backend == 'nccl' else 'gloo (rehearsal)'} all-gather of returns per episode"
                                      if world > 1 else "none (1 GPU)"),
                       "kernel_variant": variant},
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": load_traffic(f"d{args.dof}_k{args.targets}_n{n_local}"),
                "kernel": (f"rollout_kernel<{table_name}> ({L} steps per launch; us per step quoted)" if args.fused else
                           f"step_kernel<{table_name}, sample=true, trig={trig}, lds={str(args.dh_in_lds).lower()}>"),
                "bytes_per_env_step": bpe, "avg_kernel_us": avg_kernel_s * 1e6,
            },
        }
        if world == 1:
            copy_gbs = measured_copy_bandwidth(torch)
            out["roofline"]["measured_copy_gbs"] = copy_gbs
            out["roofline"]["frac_of_measured_copy"] = achieved / copy_gbs
        if world == 1 and not args.fused and not args.ablate:
            # informational, not the headline: the same 50-step episodes as ONE launch each
            eng.reset_random(args.seed, 0)
            eng.rollout_fused(L, args.seed, 0)
            eng.sync()
            reps = max(1, min(20, args.steps // L))
            eng.timer_start()
            for r in range(reps):
                eng.rollout_fused(L, args.seed, (r + 1) * L)
            ms = eng.timer_stop()
            out["secondary"] = {"fused_rollout": {
                "env_steps_per_s": n_local * L * reps / (ms / 1e3), "us_per_step": ms * 1e3 / (L * reps),
                "steps_per_launch": L, "note": "mt_rollout_fused: state stays in registers/LDS between steps, "
                "bit-identical results; arithmetic-bound, so the per-step byte model does not apply"}}
        if world == 1 and not args.fused and not args.ablate and not args.no_cpu_baseline:
            # informational: BASELINE.json's other single-GPU configurations, same kernel path, short runs
            out["secondary"]["other_configs"] = {}
            for label, n2, tbl, rad in (("configs[1]: 65536 arms, 4-DoF", 65536, m.REF_DH_TABLE, 51.3),
                                        ("configs[4]: 1048576 arms, 7-DoF table", 1048576, m.DH7_TABLE, 92.6)):
                e2 = m.StepEngine(n2, args.targets, dh_table=tbl, radius=rad, device=dev)
                e2.reset_random(args.seed, 0)
                e2.rollout(200, args.seed, 0)
                e2.sync()
                e2.timer_start()
                e2.rollout(600, args.seed, 200)
                us = e2.timer_stop() * 1e3 / 600
                b2 = algorithmic_bytes_per_env_step(len(tbl), args.targets)
                out["secondary"]["other_configs"][label] = {
                    "us_per_step": us, "env_steps_per_s": n2 / (us * 1e-6), "bytes_per_env_step": b2,
                    "frac_of_hbm_peak": b2 * n2 / (us * 1e-6) / 1e9 / HBM_PEAK_GBS}
                e2.close()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(table, args.targets)
        print(json.dumps(out), flush=True)

    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
