#!/usr/bin/env python3
"""mt_rollout on small shards: us per step for k steps per launch (MT_ROLLOUT_K), one or two chains, with and without the
cached HIP graph of the launch-per-step form -- the sweep behind kPolicy.multi_step_* (engine.hip).

Two ways of timing, both HIP events on the engine's streams, 50-step episodes with a reset between them:
  idle : every 50-step (and 20-step) segment starts on an idle device (sync, timer_start, rollout, timer_stop): what a
         fenced benchmark region or a learner that waits for every segment sees
  b2b  : episodes queued back to back, laps around the rollout calls only (no host wait in between)

    python tools/rollout_k_sweep.py [sizes ...] > gpurun_out/r04_rollout_k_sweep.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import manytor_amd as m  # noqa: E402


def measure(n, env, episodes=12):
    keep = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        e = m.StepEngine(n, 7)
    finally:
        for k, v in keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    d = e.dispatch()
    e.reset_random(1, 0)
    t0 = time.perf_counter()
    ep = 0
    while time.perf_counter() - t0 < 0.15:
        e.rollout(50, 1, 0)
        ep += 1
        e.reset_random(1, ep)
        e.sync()
    out = {"form": d["rollout"]["form"], "k": d["rollout"]["steps_per_launch"], "chains": d["chains"]["count"],
           "lanes": d["rollout"]["lanes_per_env"]}
    for T in (50, 20):
        tot = []
        for r in range(episodes):
            e.reset_random(1, r)
            e.sync()
            e.timer_start()
            e.rollout(T, 1, 0)
            tot.append(e.timer_stop() * 1e3 / T)
        tot.sort()
        out[f"idle_T{T}"] = round(tot[len(tot) // 2], 3)
    e.sync()
    e.lap_times()
    for r in range(episodes):
        e.reset_random(1, r)
        e.lap_begin()
        e.rollout(50, 1, 0)
        e.lap_end()
    laps = sorted(e.lap_times())
    out["b2b_T50"] = round(laps[len(laps) // 2] * 1e3 / 50, 3)
    e.close()
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    ks = [int(v) for a in sys.argv[1:] if a.startswith("--k=") for v in a[4:].split(",")] or [1, 2, 3, 4, 5, 8, 10, 25, 50]
    sizes = [int(v) for v in args] or [32768, 65536, 98304, 131072, 163840, 196608, 262144, 393216, 524288]
    res = {}
    for n in sizes:
        row = {}
        for k in ks:
            for chains in (1, 2):
                if chains == 2 and n < 65536:
                    continue
                row[f"k{k}_c{chains}"] = measure(n, {"MT_ROLLOUT_K": str(k), "MT_CHAINS": str(chains)})
                if k > 1 and "--early-ab" in sys.argv:      # the rollout kernels' plain prologue (RPF = 0) beside the default
                    row[f"k{k}_c{chains}_plain_prologue"] = measure(n, {"MT_ROLLOUT_K": str(k), "MT_CHAINS": str(chains),
                                                                     "MT_ROLLOUT_EARLY": "0"})
        row["k1_c1_nograph"] = measure(n, {"MT_ROLLOUT_K": "1", "MT_CHAINS": "1", "MT_GRAPH": "0"})
        row["k1_c2_graph"] = measure(n, {"MT_ROLLOUT_K": "1", "MT_CHAINS": "2", "MT_GRAPH": "1"}) if n >= 65536 else None
        row["default"] = measure(n, {})
        res[n] = row
        print(n, json.dumps(row), file=sys.stderr, flush=True)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
