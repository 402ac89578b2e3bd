#!/usr/bin/env python3
"""mt_rollout as 1..4 independent chains of launches (MT_CHAINS), plain launches vs the cached HIP graph (MT_GRAPH):
us per step by HIP events on the handle's stream around 50-step segments (fork and join included), resets untimed.
    python tools/chain_sweep.py [--dh7] [sizes ...] > gpurun_out/chain_sweep.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import manytor_amd as m  # noqa: E402
from tools.split_variants_check import timing  # noqa: E402

argv = [a for a in sys.argv[1:] if not a.startswith("--")]
kw = dict(dh_table=m.DH7_TABLE, radius=92.6) if "--dh7" in sys.argv else {}
sizes = [int(v) for v in argv] or [32768, 65536, 131072, 262144, 524288, 1048576, 4194304]
res = {}
os.environ.pop("MT_SPLIT", None)
os.environ.pop("MT_PREFETCH", None)


STEADY = "--steady" in sys.argv


def cell(n, chains, graph):
    os.environ["MT_CHAINS"] = str(chains)
    os.environ["MT_GRAPH"] = str(graph)
    fn = timing_steady if STEADY else timing_default
    return [fn(n), fn(n)]


def timing_steady(n, episodes=12):
    """Steady state: `episodes` x (reset + 50-step segment) enqueued back to back, one pair of events around the lot, no
    host synchronisation in between (us per step INCLUDING the per-episode reset, the same in every cell)."""
    import time
    e = m.StepEngine(n, 7, **kw)
    t0 = time.perf_counter()
    ep = 0
    while time.perf_counter() - t0 < 0.15:
        e.reset_random(1, ep)
        e.rollout(50, 1, 0)
        ep += 1
        e.sync()
    e.timer_start()
    for r in range(episodes):
        e.reset_random(1, r)
        e.rollout(50, 1, 0)
    ms = e.timer_stop()
    e.close()
    return round(ms * 1e3 / (episodes * 50), 3)


def timing_default(n):
    # the library's own choice of step kernel for the (chain's) batch size is NOT what is swept here: the kernel is the
    # one mt_create picks for the whole batch
    import time
    e = m.StepEngine(n, 7, **kw)
    e.reset_random(1, 0)
    t0 = time.perf_counter()
    ep = 0
    while time.perf_counter() - t0 < 0.15:
        e.rollout(50, 1, 0)
        ep += 1
        e.reset_random(1, ep)
        e.sync()
    tot = 0.0
    reps = 8 if n <= (1 << 20) else 4
    for r in range(reps):
        e.reset_random(1, r)
        e.sync()
        e.timer_start()
        e.rollout(50, 1, 0)
        tot += e.timer_stop()
    e.close()
    return round(tot * 1e3 / (reps * 50), 3)


for n in sizes:
    row = {}
    for chains in (1, 2, 3, 4):
        for graph in (0, 1):
            row[f"chains={chains} graph={graph}"] = cell(n, chains, graph)
    res[n] = row
    print(n, row, file=sys.stderr, flush=True)
print(json.dumps(res, indent=1))
