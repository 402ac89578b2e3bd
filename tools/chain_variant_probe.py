#!/usr/bin/env python3
"""Around 131 072 arms: one launch per step against two / three chains with the cached graph or plain launches, with the
two-lane kernels the chain's env count would pick or one env per lane (MT_SPLIT=0), steady-state protocol (12 x (reset + 50
steps) back to back, one event pair).  Shows the block-count quantisation: 131 072 envs = exactly two 256-thread blocks per CU
is the best case of a single launch; 98 304 and 163 840 envs gain from two chains.
    python tools/chain_variant_probe.py"""
import os, sys, time, json
sys.path.insert(0, os.getcwd())
import manytor_amd as m
def steady(n, env, episodes=12):
    for k in ("MT_CHAINS","MT_GRAPH","MT_SPLIT","MT_PREFETCH"): os.environ.pop(k, None)
    os.environ.update(env)
    e = m.StepEngine(n, 7)
    t0 = time.perf_counter(); ep = 0
    while time.perf_counter() - t0 < 0.15:
        e.reset_random(1, ep); e.rollout(50, 1, 0); ep += 1; e.sync()
    e.timer_start()
    for r in range(episodes):
        e.reset_random(1, r); e.rollout(50, 1, 0)
    ms = e.timer_stop(); name = e.step_kernel_name(); e.close()
    return round(ms * 1e3 / (episodes * 50), 3), name
import sys as _sys
sizes = [int(v) for v in _sys.argv[1:]] or [98304, 131072, 163840]
for n in sizes:
    for env in ({}, {"MT_CHAINS": "1"}, {"MT_CHAINS": "2", "MT_GRAPH": "0"}, {"MT_CHAINS": "2", "MT_GRAPH": "0", "MT_SPLIT": "0"},
                {"MT_CHAINS": "2", "MT_GRAPH": "1"}, {"MT_CHAINS": "2", "MT_GRAPH": "1", "MT_SPLIT": "0"},
                {"MT_CHAINS": "3", "MT_GRAPH": "0"}, {"MT_CHAINS": "3", "MT_GRAPH": "0", "MT_SPLIT": "0"}):
        a = steady(n, env)
        b = steady(n, env)
        print(n, env, a[0], b[0], a[1][-60:], flush=True)
