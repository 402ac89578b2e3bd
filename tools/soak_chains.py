#!/usr/bin/env python3
"""Stability run of the forked chains: the benchmark's episode loop (rollout, per-chain snapshot gather, per-chain reset; no
host synchronisation for hundreds of episodes) on a 2-chain engine against a 1-chain engine fed the same seeds, compared
bit for bit at checkpoints.  1 048 576 arms; a ragged 700 001-arm batch with 3 chains; and the episode boundary folded into
mt_rollout's launches (131 072 arms: the default; 1 048 576 arms: MT_DEFER_RESET_CHAINS=1) against the eager reset / snapshot.
    python tools/soak_chains.py [episodes]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import manytor_amd as m  # noqa: E402

EPISODES = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
FIELDS = ("F_GOALS", "F_ALIVE", "F_TOTAL_REWARD", "F_POINTS", "F_LAST_RETURN", "F_OBS", "F_REWARD", "F_DONE", "F_EE", "F_DONE_BITS")


def make(n, chains, env=None):
    env = dict(env or {})
    if chains:
        env["MT_CHAINS"] = str(chains)
    os.environ.update(env)
    e = m.StepEngine(n, 7)
    for k in env:
        os.environ.pop(k)
    return e


EAGER = {"MT_DEFER_RESET": "0", "MT_ROLLOUT_SNAP": "0"}     # reset and snapshot as launches of their own
# (n, chains of the engine under test, episode length, reference engine, engine under test)
CASES = ((1048576, 2, 20, lambda n: make(n, 1), lambda n: make(n, 2)),
         (700001, 3, 13, lambda n: make(n, 1), lambda n: make(n, 3)),
         # the episode boundary folded into mt_rollout's multi-step launches (default at this size) against the eager forms
         (131072, 0, 20, lambda n: make(n, 0, EAGER), lambda n: make(n, 0)),
         (1048576, 2, 20, lambda n: make(n, 2, EAGER), lambda n: make(n, 2, {"MT_DEFER_RESET_CHAINS": "1"})))
for n, chains, L, make_a, make_b in CASES:
    a, b = make_a(n), make_b(n)
    bufs = {a: [None, None], b: [None, None]}
    for e in (a, b):
        e.reset_random(7, 0)
    t0 = time.perf_counter()
    checks = 0
    for ep in range(EPISODES):
        for e in (a, b):
            e.rollout(L, 7, ep * L)
            e.gather_wait()
            bufs[e][ep % 2] = e.gather_begin(bufs[e][ep % 2])
            e.reset_random(7, ep + 1)
        if ep % 250 == 249 or ep == EPISODES - 1:
            for e in (a, b):
                e.rollout(3, 7, 5)                       # compare mid-episode state too
            for e in (a, b):
                e.gather_wait(host=True)
            assert np.array_equal(bufs[a][ep % 2].cpu().numpy(), bufs[b][ep % 2].cpu().numpy()), ("gathered", n, ep)
            for f in FIELDS:
                assert np.array_equal(a.get(getattr(m.lib, f)), b.get(getattr(m.lib, f))), (f, n, ep)
            checks += 1
            print(f"... {n} arms, {chains or 'default'} chains: episode {ep + 1} identical", flush=True)
            for e in (a, b):
                e.reset_random(7, ep + 1)
    dt = time.perf_counter() - t0
    print(f"soak_chains ok: {n} arms x {EPISODES} episodes of {L} steps, {b.dispatch()['rollout']} vs {a.dispatch()['rollout']}: "
          f"{checks} checkpoints bit-identical on {len(FIELDS)} fields + the gathered returns ({dt:.0f} s)")
    a.close()
    b.close()
