#!/usr/bin/env python3
"""Stability run of the forked chains: the benchmark's episode loop (rollout, per-chain snapshot gather, per-chain reset; no
host synchronisation for hundreds of episodes) on a 2-chain engine against a 1-chain engine fed the same seeds, compared
bit for bit at checkpoints.  1 048 576 arms; also a ragged 700 001-arm batch with 3 chains.
    python tools/soak_chains.py [episodes]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import manytor_amd as m  # noqa: E402

EPISODES = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
FIELDS = ("F_GOALS", "F_ALIVE", "F_TOTAL_REWARD", "F_POINTS", "F_LAST_RETURN", "F_OBS", "F_REWARD", "F_DONE", "F_EE", "F_DONE_BITS")


def make(n, chains):
    os.environ["MT_CHAINS"] = str(chains)
    e = m.StepEngine(n, 7)
    os.environ.pop("MT_CHAINS")
    return e


for n, chains, L in ((1048576, 2, 20), (700001, 3, 13)):
    a, b = make(n, 1), make(n, chains)
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
            print(f"... {n} arms, {chains} chains: episode {ep + 1} identical", flush=True)
            for e in (a, b):
                e.reset_random(7, ep + 1)
    dt = time.perf_counter() - t0
    print(f"soak_chains ok: {n} arms x {EPISODES} episodes of {L} steps, {chains} chains vs 1: {checks} checkpoints bit-identical on "
          f"{len(FIELDS)} fields + the gathered returns ({dt:.0f} s)")
    a.close()
    b.close()
