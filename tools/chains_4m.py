#!/usr/bin/env python3
"""4 194 304 arms (one launch per step by default): one chain against two chains with / without an occupancy cap.  The
arena's placement changes the figure by ~10 % from one allocation to the next, so every configuration is measured on
`reps` fresh engines, interleaved, and the medians are compared.   python tools/chains_4m.py [reps]"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import manytor_amd as m  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
configs = [("1 chain", {"MT_CHAINS": "1"}), ("2 chains", {"MT_CHAINS": "2", "MT_BLOCKS_PER_CU": "0"}),
           ("2 chains, 5 blocks/CU", {"MT_CHAINS": "2", "MT_BLOCKS_PER_CU": "5"}),
           ("2 chains, 4 blocks/CU", {"MT_CHAINS": "2", "MT_BLOCKS_PER_CU": "4"})]
res = {name: [] for name, _ in configs}
for rep in range(reps):
    for name, env in configs:
        for k in ("MT_CHAINS", "MT_BLOCKS_PER_CU"):
            os.environ.pop(k, None)
        os.environ.update(env)
        e = m.StepEngine(4194304, 7)
        e.reset_random(1, 0)
        for _ in range(3):
            e.rollout(50, 1, 0)
        e.sync()
        e.lap_times()
        for ep in range(4):
            e.reset_random(1, ep + 1)
            e.lap_begin(); e.rollout(50, 1, 0); e.lap_end()
        e.sync()
        res[name].append(sum(e.lap_times()) * 1e3 / 200)
        e.close()
    print(rep, {k: round(v[-1], 1) for k, v in res.items()}, flush=True)
for name, v in res.items():
    print(f"{name:24s} median {statistics.median(v):7.1f} us per step   min {min(v):7.1f}   max {max(v):7.1f}")
