#!/usr/bin/env python3
"""Does a HIP graph shorten the launch-per-step path where it is launch-bound (small shards)?

A 50-step episode segment: (a) 50 plain step_random launches (MT_GRAPH=0), (b) mt_rollout's own cached HIP graph of
those launches, whose nodes take their step index from a device word so that one graph serves every segment
(MT_GRAPH=1; the default up to 131 072 arms), (c) the plain launches captured by torch into a graph with the step
indices frozen (a lower bound for any graph: no indirection), (d) mt_rollout_fused.  us per step, wall clock over 40
episodes after a warm-up; every episode starts with mt_reset_random (its time is included, the same in every column).

    python tools/graph_replay.py > profiles/rNN_graph_replay.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import manytor_amd as m  # noqa: E402

T, EPISODES = 50, 40


def wall_per_step(fn, sync):
    for _ in range(5):
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(EPISODES):
        fn()
    sync()
    return (time.perf_counter() - t0) / (EPISODES * T) * 1e6


out = {}
for n in (1024, 8192, 32768, 65536, 131072, 262144, 1048576):
    row = {}
    for label, mode in (("plain_launches", "0"), ("mt_rollout_internal_graph", "1")):
        os.environ["MT_GRAPH"] = mode
        eng = m.StepEngine(n, 7)
        eng.reset_random(1, 0)
        step = [0]

        def episode():
            eng.reset_random(1, step[0] // T)   # every variant steps freshly reset envs: comparable work
            eng.rollout(T, 1, step[0])          # the step index moves on, as in a real run: the graph is reused anyway
            step[0] += T
        row[label] = wall_per_step(episode, eng.sync)
        if mode == "0":
            def fused_episode():
                eng.reset_random(1, 0)
                eng.rollout_fused(T, 1, 0)
            row["fused"] = wall_per_step(fused_episode, eng.sync)
            side = torch.cuda.Stream()
            torch.cuda.synchronize()
            with torch.cuda.stream(side):
                eng.use_torch_stream()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    eng.rollout(T, 1, 0)
                torch.cuda.synchronize()

                def replay_episode():
                    eng.reset_random(1, 0)
                    graph.replay()
                row["torch_captured_replay_fixed_step_index"] = wall_per_step(replay_episode, torch.cuda.synchronize)
            eng.set_stream(None)
        eng.close()
    out[str(n)] = {k: round(v, 3) for k, v in row.items()}
    print(n, out[str(n)], file=sys.stderr)
print(json.dumps(out, indent=1))
