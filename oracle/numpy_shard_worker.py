"""One worker of bench.py's `numpy x multiprocessing` CPU-baseline leg -- TEST / BASELINE INFRASTRUCTURE ONLY.

SURVEY.md 8(d)(iii): the reference's own kind of CPU parallelism is numpy in one process per core.  bench.py starts one
of these per usable core (plain subprocesses: nothing of the parent's HIP state is shared, no pickling of work items);
each steps the vectorised numpy restatement (oracle/manytor_oracle.py BatchOracle, fp64, manytor.py:255-260 semantics) on
its own shard of envs for about `budget` seconds and prints one JSON line {"env_steps": ..., "seconds": ...}.

    python -m oracle.numpy_shard_worker <worker index> <envs per worker> <dof: 4|7> <targets> <budget seconds>
"""
import json
import sys
import time

import numpy as np


def main():
    w, n, dof, k, budget = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5])
    from oracle import manytor_oracle as mo
    from oracle import philox_ref as px
    if dof == 4:
        table, radius = mo.REF_DH_TABLE, 51.3
    else:
        import math
        table = np.array([(0.0, -math.pi / 2, 34.0, 0.0), (0.0, math.pi / 2, 0.0, 0.0), (4.5, math.pi / 2, 40.0, 0.0),
                          (-4.5, -math.pi / 2, 0.0, 0.0), (0.0, -math.pi / 2, 40.0, 0.0), (8.8, math.pi / 2, 0.0, -math.pi / 2),
                          (0.0, 0.0, 12.6, 0.0)])
        radius = 92.6
    ids = np.arange(w * n, (w + 1) * n, dtype=np.uint64)
    ora = mo.BatchOracle(n, k, table=table, radius=radius)
    ora.reset(px.sample_targets(0x5EED, ids, 0, k, radius).astype(np.float64))
    acts = [px.sample_actions(0x5EED, ids, t, dof).astype(np.float64) for t in range(8)]
    ora.step(acts[0])
    print("ready", flush=True)                 # start-up (imports, target draw) is over
    sys.stdin.readline()                       # the parent releases all workers together
    t0 = time.perf_counter()
    steps = 0
    while steps < 1 or time.perf_counter() - t0 < budget:
        ora.step(acts[steps % 8])
        steps += 1
    print(json.dumps({"env_steps": n * steps, "seconds": time.perf_counter() - t0}), flush=True)


if __name__ == "__main__":
    main()
