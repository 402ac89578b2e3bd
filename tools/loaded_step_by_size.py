#!/usr/bin/env python3
"""The policy-in-the-loop step (mt_step with the actions read from HBM) by batch size: us per step = (sample, step) pairs
minus sample launches alone, as bench.time_loaded_action_steps measures it, for the default dispatch and for one chain.
    python tools/loaded_step_by_size.py [sizes ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import manytor_amd as m  # noqa: E402

sizes = [int(v) for v in sys.argv[1:]] or [32768, 65536, 131072, 163840, 262144, 524288, 1048576]
out = {}
for n in sizes:
    row = {}
    for label, env in (("default", {}), ("one_chain", {"MT_CHAINS": "1"}), ("two_chains", {"MT_CHAINS": "2"})):
        keep = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            us, us_sample, us_pair, name = bench.time_loaded_action_steps(m, n, m.REF_DH_TABLE, 51.3, 7, 0, 0x5EED)
        finally:
            for k, v in keep.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        row[label] = {"us_per_step": round(us, 2), "sample_us": round(us_sample, 2), "pair_us": round(us_pair, 2), "kernel": name}
    out[n] = row
    print(n, {k: (v["us_per_step"], v["pair_us"]) for k, v in row.items()}, file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
