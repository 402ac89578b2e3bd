#!/usr/bin/env python3
"""mt_rollout_fused with one / two / four lanes per env (MT_SPLIT = 0 / 2 / 4) at 32 768 ... 262 144 arms: us per step inside
50-step launches, two passes per cell.  Where mt_create's rollout_split thresholds come from.
    python tools/fused_split_sweep.py > profiles/rNN_fused_split_sweep.json"""
import json, os, sys
sys.path.insert(0, os.getcwd())
from tools.split_variants_check import timing_fused
res = {}
for n in (32768, 65536, 98304, 131072, 196608, 262144):
    res[n] = {s: [timing_fused(s, n), timing_fused(s, n)] for s in (0, 2, 4)}
    print(n, res[n], file=sys.stderr, flush=True)
print(json.dumps(res, indent=1))
