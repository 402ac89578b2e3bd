#!/usr/bin/env python3
"""Which step-kernel variant for which batch size?  us per step (HIP events, 50-step episodes with resets) for the
streaming variant, the register-prefetch variant and the 2- / 4-lane split kernels, two passes per cell.
    python tools/variant_sweep.py > gpurun_out/variant_sweep.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.split_variants_check import timing  # noqa: E402

sizes = [int(v) for v in sys.argv[1:]] or [8192, 32768, 49152, 65536, 98304, 131072, 196608, 262144, 393216, 524288, 1048576]
res = {}
for n in sizes:
    row = {}
    for name, split, pf in (("streaming", 0, 0), ("prefetch", 0, 1), ("split2", 2, 0), ("split4", 4, 0)):
        if split == 4 and n > 262144:
            continue
        row[name] = [timing(split, pf, n, steps=400), timing(split, pf, n, steps=400)]
    res[n] = row
    print(n, row, file=sys.stderr, flush=True)
print(json.dumps(res, indent=1))
