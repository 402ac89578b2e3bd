#!/usr/bin/env python3
"""Which step-kernel variant for which batch size?  us per step (HIP events, 50-step episodes with resets) for the
streaming variant, the register-prefetch variant and the 2- / 4-lane split kernels, each without and with the
whole-degree sin / cos table (TT), two passes per cell.
    python tools/variant_sweep.py [--dh7] [sizes ...] > gpurun_out/variant_sweep.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import manytor_amd as m  # noqa: E402
from tools.split_variants_check import timing  # noqa: E402

argv = [a for a in sys.argv[1:] if not a.startswith("--")]
kw = dict(dh_table=m.DH7_TABLE, radius=92.6) if "--dh7" in sys.argv else {}
sizes = [int(v) for v in argv] or [8192, 32768, 49152, 65536, 98304, 131072, 196608, 262144, 393216, 524288, 1048576]
res = {}
for n in sizes:
    row = {}
    for name, split, pf in (("streaming", 0, 0), ("prefetch", 0, 1), ("split2", 2, 0), ("split4", 4, 0)):
        if split == 4 and n > 262144 or split == 2 and n > 524288:
            continue
        for tt in (0, 1):
            row[name + ("+table" if tt else "")] = [timing(split, pf, n, steps=400, trig_table=tt, **kw) for _ in range(2)]
    res[n] = row
    print(n, row, file=sys.stderr, flush=True)
print(json.dumps(res, indent=1))
