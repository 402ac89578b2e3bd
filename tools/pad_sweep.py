#!/usr/bin/env python3
"""Row padding (MT_LD_PAD, floats) vs step time at batch sizes whose rows are a power of two long."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.split_variants_check import timing  # noqa: E402

res = {}
for n in (131072, 262144, 524288, 1048576, 2097152):
    row = {}
    for pad in (0, 64, 192, 256, 320, 1024, 1088, 1280, 2112, 4160):
        os.environ["MT_LD_PAD"] = str(pad)
        row[pad] = [timing(0, 0, n, steps=300), timing(0, 0, n, steps=300)]
    res[n] = row
    print(n, row, file=sys.stderr, flush=True)
os.environ.pop("MT_LD_PAD", None)
print(json.dumps(res, indent=1))
