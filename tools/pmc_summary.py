#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc CSVs written by tools/pmc_step.sh: per-kernel average of every counter."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
summary = {k: {c: {"avg": sum(v) / len(v), "n": len(v)} for c, v in cs.items()} for k, cs in acc.items()}
with open(os.path.join(out, "pmc_summary.json"), "w") as f:
    json.dump(summary, f, indent=1, sort_keys=True)
for k, cs in summary.items():
    if "step_kernel" in k or "rollout" in k or "stream_probe" in k:
        print(k)
        for c, v in sorted(cs.items()):
            print(f"   {c:24s} {v['avg']:.6g}  (n={v['n']})")
