#!/usr/bin/env python3
"""Per-dispatch PMC counters of ONE kernel split into populations by dispatch order: bench.py times every secondary
configuration on two fresh engines (list forward, then reversed), so the dispatches of e.g. the 7-joint step kernel come in
two runs of consecutive dispatch ids -- one per arena.  Prints the mean of every counter per population: what differs between
a slow arena and a fast one of the same kernel on the same inputs?
    python tools/pmc_two_populations.py <dir with *counter_collection.csv> <kernel substring> [grid size]"""
import csv
import glob
import os
import sys
from collections import defaultdict

root, needle = sys.argv[1], sys.argv[2]
grid = int(sys.argv[3]) if len(sys.argv) > 3 else None
rows = defaultdict(dict)            # dispatch id -> counter -> value
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            if needle not in r["Kernel_Name"]:
                continue
            if grid is not None and int(r["Grid_Size"]) != grid:
                continue
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows)
if not ids:
    sys.exit(f"no dispatch of a kernel matching {needle!r}")
# populations = runs of dispatch ids without a large hole (other kernels of other configurations in between)
pops, cur = [], [ids[0]]
for a, b in zip(ids, ids[1:]):
    if b - a > 2000:
        pops.append(cur)
        cur = []
    cur.append(b)
pops.append(cur)
counters = sorted({c for d in rows.values() for c in d})
print(f"{needle}: {len(ids)} dispatches in {len(pops)} populations {[len(p) for p in pops]}")
print("counter".ljust(36) + "".join(f"pop {i} (ids {p[0]}..{p[-1]})".rjust(34) for i, p in enumerate(pops)))
for c in counters:
    line = c.ljust(36)
    for p in pops:
        vals = [rows[i][c] for i in p[len(p) // 4:] if c in rows[i]]       # skip each population's warm-up quarter
        line += (f"{sum(vals) / len(vals):.6g}" if vals else "-").rjust(34)
    print(line)
