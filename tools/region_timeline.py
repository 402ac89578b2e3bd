#!/usr/bin/env python3
"""Timeline of the kernels around an episode boundary of bench.py's headline run, from a rocprofv3 --kernel-trace CSV:
start / end (us, relative to a reset of the 1 M-arm batch in the middle of the run), duration, queue, grid, kernel.
    python tools/region_timeline.py <kernel_trace.csv> [rows] [min grid of the reset to centre on, default 262144]"""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("(")[0][-44:], int(r["Grid_Size_X"]), r.get("Queue_Id", "?")))
rows.sort()
min_grid = int(sys.argv[3]) if len(sys.argv) > 3 else 262144
resets = [i for i, r in enumerate(rows) if "reset" in r[2] and "kernel" in r[2] and r[3] >= min_grid]
j = resets[len(resets) // 2]
base = rows[j][0]
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
for r in rows[max(0, j - 8):j + count]:
    print(f"{(r[0] - base) / 1e3:9.1f} {(r[1] - base) / 1e3:9.1f} dur {(r[1] - r[0]) / 1e3:7.1f} q{r[4]} grid {r[3]:8d} {r[2]}")
