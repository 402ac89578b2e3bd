#!/usr/bin/env python3
"""Per-(kernel, grid) statistics from a rocprofv3 --kernel-trace CSV.  The driver's bench command also times the
secondary configurations with the SAME kernel template at other batch sizes, so rocprofv3's own per-name --stats mixes
them; grouping by grid size separates the headline launches (grid = n_envs threads).
    python tools/trace_summary.py <kernel_trace.csv> > profiles/rNN_kernel_stats_by_grid.csv"""
import csv
import re
import sys
from collections import defaultdict

acc = defaultdict(list)
meta = {}
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        name = re.sub(r"^void ", "", row["Kernel_Name"])
        if not name.startswith("mt::"):
            continue
        key = (name, int(row["Grid_Size_X"]))
        acc[key].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        meta[key] = (row["VGPR_Count"], row["SGPR_Count"], row["LDS_Block_Size"], row["Scratch_Size"])
w = csv.writer(sys.stdout)
w.writerow(["kernel", "grid_threads", "calls", "avg_us", "min_us", "median_us", "max_us", "vgpr", "sgpr", "lds", "scratch"])
for key in sorted(acc, key=lambda k: -sum(acc[k])):
    v = sorted(acc[key])
    w.writerow([key[0], key[1], len(v), f"{sum(v) / len(v) / 1e3:.3f}", f"{v[0] / 1e3:.3f}", f"{v[len(v) // 2] / 1e3:.3f}",
                f"{v[-1] / 1e3:.3f}", *meta[key]])
