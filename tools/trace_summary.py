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

# ---- device time per STEP when a step is several concurrent launches (mt_rollout's chains) ---------------------------
# usage: trace_summary.py <kernel_trace.csv> --union <kernel substring> <grid threads> <launches per step>
# The launches of one step overlap on different queues, so the per-launch average above says nothing about the step's
# device time; the union of their [start, end] intervals does.  Printed as one JSON object on stderr-free stdout tail.
if "--union" in sys.argv:
    import json
    i = sys.argv.index("--union")
    sub, grid, per_step = sys.argv[i + 1], int(sys.argv[i + 2]), int(sys.argv[i + 3])
    iv = []
    with open(sys.argv[1]) as f:
        for row in csv.DictReader(f):
            if sub in row["Kernel_Name"] and int(row["Grid_Size_X"]) == grid:
                iv.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"])))
    iv.sort()
    union = overlap = 0
    cur_s, cur_e = iv[0]
    for s_, e_ in iv[1:]:
        if s_ <= cur_e:
            overlap += min(e_, cur_e) - s_
            cur_e = max(cur_e, e_)
        else:
            union += cur_e - cur_s
            cur_s, cur_e = s_, e_
    union += cur_e - cur_s
    total = sum(e_ - s_ for s_, e_ in iv)
    print("# " + json.dumps({"kernel": sub, "grid_threads": grid, "launches": len(iv), "launches_per_step": per_step,
                             "avg_launch_us": total / len(iv) / 1e3, "union_us_per_step": union / (len(iv) / per_step) / 1e3,
                             "sum_of_launches_us_per_step": total / (len(iv) / per_step) / 1e3,
                             "concurrency": total / union}))
