#!/usr/bin/env python3
"""A small shard under bench.py, from a rocprofv3 --kernel-trace CSV: the rollout launches by duration (the episode's first
launch carries the reset as its prologue, the last one writes the gather's snapshot: same kernel name, different durations),
and the timeline of two regions' worth of kernels from the middle of the run (start / end relative to the first of them, gap
to the previous kernel on any queue, queue, grid, kernel).
    python tools/shard_timeline.py <kernel_trace.csv> [rows] [position in the run, 0 .. 1, or -N = N rows before the last step launch]"""
import csv
import statistics
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mt::", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name[:64], int(r["Grid_Size_X"]), r.get("Queue_Id", "?")))
rows.sort()
roll = [r for r in rows if "rollout" in r[2]]
if roll:
    grid = statistics.mode(r[3] for r in roll)
    d = sorted((r[1] - r[0]) / 1e3 for r in roll if r[3] == grid)
    half = d[len(d) // 2]
    plain = [x for x in d if x <= half * 1.12]
    heavy = [x for x in d if x > half * 1.12]
    print(f"# {roll[0][2]} grid {grid}: {len(d)} launches; median {half:.2f} us; <= 1.12 x median: {len(plain)} launches, mean "
          f"{statistics.mean(plain):.2f}; longer: {len(heavy)} launches, mean {statistics.mean(heavy) if heavy else 0:.2f} "
          f"(the episodes' first launches: reset prologue)")
count = int(sys.argv[2]) if len(sys.argv) > 2 else 28
where = float(sys.argv[3]) if len(sys.argv) > 3 else 0.6   # position in the run: 0.6 = the pre-warm loop (unfenced), 0.9 = timed regions
if where < 0:     # counted back from the last rollout / step launch of the run: the timed regions
    last = max(i for i, r in enumerate(rows) if "rollout" in r[2] or "step_kernel" in r[2])
    j = max(1, last + int(where))
else:
    j = int(len(rows) * where)
base, prev_end = rows[j][0], rows[j - 1][1]
last_on_queue = {}
for r in rows[j:j + count]:
    own = (r[0] - last_on_queue[r[4]]) / 1e3 if r[4] in last_on_queue else float("nan")
    print(f"{(r[0] - base) / 1e3:9.1f} {(r[1] - base) / 1e3:9.1f} dur {(r[1] - r[0]) / 1e3:7.1f} gap {(r[0] - prev_end) / 1e3:7.1f} "
          f"(own queue {own:6.1f}) q{r[4]} grid {r[3]:8d} {r[2]}")
    prev_end = max(prev_end, r[1])
    last_on_queue[r[4]] = r[1]
