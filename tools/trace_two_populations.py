#!/usr/bin/env python3
"""Kernel-trace view of ONE kernel's dispatches split into populations by time (bench.py's two passes over a secondary
configuration = two arenas): per population the launch count, mean / median duration, the union of the busy intervals per
launch pair (two chains), the queues used, and the mean gap between consecutive launches on the same queue.
    python tools/trace_two_populations.py <kernel_trace.csv> <kernel substring> [grid size]"""
import csv
import statistics as st
import sys

path, needle = sys.argv[1], sys.argv[2]
grid = int(sys.argv[3]) if len(sys.argv) > 3 else None
rows = []
for r in csv.DictReader(open(path)):
    if needle in r["Kernel_Name"] and (grid is None or int(r["Grid_Size_X"]) == grid):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")))
rows.sort()
pops, cur = [], [rows[0]]
for a, b in zip(rows, rows[1:]):
    if b[0] - a[1] > 5_000_000:          # > 5 ms without this kernel: another configuration ran in between
        pops.append(cur)
        cur = []
    cur.append(b)
pops.append(cur)
print(f"{needle}: {len(rows)} dispatches, {len(pops)} populations")
for i, p in enumerate(pops):
    p = p[len(p) // 4:]                                     # skip the warm-up quarter
    dur = [(e - s) / 1e3 for s, e, _ in p]
    queues = sorted({q for _, _, q in p})
    busy, last_end = 0, 0
    for s, e, _ in p:
        s2 = max(s, last_end)
        if e > s2:
            busy += e - s2
            last_end = e
    span = (p[-1][1] - p[0][0]) / 1e3
    gaps = []
    by_q = {}
    for s, e, q in p:
        if q in by_q:
            gaps.append((s - by_q[q]) / 1e3)
        by_q[q] = e
    print(f"  pop {i}: {len(p)} launches on queues {queues}; duration mean {st.mean(dur):7.2f} median {st.median(dur):7.2f} us; "
          f"span {span:9.1f} us, union busy {busy / 1e3:9.1f} us ({busy / 1e3 / span:.2f} of the span); "
          f"mean gap on a queue {st.mean(gaps):6.2f} us; span per launch {span / len(p):6.2f} us")
