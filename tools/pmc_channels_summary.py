#!/usr/bin/env python3
"""Summarise tools/pmc_channels.sh: per allocation of the 4 194 304-arm engine (dispatch order = allocation order, 50
step launches each) and per counter, the sum over the 128 TCC instances (16 channels x 8 XCDs) and how unevenly the
instances are loaded (max / mean, min / mean), next to the HIP-event step time of that allocation.
    python tools/pmc_channels_summary.py <outdir> > profiles/rNN_placement_channels.json"""
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
res = {"what": "4 194 304 arms, reference arm, one chain (MT_CHAINS=1); 6 fresh arenas in one process; rocprofv3 --pmc with the "
               "raw per-instance TCC counters (DIMENSION_INSTANCE[0:15] x DIMENSION_XCC[0:7]); timed launches = the last 20 "
               "of the 50 of every allocation",
       "passes": {}}
for name in ("rd", "wr", "req"):
    files = glob.glob(os.path.join(out, name, "**", "*_results.json"), recursive=True)
    if not files:
        continue
    r = json.load(open(files[0]))["rocprofiler-sdk-tool"][0]
    cname = {c["id"]["handle"]: c["name"] for c in r["counters"]}
    times = json.load(open(os.path.join(out, name + ".times.json")))["us_per_step_by_allocation"]
    disp = [d for d in r["callback_records"]["counter_collection"]
            if d["dispatch_data"]["dispatch_info"]["grid_size"]["x"] == 4194304
            and d["dispatch_data"]["dispatch_info"]["group_segment_size"] > 0]          # the step kernel (LDS table), not the reset
    disp.sort(key=lambda d: d["dispatch_data"]["dispatch_info"]["dispatch_id"])
    per_alloc = len(disp) // len(times)
    allocs = []
    for k, t in enumerate(times):
        acc = defaultdict(list)
        for d in disp[k * per_alloc + 30:(k + 1) * per_alloc]:            # the 20 timed launches
            by = defaultdict(list)
            for rec in d["records"]:
                by[cname[rec["counter_id"]["handle"]]].append(rec["value"])
            for c, v in by.items():
                mean = sum(v) / len(v)
                acc[c].append((sum(v), max(v) / mean if mean else 0.0, min(v) / mean if mean else 0.0, len(v)))
        row = {"us_per_step_under_pmc": t}
        for c, v in acc.items():
            row[c] = {"sum_per_launch": sum(x[0] for x in v) / len(v), "max_over_mean_instance": max(x[1] for x in v),
                      "min_over_mean_instance": min(x[2] for x in v), "instances": v[0][3]}
        allocs.append(row)
    res["passes"][name] = allocs
plain = os.path.join(out, "plain.times.json")
if os.path.exists(plain):
    res["us_per_step_by_allocation_without_profiler"] = json.load(open(plain))["us_per_step_by_allocation"]
print(json.dumps(res, indent=1))
