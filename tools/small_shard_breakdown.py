#!/usr/bin/env python3
"""Where do the ~7 us of a 65 536 / 131 072-arm step launch go?  Full kernel vs the diagnostic builds (memory only,
arithmetic only, no interior sub-steps) vs the fused rollout, same process, HIP events around 600 launches each.
    python tools/small_shard_breakdown.py > gpurun_out/small_shard.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import manytor_amd as m  # noqa: E402


def us_per_step(n, steps=600, fused=False, prefetch=None, dof7=False, **kw):
    if prefetch is None:
        os.environ.pop("MT_PREFETCH", None)
    else:
        os.environ["MT_PREFETCH"] = str(int(prefetch))
    if dof7:
        kw.update(dh_table=m.DH7_TABLE, radius=92.6)
    e = m.StepEngine(n, 7, **kw)
    e.reset_random(1, 0)
    t0 = time.perf_counter()
    ep = 0
    while time.perf_counter() - t0 < 0.15:
        e.rollout(50, 1, 0)
        ep += 1
        e.reset_random(1, ep)
        e.sync()
    tot = 0.0
    for r in range(steps // 50):
        e.reset_random(1, r)
        e.sync()
        e.timer_start()
        if fused:
            e.rollout_fused(50, 1, 0)
        else:
            e.rollout(50, 1, 0)
        tot += e.timer_stop()
    e.close()
    return round(tot * 1e3 / (steps // 50 * 50), 3)


def main():
    out = {}
    for n in (16384, 65536, 131072, 262144, 524288, 1048576, 4194304):
        out[n] = {
            "streaming_variant(MT_PREFETCH=0)": us_per_step(n, prefetch=0),
            "latency_variant_registers(MT_PREFETCH=1)": us_per_step(n, prefetch=1),
            "7dof_streaming": us_per_step(n, prefetch=0, dof7=True),
            "7dof_latency_registers": us_per_step(n, prefetch=1, dof7=True),
            "full": us_per_step(n),
            "fused_rollout_per_step": us_per_step(n, fused=True),
            "runtime_table_streaming": us_per_step(n, specialize=False, prefetch=0),
            "runtime_table_latency": us_per_step(n, specialize=False, prefetch=1),
        }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
