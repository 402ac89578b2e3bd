#!/usr/bin/env python3
"""step_split_kernel (one env over 2 / 4 lanes, MT_SPLIT=2|4) against the one-env-per-lane kernels: bit-identical
fields over several steps for static / runtime tables, K = 1 .. 32, ragged sizes, both action sources; then timings."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import manytor_amd as m  # noqa: E402

FIELDS = ("F_GOALS", "F_ALIVE", "F_TOTAL_REWARD", "F_POINTS", "F_OBS", "F_REWARD", "F_DONE", "F_EE", "F_DONE_BITS")


def run(split, n, k, steps, **kw):
    os.environ["MT_SPLIT"] = str(split)
    os.environ["MT_PREFETCH"] = "0"
    e = m.StepEngine(n, k, pickup_tol=20.0, **kw)
    e.reset_random(3, 0)
    e.rollout(steps, 3, 0)
    acts = np.random.RandomState(1).randint(-180, 180, size=(n, e.dof)).astype(np.float32)
    acts[n // 2, 0] = np.nan
    e.step(acts)
    out = {f: e.get(getattr(m.lib, f)) for f in FIELDS}
    out["bad"] = e.bad_action_count()
    e.close()
    return out


def run_fused(split, n, k, auto_reset, **kw):
    os.environ["MT_SPLIT"] = str(split)
    os.environ["MT_PREFETCH"] = "0"
    e = m.StepEngine(n, k, pickup_tol=25.0, **kw)
    e.reset_random(9, 0)
    e.rollout_fused(3, 9, 0, auto_reset=auto_reset)
    e.rollout_fused(27, 9, 3, auto_reset=auto_reset)
    out = {f: e.get(getattr(m.lib, f)) for f in FIELDS + ("F_EPISODES", "F_LAST_RETURN", "F_RETURN_RING")}
    e.close()
    return out


def timing_fused(split, n, steps=600):
    os.environ["MT_SPLIT"] = str(split)
    e = m.StepEngine(n, 7)
    e.reset_random(1, 0)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:
        e.rollout_fused(50, 1, 0)
        e.reset_random(1, 1)
        e.sync()
    tot = 0.0
    for r in range(steps // 50):
        e.reset_random(1, r)
        e.sync()
        e.timer_start()
        e.rollout_fused(50, 1, 0)
        tot += e.timer_stop()
    e.close()
    return round(tot * 1e3 / (steps // 50 * 50), 3)


def timing(split, prefetch, n, steps=600, trig_table=None, **kw):
    os.environ["MT_SPLIT"] = str(split)
    os.environ["MT_PREFETCH"] = str(prefetch)
    if trig_table is None:
        os.environ.pop("MT_TRIG_TABLE", None)
    else:
        os.environ["MT_TRIG_TABLE"] = str(int(trig_table))
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
        e.rollout(50, 1, 0)
        tot += e.timer_stop()
    e.close()
    return round(tot * 1e3 / (steps // 50 * 50), 3)


def main():
    rng = np.random.RandomState(2)
    rt5 = np.column_stack([rng.uniform(0, 9, 5), rng.choice([-np.pi / 2, 0.3, np.pi / 2], 5), rng.uniform(2, 12, 5), np.zeros(5)])
    cases = [(dict(), 100003, 7, 6), (dict(), 31, 1, 3), (dict(), 65, 32, 4), (dict(specialize=False), 5000, 10, 5),
             (dict(dh_table=m.DH7_TABLE, radius=92.6), 70001, 7, 5), (dict(dh_table=rt5, radius=40.0), 1234, 3, 5),
             (dict(substeps=2), 999, 5, 4), (dict(substeps=8), 999, 5, 4), (dict(terminate_on_ground=True), 4097, 7, 5)]
    for kw, n, k, steps in cases:
        ref = run(0, n, k, steps, **kw)
        for split in (2, 4):
            got = run(split, n, k, steps, **kw)
            for f in ref:
                assert np.array_equal(ref[f], got[f]), (kw, n, k, split, f)
    print("split kernels bit-identical to the one-env-per-lane kernels on", len(cases), "configurations", file=sys.stderr)
    fused_cases = [(dict(), 20000, 2), (dict(), 777, 32), (dict(dh_table=m.DH7_TABLE, radius=92.6), 3001, 3),
                   (dict(specialize=False), 1000, 5), (dict(dh_table=rt5, radius=40.0), 513, 1)]
    for kw, n, k in fused_cases:
        for auto in (False, True):
            ref = run_fused(0, n, k, auto, **kw)
            for split in (2, 4):
                got = run_fused(split, n, k, auto, **kw)
                for f in ref:
                    assert np.array_equal(ref[f], got[f]), (kw, n, k, auto, split, f)
    print("split rollout kernels bit-identical to rollout_kernel on", len(fused_cases), "configurations x auto-reset off/on",
          file=sys.stderr)
    res = {}
    for n in (1024, 16384, 32768, 65536, 131072, 262144):
        res[n] = {"one_lane_streaming": timing(0, 0, n), "one_lane_prefetch": timing(0, 1, n), "split2": timing(2, 0, n),
                  "split4": timing(4, 0, n)}
    res["fused_us_per_step"] = {n: {"one_lane": timing_fused(0, n), "split2": timing_fused(2, n), "split4": timing_fused(4, n)}
                                for n in (8192, 16384, 32768, 65536, 131072, 262144, 1048576)}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
