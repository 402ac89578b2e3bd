"""Device time of the reset launches next to the step kernel (HIP-event laps on the engine's stream).

    python tools/reset_cost.py [--envs 1048576] > profiles/rNN_reset_cost.json

reset_random = mt_reset_random (all envs: zero pose, FK of the zero pose, K targets by rejection sampling from the
Philox stream); reset_done = mt_reset_done after a 50-step episode (only the finished envs are re-armed)."""
import argparse
import json
import statistics
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import manytor_amd as m  # noqa: E402


def laps(eng, fn, reps):
    eng.lap_times()
    for r in range(reps):
        eng.lap_begin()
        fn(r)
        eng.lap_end()
    t = eng.lap_times()
    return {"median_us": statistics.median(t) * 1e3, "min_us": min(t) * 1e3, "max_us": max(t) * 1e3, "reps": reps}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=1048576)
    ap.add_argument("--reset-split", type=int, default=-1, help="force MT_RESET_SPLIT (0/1); default: the library's choice")
    a = ap.parse_args()
    if a.reset_split >= 0:
        os.environ["MT_RESET_SPLIT"] = str(a.reset_split)
    out = {"envs": a.envs, "reset_split": a.reset_split}
    for name, kw in (("4-DoF reference table", {}), ("7-DoF table", dict(dh_table=m.DH7_TABLE))):
        eng = m.StepEngine(a.envs, 7, **kw)
        eng.reset_random(1, 0)
        for _ in range(300):                       # clocks up
            eng.rollout(1, 1, 0)
        row = {"step_random": laps(eng, lambda r: eng.rollout(1, 1, r), 200),
               "reset_random": laps(eng, lambda r: eng.reset_random(1, r + 1), 50)}

        eng.lap_times()
        t = []
        finished = 0.0
        for r in range(10):
            eng.rollout(50, 1, 0)
            finished = float((eng.done() != 0).mean())
            eng.lap_begin()
            eng.reset_done(1)
            eng.lap_end()
            t.extend(eng.lap_times())
        row["reset_done_after_50_steps"] = {"median_us": statistics.median(t) * 1e3, "min_us": min(t) * 1e3,
                                            "max_us": max(t) * 1e3, "reps": len(t),
                                            "finished_fraction": finished}
        out[name] = row
        eng.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
