#!/usr/bin/env python3
"""What does the end of an episode cost on the device (small shards)?  50 episodes of T steps queued back to back, timed with
one HIP-event pair around all of them (mt_timer_start / mt_timer_stop), for growing pieces of bench.py's episode end.
    python tools/episode_end_cost.py [n_envs] [T]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import manytor_amd as m  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
T = int(sys.argv[2]) if len(sys.argv) > 2 else 20
E = 50
F_LAST = m.lib.F_LAST_RETURN


def run(label, body):
    e = m.StepEngine(n, 7)
    e.reset_random(1, 0)
    buf = [None, None]
    for rep in range(3):                      # warm
        for ep in range(E):
            body(e, ep, buf)
        e.sync()
    import time
    best, host = 1e9, 1e9
    for rep in range(5):
        e.sync()
        e.timer_start()
        t0 = time.perf_counter()
        for ep in range(E):
            body(e, ep, buf)
        host = min(host, (time.perf_counter() - t0) * 1e6 / E)        # the host's enqueue time per episode
        best = min(best, e.timer_stop() * 1e3 / E)
    e.gather_wait(host=True)
    e.close()
    print(f"{label:58s} {best:8.2f} us per episode  ({best / T:6.2f} us per step)   host enqueue {host:7.1f} us per episode", flush=True)
    return best


def steps_only(e, ep, buf):
    e.rollout(T, 1, 0)


def steps_reset(e, ep, buf):
    e.rollout(T, 1, 0)
    e.reset_random(1, ep)


def steps_snapshot_reset(e, ep, buf):
    e.rollout(T, 1, 0)
    e.gather_wait()
    buf[ep % 2] = e.gather_begin(buf[ep % 2])
    e.reset_random(1, ep)


def steps_inline_gather_reset(e, ep, buf):
    e.rollout(T, 1, 0)
    buf[0] = e.gather_returns(buf[0])
    e.reset_random(1, ep)


def steps_reset_inplace(e, ep, buf):
    e.rollout(T, 1, 0)
    e.reset_random(1, ep)
    buf[ep % 2] = e.gather_begin(buf[ep % 2], field=F_LAST, snapshot=False)


def steps_fused(e, ep, buf):
    e.rollout_fused(T, 1, 0)
    e.reset_random(1, ep)


print(f"{n} envs, episodes of {T} steps, kernel {m.StepEngine(n, 7).step_kernel_name()}")
a = run("steps only", steps_only)
b = run("steps + reset", steps_reset)
c = run("steps + snapshot gather (side stream) + reset", steps_snapshot_reset)
d = run("steps + gather in line + reset", steps_inline_gather_reset)
f = run("steps + reset + in-place gather (side stream)", steps_reset_inplace)
g = run("fused steps + reset", steps_fused)
print(f"reset incl. its boundaries: {b - a:.1f} us;  snapshot gather on top: {c - b:.1f};  in-line gather on top: {d - b:.1f};  in-place on top: {f - b:.1f}")
