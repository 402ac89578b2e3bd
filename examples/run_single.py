#!/usr/bin/env python3
"""One arm on the HIP engine, driven the way the reference's single-env demo drives it (test_single.py:9-37):
K = 10 targets, an episode ends early when every target has been collected, a fresh reset before the next one.
This is BASELINE.json configs[0] -- plumbing: one env per launch, every result copied back to the host."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import manytor as tor  # noqa: E402


def run_episode(arm, step_limit):
    """Random actions until done or the step limit; returns (steps taken, episode return)."""
    taken = 0
    for taken in range(1, step_limit + 1):
        _obs, _reward, finished = arm.step(arm.action_sample())
        if finished:
            break
    return taken, arm.total_reward


def main():
    cli = argparse.ArgumentParser(description=__doc__)
    cli.add_argument("--episodes", type=int, default=19)
    cli.add_argument("--step-limit", type=int, default=200)
    cli.add_argument("--targets", type=int, default=10)
    opt = cli.parse_args()

    arm = tor.Environment(opt.targets)
    arm.reset(returnable=True)
    began = time.perf_counter()
    total_steps = 0
    for number in range(1, opt.episodes + 1):
        t0 = time.perf_counter()
        steps, ret = run_episode(arm, opt.step_limit)
        total_steps += steps
        # the reference toggles its viewer on every 10th epoch (test_single.py:23-26); render() only keeps the flag
        if number % 10 == 0:
            arm.render()
        elif arm.rendering:
            arm.render(stop_render=True)
        print(f"episode {number:3d}: return {ret:6.1f} after {steps:3d} steps, {time.perf_counter() - t0:.3f} s")
        arm.reset()
    wall = time.perf_counter() - began
    print(f"{total_steps} env-steps in {wall:.2f} s ({total_steps / wall:.0f} env-steps/s, host round trip per step)")


if __name__ == "__main__":
    main()
