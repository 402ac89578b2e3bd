#!/usr/bin/env python3
"""Many arms in lock step on the HIP engine, in the shape of the reference's multi-env demo (test_multi.py:11-39):
reset, then epochs of `--steps` x (action_sample, step), per-env returns printed, everything reset again.

    python examples/run_multi.py                                   # 3 x 2 arms, K = 7: the reference's own sizes
    python examples/run_multi.py --grid 1024 1024 --rng device --epochs 5
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import manytor as tor  # noqa: E402


def run_epoch(batch, steps):
    for _ in range(steps):
        _obs, _reward, done = batch.step(batch.action_sample())
        # test_multi.py:22 tests `done == True` on the returned sequence, which is never true, so the reference
        # always plays the full epoch; the sequences returned here behave the same way.
        assert not (done == True)  # noqa: E712


def main():
    cli = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    cli.add_argument("--grid", type=int, nargs=2, default=(3, 2), metavar=("ROWS", "COLS"))
    cli.add_argument("--epochs", type=int, default=49)
    cli.add_argument("--steps", type=int, default=50)
    cli.add_argument("--targets", type=int, default=7)
    cli.add_argument("--rng", choices=("numpy", "device"), default="numpy")
    opt = cli.parse_args()

    grid = tuple(opt.grid)
    n_arms = grid[0] * grid[1]
    batch = tor.Multienv(env_shape=grid, obj_number=opt.targets, rng=opt.rng)
    batch.reset(returnable=True)
    began = time.perf_counter()
    for number in range(1, opt.epochs + 1):
        t0 = time.perf_counter()
        run_epoch(batch, opt.steps)
        if number % 10 == 0:                      # viewer toggle of test_multi.py:25-28 (flag only here)
            batch.render()
        elif batch.rendering:
            batch.render(stop_render=True)
        batch.engine.sync()
        shown = [batch.environment[j].total_reward for j in range(min(n_arms, 8))]
        print(f"epoch {number:3d}: returns {shown}{' ...' if n_arms > 8 else ''}  ({time.perf_counter() - t0:.4f} s)")
        batch.reset()
    wall = time.perf_counter() - began
    total = opt.epochs * opt.steps * n_arms
    print(f"{total} env-steps in {wall:.2f} s: {total / wall:.3e} env-steps/s (host loop included)")
    batch.render(stop_render=True)


if __name__ == "__main__":
    main()
