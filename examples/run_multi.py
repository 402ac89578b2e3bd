#!/usr/bin/env python3
"""The reference's multi-env demo loop (test_multi.py:11-39) on the HIP engine.

Same loop shape -- reset, `epochs` x `max_steps` of (action_sample, step), the never-true `done == True`
test, render toggle every 10th epoch, per-env returns, reset -- with the number of arms configurable:

    python examples/run_multi.py                       # 3x2 arms, K=7, numpy RNG: the reference's own sizes
    python examples/run_multi.py --shape 1024 1024 --rng device --epochs 5
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import manytor as tor  # noqa: E402  (the drop-in module of this repo)

ap = argparse.ArgumentParser()
ap.add_argument("--shape", type=int, nargs=2, default=(3, 2))
ap.add_argument("--epochs", type=int, default=50)
ap.add_argument("--max-steps", type=int, default=50)
ap.add_argument("--obj-number", type=int, default=7)
ap.add_argument("--rng", choices=("numpy", "device"), default="numpy")
args = ap.parse_args()

env_shape = tuple(args.shape)
multienv = tor.Multienv(env_shape=env_shape, obj_number=args.obj_number, rng=args.rng)
obs = multienv.reset(returnable=True)
env_number = env_shape[0] * env_shape[1]
epochs_time = []
epoch = 0
timer = time.time()
for i in range(1, args.epochs):
    time_epoch = time.time()
    for p in range(args.max_steps):
        action = multienv.action_sample()
        obs2, reward, done = multienv.step(action)
        if done == True:  # noqa: E712  -- a sequence never equals True: kept as in test_multi.py:22
            break
    if not i % 10:
        multienv.render()
    elif multienv.rendering:
        multienv.render(stop_render=True)
    epoch += 1
    multienv.engine.sync()
    epochs_time.append([i, time.time() - time_epoch])
    shown = min(env_number, 8)
    print("Total Reward: ", [multienv.environment[j].total_reward for j in range(shown)],
          "..." if shown < env_number else "")
    print("Epoch: ", epoch)
    multienv.reset()

total_time = time.time() - timer
steps = (args.epochs - 1) * args.max_steps * env_number
print("Total Time: ", total_time)
print(f"{steps} env-steps, {steps / total_time:.3e} env-steps/s (host loop included)")
multienv.render(stop_render=True)
