#!/usr/bin/env python3
"""The reference's single-env demo loop (test_single.py:9-37) on the HIP engine: one arm, K=10 targets,
break on `done`, reset per epoch.  (BASELINE.json configs[0]: plumbing, one env per launch.)"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import manytor as tor  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--epochs", type=int, default=20)
ap.add_argument("--max-steps", type=int, default=200)
ap.add_argument("--obj-number", type=int, default=10)
args = ap.parse_args()

env = tor.Environment(args.obj_number)
obs = env.reset(returnable=True)
epochs_time = []
epoch = 0
timer = time.time()
for i in range(1, args.epochs):
    time_epoch = time.time()
    for p in range(args.max_steps):
        action = env.action_sample()
        obs2, reward, done = env.step(action)
        if done:
            break
    if not i % 10:
        env.render()
    elif env.rendering:
        env.render(stop_render=True)
    epoch += 1
    epochs_time.append([i, time.time() - time_epoch])
    print("Total Reward: ", env.total_reward)
    print("Epoch: ", epoch)
    env.reset()
print("Total Time: ", time.time() - timer)
env.render(stop_render=True)
