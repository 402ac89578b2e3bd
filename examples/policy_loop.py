#!/usr/bin/env python3
"""Policy-in-the-loop use of the engine with everything resident on the GPU.

A toy torch policy (one linear layer on the observation) picks the next action of every arm; the action is written
straight into the engine's SoA action buffer (a zero-copy (D, N) view), the engine steps, and the next observation
is read through another zero-copy view.  Nothing crosses PCIe; torch and the engine share one stream.

    python examples/policy_loop.py --envs 1048576 --steps 200
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import manytor_amd as m  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=1 << 20)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--targets", type=int, default=7)
ap.add_argument("--graph", action="store_true", help="capture policy + env step into one HIP graph and replay it")
args = ap.parse_args()

eng = m.StepEngine(args.envs, args.targets)
eng.use_torch_stream()
eng.reset_random(seed=1, episode=0)
eng.observe()

obs = eng.device_tensor(m.lib.F_OBS)            # (3K, N) view of the observation rows
act = eng.device_tensor(m.lib.F_ACTIONS)        # (D, N)  view of the staging buffer mt_step reads
ret = eng.device_tensor(m.lib.F_TOTAL_REWARD)   # (N,)
w = torch.randn(4, 3 * args.targets, device="cuda") * 0.5


def policy():
    # SoA in, SoA out: (D, 3K) @ (3K, N) -> (D, N) degrees, squashed into [-180, 180)
    torch.tanh(w @ obs, out=act).mul_(179.0)


def one_step():
    policy()
    eng.step()


for _ in range(5):
    one_step()
torch.cuda.synchronize()
if args.graph:
    # The engine launches on torch's stream and mt_step calls nothing capture-hostile (no malloc, no sync),
    # so torch's stream capture records the policy kernels and the env step into one HIP graph.
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        eng.use_torch_stream()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            one_step()
    torch.cuda.synchronize()
    run = g.replay
else:
    run = one_step
t0 = time.perf_counter()
for _ in range(args.steps):
    run()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{args.envs} arms x {args.steps} steps with a torch policy in the loop: {args.envs * args.steps / dt:.3e} env-steps/s "
      f"({dt / args.steps * 1e6:.1f} us per step incl. the policy GEMM{', one HIP graph per step' if args.graph else ''}); "
      f"mean return {ret.mean().item():.2f}")
