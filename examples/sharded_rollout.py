#!/usr/bin/env python3
"""Envs sharded over the GPUs of one node, in the shape of the reference's multi-env loop (test_multi.py:17-34): episodes of
random-action steps, the per-env returns of ALL ranks gathered at the end of every episode (what test_multi.py:32
prints), everything reset.  One process per GPU; the gather is mt_gather_returns_begin (RCCL all-gather over xGMI on the
engine's side stream), so the next episode runs beside it.

    python examples/sharded_rollout.py --envs-total 1048576                       # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
           examples/sharded_rollout.py --envs-total 4194304                       # BASELINE.json configs[3]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import manytor_amd as m  # noqa: E402
from manytor_amd import distributed as D  # noqa: E402


def main():
    cli = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    cli.add_argument("--envs-total", type=int, default=1 << 20)
    cli.add_argument("--episodes", type=int, default=10)
    cli.add_argument("--steps", type=int, default=50)                  # test_multi.py:8
    cli.add_argument("--targets", type=int, default=7)                 # test_multi.py:9
    cli.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EED)
    cli.add_argument("--stats-only", action="store_true",
                     help="log the returns' sum / min / max over all ranks (mt_reduce_returns: five numbers per rank "
                          "exchanged) instead of gathering every env's return")
    opt = cli.parse_args()

    rank, local_rank, world = D.init_process_group()                   # no-op for a single process
    base, n_local = D.shard_range(opt.envs_total, rank, world)         # contiguous global env ids of this rank
    eng = m.StepEngine(n_local, opt.targets, device=local_rank % m.device_count(), env_id_base=base)
    if world > 1:
        D.connect(eng, rank, world)                                    # RCCL communicator of the engine (C ABI)
    eng.reset_random(opt.seed, 0)
    bufs, pending = [None, None], None
    began = time.perf_counter()
    for ep in range(opt.episodes):
        eng.rollout(opt.steps, opt.seed, ep * opt.steps)               # one launch per step, actions drawn in-kernel
        if opt.stats_only:
            st = eng.return_stats()                                    # collective, synchronous, 40 bytes per rank
            if rank == 0:
                print(f"episode {ep:3d}: {st['count']} envs, mean return {st['mean']:8.3f}, best {st['max']:4.0f}")
            eng.reset_random(opt.seed, ep + 1)
            continue
        if pending is not None:                                        # last episode's returns: complete by now
            eng.gather_wait(host=True)
            if rank == 0:
                r = pending.cpu().numpy()
                print(f"episode {ep - 1:3d}: {r.size} returns gathered, mean {r.mean():8.3f}, best {r.max():4.0f}")
        bufs[ep % 2] = pending = eng.gather_begin(bufs[ep % 2])        # snapshot + exchange on the side stream
        eng.reset_random(opt.seed, ep + 1)                             # ... while this and the next steps proceed
    eng.gather_wait(host=True)
    eng.sync()
    wall = time.perf_counter() - began
    if rank == 0:
        if pending is not None:
            r = pending.cpu().numpy()
            print(f"episode {opt.episodes - 1:3d}: {r.size} returns gathered, mean {r.mean():8.3f}, best {r.max():4.0f}")
        total = opt.envs_total * opt.episodes * opt.steps
        print(f"{total} env-steps on {world} GPU(s) in {wall:.3f} s: {total / wall:.3e} env-steps/s")
    eng.close()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
