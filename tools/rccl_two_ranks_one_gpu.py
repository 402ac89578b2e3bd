#!/usr/bin/env python3
"""Probe: does this RCCL accept two ranks on ONE device?  If it does, mt_comm_init / mt_gather_returns get a real
multi-rank check on a 1-GPU box (equal and ragged shards); if it refuses ("duplicate GPU"), the error must come back
through the C ABI as a ManytorError, not as a hang.  Launch:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
           tools/rccl_two_ranks_one_gpu.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import manytor_amd as m  # noqa: E402
from manytor_amd import distributed as D  # noqa: E402

rank, _, world = D.env_from_torchrun()
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
res = {"rank": rank}
for n_total in (40000, 40001):
    base, cnt = D.shard_range(n_total, rank, world)
    eng = m.StepEngine(cnt, 7, device=0, env_id_base=base)
    eng.reset_random(5, 0)
    eng.rollout(7, 5, 0)
    try:
        D.connect(eng, rank, world)
        out = eng.gather_returns()
        eng.sync()
        whole = m.StepEngine(n_total, 7, device=0)
        whole.reset_random(5, 0)
        whole.rollout(7, 5, 0)
        ok = bool(np.array_equal(out.cpu().numpy(), whole.total_reward()))
        res[str(n_total)] = {"gathered_equals_single_handle": ok, "total_envs": eng.total_envs()}
        eng.comm_destroy()
    except Exception as exc:  # noqa: BLE001
        res[str(n_total)] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
    dist.barrier()
print(json.dumps(res), flush=True)
dist.barrier()
dist.destroy_process_group()
