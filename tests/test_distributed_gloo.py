"""world_size-2 gloo rehearsal (CPU) of the N>1 path: shard ranges and the return gather."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from manytor_amd import distributed as D
from oracle import philox_ref as px


def test_shard_ranges_cover_all_envs():
    for n, w in ((1048576, 8), (4194304, 8), (10, 3), (7, 8), (65536, 1)):
        spans = [D.shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == n
        for (b0, c0), (b1, _) in zip(spans, spans[1:]):
            assert b0 + c0 == b1
        assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, _, w = D.init_process_group("gloo")
    base, cnt = D.shard_range(n_total, r, w)
    ids = np.arange(base, base + cnt, dtype=np.uint64)
    # each rank derives its shard's inputs from GLOBAL env ids (what the device RNG does)
    local_actions = px.sample_actions(0x5EED, ids, 7, 4)
    local_returns = torch.from_numpy(local_actions.sum(axis=1).astype(np.float32))
    full = D.gather_returns(local_returns, n_total)
    stats = D.reduce_return_stats(local_returns)
    if r == 0:
        q.put((full.numpy(), stats))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [4096, 1001])
def test_gather_returns_world2_gloo(n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    full, stats = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    expect = px.sample_actions(0x5EED, np.arange(n_total, dtype=np.uint64), 7, 4).sum(axis=1).astype(np.float32)
    np.testing.assert_array_equal(full, expect)            # shard-invariant: same as one rank owning everything
    assert stats[3] == n_total and stats[0] == pytest.approx(float(expect.astype(np.float64).sum()))
    assert stats[1] == expect.min() and stats[2] == expect.max()


def test_gather_returns_single_process_is_identity():
    t = torch.arange(10, dtype=torch.float32)
    assert torch.equal(D.gather_returns(t, 10), t)
