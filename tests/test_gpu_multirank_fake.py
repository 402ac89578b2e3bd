"""comm.hip's MULTI-RANK code path on a one-GPU box.  The real RCCL refuses two ranks on one device
(profiles/r02_rccl_two_ranks_one_gpu_probe.txt), so here librccl is replaced -- through the MT_RCCL_LIB hook the library
already has -- by tests/fake_rccl/ (five entry points over POSIX shared memory; test infrastructure only).  Everything
else is the product path: mt_comm_unique_id -> shipped over the torch.distributed group -> mt_comm_init -> shard-size
exchange -> mt_gather_returns with equal and ragged shards, in line and overlapped on the side stream; the gathered
vector must equal what ONE handle owning all the envs computes.  What this cannot show is RCCL itself."""
import os
import socket
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "fake_rccl", "fake_rccl.cpp")
LIB = os.path.join(HERE, "fake_rccl", "_build", "libfake_rccl.so")


def _build_fake():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        subprocess.run([hipcc, "-O1", "-std=c++17", "-fPIC", "-shared", SRC, "-o", LIB, "-lrt"], check=True)
    return LIB


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, lib, q):
    os.environ["MT_RCCL_LIB"] = lib                      # before the package is imported: comm.hip dlopens this
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist

    import manytor_amd as m
    from manytor_amd import distributed as D
    D.init_process_group("gloo")                         # control plane only
    base, cnt = D.shard_range(n_total, rank, world)
    eng = m.StepEngine(cnt, 7, device=0, env_id_base=base, return_ring=2)
    eng.reset_random(5, 0)
    eng.rollout(7, 5, 0)
    D.connect(eng, rank, world)                          # unique id from rank 0 -> mt_comm_init everywhere
    assert eng.total_envs() == n_total
    in_line = eng.gather_returns()
    eng.sync()
    stats = eng.return_stats()                           # mt_reduce_returns: five numbers per rank over the communicator
    overlapped = eng.gather_begin()                      # snapshot + exchange on the side stream ...
    eng.reset_random(5, 1)                               # ... while the returns are zeroed and the next steps run
    eng.rollout(3, 5, 0)
    ms = eng.gather_wait(host=True)
    assert ms is not None
    second = eng.gather_returns(field=m.lib.F_TOTAL_REWARD)    # the returns of the 3 steps after the reset
    eng.sync()
    stats2 = eng.return_stats()                          # right behind an overlapped gather on the same communicator
    every = [None] * world
    dist.all_gather_object(every, (stats, stats2))
    assert all(s == every[0] for s in every)             # the same numbers on every rank
    if rank == 0:
        q.put((in_line.cpu().numpy(), overlapped.cpu().numpy(), second.cpu().numpy(), stats, stats2))
    dist.barrier()
    eng.comm_destroy()
    eng.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total", [(2, 40000), (2, 40001), (3, 100003)])
def test_multi_rank_gather_through_the_c_abi_with_a_stand_in_transport(world, n_total):
    import torch.multiprocessing as mp

    import manytor_amd as m
    lib = _build_fake()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, lib, q)) for r in range(world)]
    for p in procs:
        p.start()
    in_line, overlapped, second, stats, stats2 = q.get(timeout=240)
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    whole = m.StepEngine(n_total, 7)                     # one handle owning every env: the reference
    whole.reset_random(5, 0)
    whole.rollout(7, 5, 0)
    want = whole.total_reward()
    np.testing.assert_array_equal(in_line, want)
    np.testing.assert_array_equal(overlapped, want)
    whole.reset_random(5, 1)
    whole.rollout(3, 5, 0)
    np.testing.assert_array_equal(second, whole.total_reward())
    assert np.abs(want).max() > 0
    for got, ret in ((stats, want), (stats2, whole.total_reward())):
        assert got["count"] == n_total and got["sum"] == float(ret.astype(np.float64).sum())
        assert got["min"] == float(ret.min()) and got["max"] == float(ret.max())
    assert stats2 == whole.return_stats()                # one handle owning every env gives the same five numbers
