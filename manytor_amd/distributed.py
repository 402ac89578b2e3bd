"""Env sharding over the GPUs of one node and the set-up of the per-episode return gather.

The reference has no distributed layer (its "multi-env" is a serial Python loop, manytor.py:115-122).  Envs share
nothing, so the step path needs no collective: rank r owns the contiguous block of global env ids given by
``shard_range`` and keys its device RNG with those ids, which makes per-env results independent of the number of
ranks.  The only exchange is the gather of ``total_reward`` at the end of an episode (what test_multi.py:32
prints).  That gather is part of the C ABI (``mt_gather_returns``: RCCL all-gather over xGMI, straight from the arena on
the engine's stream); this module only does the control plane around it with torch.distributed: process-group
initialisation from the torchrun environment and shipping the 128-byte RCCL unique id from rank 0 to the others.

``gloo_gather_returns`` is NOT part of the product path: it is the CPU stand-in used by the world-size-2 gloo tests
and by ``bench.py --rehearsal`` (two ranks on one GPU, which RCCL refuses), to exercise the N > 1 control flow
where no multi-GPU node is available.
"""
from __future__ import annotations

import os

# RCCL shares device buffers between the processes of a node through IPC handles; on hosts whose driver only supports
# dmabuf IPC the legacy mode fails with "hipIpcGetMemHandle: invalid argument".  Read when the HIP runtime starts, so it
# has to be in the environment before the first device call; a value set by the launcher wins.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def shard_range(n_total: int, rank: int, world_size: int):
    """(first global env id, number of envs) of `rank`: contiguous blocks, sizes differ by at most one."""
    if not 0 <= rank < world_size:
        raise ValueError("rank out of range")
    q, r = divmod(int(n_total), int(world_size))
    count = q + (1 if rank < r else 0)
    base = rank * q + min(rank, r)
    return base, count


def env_from_torchrun():
    """(rank, local_rank, world_size) from the variables torch.distributed.run exports; (0, 0, 1) when absent."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend=None, device=None):
    """Initialise torch.distributed from the torchrun environment (no-op for a single process).  `device`: the GPU of
    this rank (default LOCAL_RANK, modulo the number of visible devices in case the launcher shows each rank one card)."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world = env_from_torchrun()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            if device is None:
                device = local_rank % max(1, torch.cuda.device_count())
            torch.cuda.set_device(device)
            dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def exchange_unique_id(make_id, rank: int, group=None) -> bytes:
    """Rank 0 calls `make_id()` (-> 128 bytes naming a new RCCL communicator); every rank returns those bytes.
    Shipped through the torch.distributed process group (any backend), i.e. the rendezvous torchrun already set up.
    If `make_id` fails on rank 0 (librccl not loadable, ncclGetUniqueId error) the failure travels in the SAME
    broadcast, so every rank raises together instead of rank 0 leaving its peers blocked in the collective."""
    import torch.distributed as dist
    box = [None]
    if rank == 0:
        try:
            box = [(bytes(make_id()), None)]
        except Exception as e:                 # noqa: BLE001 -- shipped to every rank below
            box = [(None, f"{type(e).__name__}: {e}")]
    dist.broadcast_object_list(box, src=0, group=group)
    uid, err = box[0]
    if uid is None:
        raise RuntimeError(f"rank 0 could not create the RCCL unique id ({err})")
    if len(uid) != 128:
        raise RuntimeError(f"unique id has {len(uid)} bytes, expected 128")
    return uid


def connect(engine, rank=None, world_size=None, group=None):
    """Give `engine` (a StepEngine holding this rank's shard) its RCCL communicator: rank 0 creates the unique id
    (mt_comm_unique_id), it travels through the torch.distributed group, every rank calls mt_comm_init.  Afterwards
    ``engine.gather_returns()`` all-gathers over xGMI.  Collective: every rank must call it."""
    import torch.distributed as dist

    from .engine import comm_unique_id
    if rank is None:
        rank = dist.get_rank(group)
    if world_size is None:
        world_size = dist.get_world_size(group)
    if world_size == 1:
        return engine
    uid = exchange_unique_id(comm_unique_id, rank, group)
    engine.comm_init(uid, rank, world_size)
    return engine


# ---- a cheap barrier between the ranks of one node -----------------------------------------------------------------
class HostBarrier:
    """Barrier between the processes of ONE node through a small file in /dev/shm: every rank owns a 64-byte slot
    holding the number of barriers it has entered (single writer, monotonic, one aligned 8-byte store), and leaves
    wait() once every slot has reached its own count.  A few microseconds, against the tens of a collective-based
    ``dist.barrier()``; bench.py brackets its timed regions with it (one process per GPU, one node, as launched by
    torch.distributed.run).  Purely host-side: the caller synchronises its device before entering."""

    SLOT = 8                                   # int64 per slot = one 64-byte line

    def __init__(self, path: str, rank: int, world_size: int, create: bool):
        import mmap

        import numpy as np
        self.path, self.rank, self.world, self.epoch = path, int(rank), int(world_size), 0
        size = 64 * self.world
        flags = os.O_RDWR | (os.O_CREAT | os.O_EXCL if create else 0)
        fd = os.open(path, flags, 0o600)
        try:
            if create:
                os.ftruncate(fd, size)         # zero-filled
            elif os.fstat(fd).st_size != size:
                raise RuntimeError(f"{path}: unexpected size")
            self._map = mmap.mmap(fd, size)
        finally:
            os.close(fd)
        self._slots = np.ndarray((self.world, self.SLOT), dtype=np.int64, buffer=self._map)

    def unlink(self):
        """Remove the name (the mapping stays valid for every process that has it open)."""
        try:
            os.unlink(self.path)
        except FileNotFoundError:
            pass

    def wait(self, timeout_s: float = 300.0, sleep_s: float = 0.0):
        """Enter the barrier; returns when every rank has.  sleep_s > 0: poll asleep instead of spinning -- for a wait that
        is expected to be long and must leave the cores to somebody else (bench.py: rank 0's CPU baseline)."""
        import time
        self.epoch += 1
        self._slots[self.rank, 0] = self.epoch
        col = self._slots[:, 0]
        spins, t0 = 0, None
        while int(col.min()) < self.epoch:
            spins += 1
            if sleep_s > 0.0:
                time.sleep(sleep_s)
            if sleep_s > 0.0 or spins & 0x3FF == 0:   # be polite if ranks outnumber cores, and never spin for ever
                time.sleep(0)
                now = time.monotonic()
                t0 = now if t0 is None else t0
                if now - t0 > timeout_s:
                    raise TimeoutError(f"HostBarrier: rank {self.rank} waited {timeout_s:.0f} s at barrier {self.epoch}: "
                                       f"counts {col.tolist()}")

    def close(self):
        self._slots = None
        try:
            self._map.close()
        except (BufferError, ValueError):
            pass


def make_host_barrier(rank: int, world_size: int, group=None):
    """Set up a HostBarrier over the ranks of the process group, or return None (on EVERY rank) if any rank cannot --
    e.g. the ranks do not share /dev/shm because they sit on different nodes.  Collective."""
    import secrets

    import torch
    import torch.distributed as dist
    box = [f"/dev/shm/manytor_barrier_{os.getpid()}_{secrets.token_hex(6)}" if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    path, bar, err = box[0], None, None
    on_gpu = dist.get_backend(group) == "nccl"

    def agree(failed: bool) -> bool:
        flag = torch.tensor([1 if failed else 0], dtype=torch.int32, device="cuda" if on_gpu else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
        return bool(int(flag.item()))

    try:
        if rank == 0:
            bar = HostBarrier(path, rank, world_size, create=True)
    except Exception as e:                     # noqa: BLE001 -- every rank learns about it below
        err = e
    if agree(err is not None):
        return None
    try:
        if rank != 0:
            bar = HostBarrier(path, rank, world_size, create=False)
    except Exception as e:                     # noqa: BLE001
        err = e
    failed = agree(err is not None)
    if rank == 0 and bar is not None:
        bar.unlink()                           # everybody who could has it mapped; no file is left behind
    if failed:
        if bar is not None:
            bar.close()
        return None
    bar.wait()
    return bar


# ---- contingency for bench.py: the gather through torch.distributed's own RCCL process group ----------------------
def attach_torch_gather(engine, n_total: int, rank: int, world_size: int, group=None):
    """If mt_comm_init cannot be set up on a node, bench.py still has to produce its N > 1 line: the returns are then
    all-gathered by torch.distributed (backend nccl = RCCL) on the process group torchrun created, device to device,
    straight from the arena view.  The engine is moved onto torch's current stream so that its launches and the
    collective stay in program order.  Same result layout as mt_gather_returns (global env order, ragged shards
    included); no overlap with the next episode."""
    import torch
    import torch.distributed as dist

    from . import _lib as L
    counts = [shard_range(n_total, r, world_size)[1] for r in range(world_size)]
    if engine.n_envs != counts[rank]:
        raise ValueError(f"rank {rank} holds {engine.n_envs} envs, expected {counts[rank]}")
    cmax, equal = max(counts), len(set(counts)) == 1
    engine.use_torch_stream()
    dev = f"cuda:{engine.device}"
    stage = None if equal else (torch.zeros(cmax, dtype=torch.float32, device=dev),
                                torch.empty(world_size * cmax, dtype=torch.float32, device=dev))

    def gather_returns(out=None, field=None, row=0):
        src = engine.device_tensor(L.F_TOTAL_REWARD if field is None else field)
        src = src if src.dim() == 1 else src[row]
        if out is None:
            out = torch.empty(n_total, dtype=torch.float32, device=dev)
        if equal:
            dist.all_gather_into_tensor(out, src, group=group)
        else:
            send, recv = stage
            send[: src.numel()].copy_(src)
            dist.all_gather_into_tensor(recv, send, group=group)
            off = 0
            for r, c in enumerate(counts):
                out[off: off + c].copy_(recv[r * cmax: r * cmax + c])
                off += c
        return out

    engine.gather_returns = gather_returns
    engine.gather_begin = lambda out=None, field=None, row=0, snapshot=True: gather_returns(out, field, row)
    engine.gather_wait = lambda host=False: 0.0 if host else None
    engine.total_envs = lambda: n_total
    return engine


# ---- CPU stand-in for tests and bench.py --rehearsal (not the product path) --------------------------------------
def gloo_gather_returns(local_returns, n_total: int, group=None):
    """All-gather per-env returns (1-D CPU tensor of this rank's shard) into one (n_total,) tensor in global env order
    with torch.distributed/gloo: same result layout as mt_gather_returns, ragged shards included."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local_returns.clone()
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    counts = [shard_range(n_total, r, world)[1] for r in range(world)]
    if local_returns.numel() != counts[rank]:
        raise ValueError(f"rank {rank} holds {local_returns.numel()} returns, expected {counts[rank]}")
    cmax = max(counts)
    send = torch.zeros(cmax, dtype=local_returns.dtype)
    send[: local_returns.numel()] = local_returns
    out = torch.empty(world * cmax, dtype=send.dtype)
    dist.all_gather_into_tensor(out, send, group=group)
    if all(c == cmax for c in counts):
        return out
    return torch.cat([out[r * cmax: r * cmax + counts[r]] for r in range(world)])


def attach_gloo_gather(engine, n_total: int, rank: int, world_size: int, group=None):
    """bench.py --rehearsal: replace engine.gather_returns by device -> host -> gloo all-gather -> device, so that two
    ranks sharing one GPU can run the N > 1 control flow.  Never used when real GPUs per rank are available."""
    import numpy as np
    import torch

    def gather_returns(out=None, field=None, row=0):
        from . import _lib as L
        vals = engine.get(L.F_TOTAL_REWARD if field is None else field)
        local = torch.from_numpy(vals if vals.ndim == 1 else np.ascontiguousarray(vals[:, row]))
        full = gloo_gather_returns(local, n_total, group)
        if out is None:
            out = torch.empty(n_total, dtype=torch.float32, device=f"cuda:{engine.device}")
        out.copy_(full)
        return out

    engine.gather_returns = gather_returns
    engine.gather_begin = lambda out=None, field=None, row=0, snapshot=True: gather_returns(out, field, row)   # no overlap
    engine.gather_wait = lambda host=False: 0.0 if host else None
    engine.total_envs = lambda: n_total
    return engine
