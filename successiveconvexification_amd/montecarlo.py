"""Monte-Carlo batches of independent initial-condition problems, sharded over the GPUs of a node.

New relative to the reference (which solves one trajectory serially): SURVEY.md 8d fixes the dispersion law, 8e the
sharding -- contiguous shards, one process per GPU, NO collective inside the SCvx iteration (solve_step reads no other
problem's data, rocketland.jl:226-321), one all-gather of the final trajectory records at the end.

The gather goes through the library's own RCCL communicator (scvx_comm_create / scvx_allgather_trajectories, what a
Julia host would call); `torch.distributed` is only the bootstrap channel for the 128-byte unique id and the clock
(barrier, max over ranks).  Everything here except the device calls runs unchanged on a CPU `gloo` group, which is how
tests/test_distributed_cpu.py drives it.
"""
import ctypes as C

import numpy as np


def disperse_ics(p, lo, hi, seed, frac=0.1):
    """rIi * (1 + frac U(-1,1)) and vIi * (1 + frac U(-1,1)) per component; trajectory b draws from Philox
    stream b of `seed`, so a shard [lo, hi) gets exactly the rows the whole batch would."""
    ic = np.zeros((hi - lo, 6))
    for b in range(lo, hi):
        rng = np.random.Generator(np.random.Philox(key=seed, counter=[0, 0, 0, b]))
        r = rng.uniform(-1.0, 1.0, size=6)
        ic[b - lo, 0:3] = np.asarray(p.rIi) * (1.0 + frac * r[0:3])
        ic[b - lo, 3:6] = np.asarray(p.vIi) * (1.0 + frac * r[3:6])
    return ic


def shard_range(total: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of `total` trajectories for `rank`; sizes differ by at most one."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class Shard:
    """This rank's part of a Monte-Carlo batch.  scaling = "weak": every rank holds `batch` trajectories (global =
    batch * world); "strong": `batch` is the GLOBAL count, split evenly (the all-gather needs equal shards)."""

    def __init__(self, problem, batch, seed, rank=0, world=1, scaling="weak"):
        if scaling not in ("weak", "strong"):
            raise ValueError("scaling must be 'weak' or 'strong'")
        if scaling == "strong" and batch % world:
            raise ValueError(f"strong scaling needs the global batch ({batch}) divisible by the world size ({world})")
        self.rank, self.world, self.scaling, self.seed = int(rank), int(world), scaling, int(seed)
        self.global_batch = batch * world if scaling == "weak" else batch
        self.lo, self.hi = shard_range(self.global_batch, rank, world)
        self.B = self.hi - self.lo
        self.ic = disperse_ics(problem, self.lo, self.hi, seed)


def bootstrap_comm(cache, dist, rank, world, lib=None):
    """scvx_comm_create on every rank: rank 0 draws the RCCL unique id, `dist` (any initialised torch.distributed
    group) broadcasts its 128 bytes.  Returns None on success, else the reason the native communicator is unavailable
    (the caller then falls back to a torch all-gather and says so)."""
    L = lib if lib is not None else cache._L
    # pre-flight, non-collective: ncclCommInitRank blocks until every rank has arrived, so a rank that cannot even bind
    # the library must be known to all of them BEFORE anyone enters it
    flags = [None] * world
    dist.all_gather_object(flags, int(L.scvx_comm_probe()))
    if any(f != 0 for f in flags):
        return "RCCL not loadable on rank(s) %s" % [i for i, f in enumerate(flags) if f != 0]
    buf = (C.c_char * 128)()
    ok = 1
    if rank == 0:
        ok = 1 if L.scvx_comm_unique_id(buf) == 0 else 0
    obj = [bytes(buf) if ok else None]
    dist.broadcast_object_list(obj, src=0)
    if obj[0] is None:
        return "scvx_comm_unique_id failed on rank 0 (RCCL not loadable)"
    rc = L.scvx_comm_create(cache.handle, C.c_char_p(obj[0]), int(rank), int(world))
    flags = [None] * world
    dist.all_gather_object(flags, int(rc))
    if any(f != 0 for f in flags):
        if rc == 0:
            L.scvx_comm_destroy(cache.handle)
        return "scvx_comm_create failed on rank(s) %s" % [i for i, f in enumerate(flags) if f != 0]
    return None


def gather_records(mine, dist=None, native=None):
    """All-gather of the per-rank records `mine` [B][n] (a torch tensor; equal B on every rank) into [world][B][n].
    native: callable(send_ptr, recv_ptr) -> rc enqueueing the library's RCCL all-gather on device tensors; without it
    the group's own backend is used (RCCL on GPU tensors, gloo on CPU tensors)."""
    import torch
    if dist is None or not dist.is_initialized():
        return mine.unsqueeze(0).clone()
    world = dist.get_world_size()
    out = torch.empty((world,) + tuple(mine.shape), dtype=mine.dtype, device=mine.device)
    if native is not None:
        rc = native(mine.data_ptr(), out.data_ptr())
        if rc != 0:
            raise RuntimeError(f"native all-gather failed ({rc})")
        return out
    if mine.is_cuda:
        dist.all_gather_into_tensor(out, mine.contiguous())
    else:
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine.contiguous())
        out = torch.stack(parts)
    return out


def reduce_clock(elapsed, done, dist=None, device="cpu"):
    """(max over ranks of the timed interval, sum over ranks of the trajectory-iterations executed)."""
    import torch
    if dist is None or not dist.is_initialized():
        return float(elapsed), int(done)
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    n = torch.tensor([float(done)], dtype=torch.float64, device=device)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return float(t.item()), int(n.item())
