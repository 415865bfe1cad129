"""The N>1 path on CPU: two gloo ranks shard a Monte-Carlo batch and all-gather the trajectory records."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, nrec, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from successiveconvexification_amd.batch import gather_trajectories, shard_range
    lo, hi = shard_range(total, rank, world)
    # each rank's "solution" is a function of the GLOBAL trajectory index, as a real solve would be
    rec = torch.tensor(np.arange(lo, hi)[:, None] * 1000.0 + np.arange(nrec)[None, :], dtype=torch.float64)
    out = gather_trajectories(rec)
    if rank == 0:
        q.put(out.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather():
    world, total, nrec = 2, 12, (50 + 1) * 17 + 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, nrec, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert out.shape == (world, total // world, nrec)
    flat = out.reshape(total, nrec)
    assert np.array_equal(flat[:, 0], np.arange(total) * 1000.0)  # global order restored, nothing lost or duplicated
    assert np.array_equal(flat[5], 5000.0 + np.arange(nrec))
