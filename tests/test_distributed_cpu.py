"""The N>1 path on CPU: two gloo ranks run the sharding code bench.py runs (successiveconvexification_amd.montecarlo:
Shard, disperse_ics by global index, gather_records, reduce_clock) -- everything of the multi-GPU path except the
device calls and the RCCL transport (the library's communicator needs GPUs; its bootstrap is exercised up to the point
where rank 0 must draw a unique id)."""
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


def _worker(rank, world, port, batch, scaling, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from successiveconvexification_amd import montecarlo as mc, sample_problems as sp
    p = sp.base_prob_scaled
    nrec = (p.K + 1) * 17 + 1
    shard = mc.Shard(p, batch, 20261004, rank, world, scaling)
    # each rank's "solution" is a function of its initial conditions (as a real solve is) and of the GLOBAL index
    rec = np.zeros((shard.B, nrec))
    rec[:, 0] = np.arange(shard.lo, shard.hi)
    rec[:, 1:7] = shard.ic
    out = mc.gather_records(torch.tensor(rec), dist)
    # the old public helper (rec, group=None) still gathers over the default group
    from successiveconvexification_amd.batch import gather_trajectories
    assert torch.equal(gather_trajectories(torch.tensor(rec)), out)
    assert torch.equal(gather_trajectories(torch.tensor(rec), dist.group.WORLD), out)
    t, n = mc.reduce_clock(1.0 + rank, shard.B * 3, dist)
    # the native communicator cannot exist without a GPU: the bootstrap must say so on every rank, not hang
    class _Cache:  # what bootstrap_comm needs of an IntegratorCache
        handle = None
    from successiveconvexification_amd import _lib
    why = mc.bootstrap_comm(_Cache(), dist, rank, world, lib=_lib.lib())
    if rank == 0:
        q.put((out.numpy(), t, n, shard.global_batch, why))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("scaling,batch", [("weak", 6), ("strong", 12)])
def test_two_rank_shard_and_gather(scaling, batch):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, batch, scaling, q)) for r in range(world)]
    for p in procs:
        p.start()
    out, t, n, total, why = q.get(timeout=180)
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    assert total == 12 and out.shape[:2] == (world, total // world)
    flat = out.reshape(total, -1)
    assert np.array_equal(flat[:, 0], np.arange(total))  # global order restored, nothing lost or duplicated
    from successiveconvexification_amd import montecarlo as mc, sample_problems as sp
    assert np.array_equal(flat[:, 1:7], mc.disperse_ics(sp.base_prob_scaled, 0, total, 20261004))  # same ICs as one big batch
    assert t == 2.0 and n == total * 3           # max over ranks of the clock, sum of the work
    assert why is not None and ("unique_id" in why or "scvx_comm_create" in why or "RCCL not loadable" in why)  # no GPU here: the bootstrap reports it, on every rank


def _worker8(rank, world, port, q):
    """the driver's N = 8 strong-scaling shape on CPU: global 8192 -> 1024 per rank, records of the real size"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from successiveconvexification_amd import montecarlo as mc, sample_problems as sp
    p = sp.base_prob_scaled
    nrec = (p.K + 1) * 17 + 1
    shard = mc.Shard(p, 8192, 20261004, rank, world, "strong")
    rec = np.zeros((shard.B, nrec))
    rec[:, 0] = np.arange(shard.lo, shard.hi)
    rec[:, 1:7] = shard.ic
    rec[:, -1] = rank
    out = mc.gather_records(torch.tensor(rec), dist)
    t, n = mc.reduce_clock(0.5 + 0.01 * rank, shard.B * 14, dist)
    if rank == 0:
        q.put((tuple(out.shape), out[:, :, 0].numpy().copy(), out[:, 0, -1].numpy().copy(), out[:, :, 1:7].numpy().copy(), t, n))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_rank_strong_shard_and_gather_shape_of_the_driver_run():
    """VERDICT r4 item 6, the part that can run without eight processes on one card (the GPU box's process guard allows six):
    eight gloo ranks on CPU run the bench's own sharding / gather / clock code at the driver's N = 8 shape -- global batch 8192,
    1024 trajectories per rank, records of (K+1)*17+1 = 868 values -- and the gathered array is (8, 1024, 868) in global order with
    exactly the initial conditions one 8192 batch would draw."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    shape, idx, ranks, ics, t, n = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert shape == (8, 1024, 868)
    assert np.array_equal(idx.reshape(-1), np.arange(8192)) and np.array_equal(ranks, np.arange(8))
    from successiveconvexification_amd import montecarlo as mc, sample_problems as sp
    assert np.array_equal(ics.reshape(8192, 6), mc.disperse_ics(sp.base_prob_scaled, 0, 8192, 20261004))
    assert abs(t - 0.57) < 1e-12 and n == 8192 * 14


def test_strong_scaling_needs_divisible_batch():
    from successiveconvexification_amd import montecarlo as mc, sample_problems as sp
    with pytest.raises(ValueError):
        mc.Shard(sp.base_prob_scaled, 13, 1, 0, 2, "strong")
    s = mc.Shard(sp.base_prob_scaled, 8192, 1, 3, 8, "strong")
    assert (s.lo, s.hi, s.B, s.global_batch) == (3072, 4096, 1024, 8192)   # BASELINE configs[3]: 1024 per GPU at N = 8
    w = mc.Shard(sp.base_prob_scaled, 16, 1, 3, 8, "weak")
    assert (w.lo, w.hi, w.global_batch) == (48, 64, 128)


def _rccl_missing_child(q):
    import ctypes as C
    os.environ["SCVX_RCCL_LIB"] = "/nonexistent/librccl-not-here.so"
    from successiveconvexification_amd import _lib
    L = _lib.lib()
    buf = (C.c_char * 128)()
    q.put((L.scvx_comm_probe(), L.scvx_comm_unique_id(buf)))


def test_missing_rccl_is_an_error_code_not_a_crash():
    """dlerror() was called twice (the second call returns NULL -> std::string(nullptr)).  With the library
    forced to a missing file both entry points must return SCVX_ERR_COMM (-5) and the process must survive."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_missing_child, args=(q,))
    p.start()
    probe, uid = q.get(timeout=120)
    p.join(timeout=60)
    assert p.exitcode == 0 and probe == -5 and uid == -5
