#!/usr/bin/env python3
"""bench.py — 6-DoF K=50 SCvx iterations/sec (batch-aggregated) on N MI355X GPUs of one node.

A *step* is one Rocketland.solve_step (rocketland.jl:226-321) applied to every trajectory of the
per-GPU batch: conic subproblem (K4) -> candidate -> K predict_state (K2) -> trust-region update (K5)
-> re-linearisation (K1), all enqueued on one HIP stream.  The metric counts trajectory-iterations:
B trajectories each advancing one solve_step count B.

Workload (BASELINE.json configs[3] shape): SampleProblems.base_prob (exo) normalised,
K=50, Monte-Carlo dispersed initial conditions rIi*(1+0.1U), vIi*(1+0.1U), Philox seed 20261004 with
trajectory b on stream b; every rank holds `--batch` trajectories (weak scaling), rank r taking global
trajectories [r*batch, (r+1)*batch).  Arithmetic: fp64 throughout.

Launch:  python bench.py --gpus 1 --steps 5 --warmup 1
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
                --master-port P bench.py --gpus N --steps K --warmup W
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # B/s, MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters)


def k1_alg_bytes(K, n_u=3, s=8):
    """algorithmic HBM bytes of the discretisation kernel per trajectory per launch."""
    return ((K + 1) * (14 + n_u) + 1 + K * (14 + 14 * (14 + 2 * n_u + 1))) * s


def k1_measured_traffic(B):
    """HBM bytes per K1 launch from the committed PMC passes (profiles/*_pmc_B<batch>.json: FETCH_SIZE and
    WRITE_SIZE collected in separate rocprofv3 --pmc runs; FETCH doubled per the gfx950 half-count note of
    MI355X_MICROARCH.md §HBM).  None when no pass at this batch size is on file."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_B%d.json" % B))):
        try:
            ks = json.load(open(f))["kernels"]
            k = next(v for name, v in ks.items() if "linearize" in name)  # whichever K1 variant was profiled
            n = k["launches_in_fetch_pass"]
            best = {"bytes": (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0 / n, "source": os.path.basename(f)}
        except Exception:
            pass
    return best


def k4_measured_traffic(B):
    """HBM bytes per socp_kernel launch from the same committed PMC passes (lower bound: FETCH not doubled,
    upper bound: FETCH doubled — the kernel mixes narrow strided and wide coalesced reads)."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_B%d.json" % B))):
        try:
            k = json.load(open(f))["kernels"]["scvx::socp_kernel"]
            n = k["launches_in_fetch_pass"]
            best = {"lo": (k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0 / n,
                    "hi": (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0 / n, "source": os.path.basename(f)}
        except Exception:
            pass
    return best


def disperse_ics(p, lo, hi, seed, frac=0.1):
    """SURVEY.md §8d dispersion law; trajectory b draws from Philox stream b."""
    ic = np.zeros((hi - lo, 6))
    for b in range(lo, hi):
        rng = np.random.Generator(np.random.Philox(key=seed, counter=[0, 0, 0, b]))
        r = rng.uniform(-1.0, 1.0, size=6)
        ic[b - lo, 0:3] = p.rIi * (1.0 + frac * r[0:3])
        ic[b - lo, 3:6] = p.vIi * (1.0 + frac * r[3:6])
    return ic


def host_cores():
    """Threads this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / per + 0.5)))
        except Exception:
            pass
    return n


def cpu_baseline(npts, seed, budget_traj_per_core=256):
    """The CPU twin (oracle/scvx_port.cpp + oracle/scvx_oracle.c, OpenMP over trajectories) running the
    same first solve_step on a bounded sample of the same workload, on this box's host cores."""
    from oracle import dynamics as od
    from oracle import model, port
    po = model.base_prob_scaled()
    # one GPU's share of a pool host is 16 cores; never oversubscribe a quota we cannot see
    cores = int(os.environ.get("SCVX_CPU_THREADS", min(host_cores(), 16)))
    B = max(cores * budget_traj_per_core, 16)
    ic = model.disperse_ics(po, B, seed)
    par = od.Params(po)
    K = po.K
    dt = 1.0 / (K + 1)
    x = np.zeros((B, K + 1, 14))
    u = np.zeros((B, K + 1, 3))
    for b in range(B):
        x[b], u[b] = model.linear_points(po, ic[b, :3], ic[b, 3:])
    sig = np.full(B, po.tf_guess)
    e, d = od.linearize(par, x, u, sig, dt, npts)  # create_initial, untimed (as on the GPU)
    port.socp(po, x[:2], u[:2], e[:2], d[:2], 100.0, ic[:2])  # warm the library
    t0 = time.perf_counter()
    r = port.socp(po, x, u, e, d, 100.0, ic, nthreads=cores)
    xn, un, sn = x + r["dx"], u + r["du"], sig + r["ds"]
    xp = od.propagate(par, xn, un, sn, dt, npts)
    jK = -xn[:, K, 0] + po.wNu * np.sqrt(np.sum((xn[:, 1:] - xp) ** 2, axis=(1, 2)))
    _ = jK  # first call: rho = NaN -> accept, grow (rocketland.jl:292-311)
    od.linearize(par, xn, un, sn, dt, npts)
    t = time.perf_counter() - t0
    return {"value": B / t, "unit": "traj-iter/s", "cores": int(cores), "kind": "port",
            "sample": f"{B} dispersed trajectories x 1 solve_step (first SCvx iteration), OpenMP over trajectories, "
                      f"{t:.1f} s wall; same algorithm as the device path (scvx_ipm_core.hpp + RK4 npts={npts})"}


def traj_linf_vs_oracle(cache_cls, batch_cls, prob, npts):
    """Second half of the headline metric ("traj L-inf vs ref"): a complete Rocketland.solve_problem of the sample
    problem (B = 1, imax-1 = 14 solve_steps) on the device against the oracle's recorded run, a committed fixture
    (tests/golden/oracle_scvx_full.npz — data; generated by tests/golden/make_oracle_full_run.py).  Outside the timed region."""
    f = os.path.join(ROOT, "tests", "golden", "oracle_scvx_full.npz")
    if not os.path.exists(f) or npts != 10:
        return None
    g = np.load(f)
    c = cache_cls(prob, npts=npts)
    b = batch_cls(c, 1).init(None)
    wx = wu = ws = 0.0
    same = True
    for n in range(len(g["log"])):
        b.solve_step()
        x, u, s = b.trajectory()
        rk, _, _ = b.scalars()
        same = same and rk[0] == g["log"][n][3]
        wx = max(wx, float(np.abs(x[0] - g["xs"][n]).max()))
        wu = max(wu, float(np.abs(u[0] - g["us"][n]).max()))
        ws = max(ws, abs(float(s[0]) - float(g["log"][n][5])))
    b.close(); c.close()
    return {"x": wx, "u": wu, "sigma": ws, "solve_steps": int(len(g["log"])), "same_accept_reject_sequence": bool(same),
            "ref": "oracle (IPM on the exact build_model rows + RK4 npts=10); parity with the Julia reference itself is unpinned"}


def k1_by_npts(cache, batch, torch, K, B, default_npts):
    """K1 alone on the batch's current trajectories for rk4 npts in (1, 2, 4, 10): the HBM fraction of the discretisation
    kernel depends on how much FP64 work a segment carries, so the bench states it per npts.  Device
    pointers through the C ABI (scvx_linearize_f64), HIP events on the stream the kernel runs on; outside the timed region."""
    import ctypes as C
    x, u, s = batch.trajectory()
    xd, ud, sd = (torch.tensor(np.ascontiguousarray(a), device="cuda") for a in (x, u, s))
    e = torch.empty((B, K, 14), dtype=torch.float64, device="cuda")
    d = torch.empty((B, K, 21, 14), dtype=torch.float64, device="cuda")
    L, out = cache._L, {}
    ts = torch.cuda.Stream()            # a real stream handle: torch's default stream is handle 0, which the library
    cache.set_stream(ts.cuda_stream)    # replaces by its own stream, where torch's events would not see the kernel
    torch.cuda.synchronize()
    for npts in (1, 2, 4, 10):
        cache.set_npts(npts)

        def call():
            return L.scvx_linearize_f64(cache.handle, B, K, C.c_void_p(xd.data_ptr()), C.c_void_p(ud.data_ptr()),
                                        C.c_void_p(sd.data_ptr()), C.c_double(1.0 / (K + 1)), C.c_void_p(e.data_ptr()),
                                        C.c_void_p(d.data_ptr()))
        for _ in range(2):
            assert call() == 0
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record(ts)
        for _ in range(5):
            call()
        t1.record(ts)
        torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / 5
        out[str(npts)] = {"ms": ms, "achieved_GBps": k1_alg_bytes(K) * B / (ms * 1e-3) / 1e9,
                          "frac": k1_alg_bytes(K) * B / (ms * 1e-3) / HBM_PEAK}
    cache.set_npts(default_npts)
    cache.set_stream(torch.cuda.current_stream().cuda_stream)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8192, help="trajectories per GPU")
    ap.add_argument("--npts", type=int, default=10, help="RK4 substeps per segment (Dynamics.rk4 npts)")
    ap.add_argument("--seed", type=int, default=20261004)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traj-check", action="store_true", help="skip the B=1 full-solve parity figure (profiling runs)")
    ap.add_argument("--no-k1-sweep", action="store_true", help="skip the K1-by-npts leg (profiling runs)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if os.environ.get("SCVX_DIST_BACKEND", "nccl") != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SCVX_DIST_BACKEND", "nccl")  # "gloo": dry-run of the N>1 logic with ranks sharing one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from successiveconvexification_amd import sample_problems as sp
    from successiveconvexification_amd.batch import ScvxBatch
    from successiveconvexification_amd.dynamics import IntegratorCache

    p = sp.base_prob_scaled
    K, B = p.K, args.batch
    cache = IntegratorCache(p, device=local_rank, npts=args.npts)
    stream = torch.cuda.current_stream()
    cache.set_stream(stream.cuda_stream)  # the library launches on torch's current stream
    batch = ScvxBatch(cache, B)
    batch.init(disperse_ics(p, rank * B, (rank + 1) * B, args.seed))  # inputs resident in HBM from here on

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        batch.solve_step_async()
    barrier()
    _, _, it0 = batch.scalars()
    batch.set_profiling(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        batch.solve_step_async()
    barrier()
    elapsed = time.perf_counter() - t0
    prof, nprof = batch.profile()
    batch.set_profiling(False)
    _, _, it1 = batch.scalars()
    done = int(np.sum(it1 - it0))  # trajectory-iterations actually executed (inactive trajectories do not count)

    # final trajectories: the only exchange step of the path — one all-gather over RCCL
    gathered = None
    if dist is not None:
        ptr, n = batch.trajectory_dev()

        class _Dev:  # zero-copy view of the library's HBM buffer for RCCL
            __cuda_array_interface__ = {"shape": (B, n // B), "typestr": "<f8", "data": (ptr, False), "version": 2}

        mine = torch.as_tensor(_Dev(), device="cuda")
        dev = "cuda"
        if dist.get_backend() != "nccl":
            mine, dev = mine.cpu(), "cpu"
        from successiveconvexification_amd.batch import gather_trajectories
        out = gather_trajectories(mine)
        gathered = tuple(out.shape)
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        tot = torch.tensor([done], dtype=torch.float64, device=dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        done_all = int(tot.item())
    else:
        done_all = done

    if rank == 0:
        st, its, merit, pobj = batch.solver_stats()
        k1_ms = prof["linearize"] / max(nprof, 1)
        alg = k1_alg_bytes(K) * B
        achieved = alg / (k1_ms * 1e-3) if k1_ms > 0 else 0.0
        traffic = k1_measured_traffic(B)
        line = {
            "metric": "6-DoF K=50 SCvx iterations/sec (batch)",
            "value": done_all / elapsed,
            "unit": "traj-iter/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "6-DoF K=50 SCvx, Monte-Carlo dispersed ICs (BASELINE configs[3] shape, SURVEY 8d law, seed %d), "
                            "SampleProblems.base_prob normalised (exo), fp64" % args.seed,
                "K": K, "batch_per_gpu": B, "global_batch": B * world, "rk4_npts": args.npts,
                "solver": "interior-point (NT scaling), tol 1e-8", "parallelism": f"batch-sharded x{world}",
                "traj_iters_timed": done_all, "all_gather_shape": gathered,
            },
            "roofline": {
                "kernel": "scvx::linearize_pc_kernel (K1)", "bound": "hbm",
                "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": achieved / HBM_PEAK, "traffic": traffic["bytes"] if traffic else None,
                "traffic_source": traffic["source"] if traffic else None,
                "alg_bytes_per_launch": alg, "avg_launch_ms": k1_ms,
                "note": "K1 at rk4_npts=%d is FP64-FMA-bound, not HBM-bound (SURVEY 8d); traffic from PMC in profiles/" % args.npts,
            },
            "kernel_ms_per_step": {k: v / max(nprof, 1) for k, v in prof.items()},
            "roofline_socp": (lambda t, ms: None if not t or ms <= 0 else {
                "kernel": "scvx::socp_kernel (K4, 98 % of the step)", "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK / 1e9,
                "traffic_lo": t["lo"], "traffic_hi": t["hi"], "traffic_source": t["source"], "avg_launch_ms": ms,
                "achieved_lo": t["lo"] / (ms * 1e-3) / 1e9, "achieved_hi": t["hi"] / (ms * 1e-3) / 1e9,
                "frac_lo": t["lo"] / (ms * 1e-3) / HBM_PEAK, "frac_hi": t["hi"] / (ms * 1e-3) / HBM_PEAK,
                "note": "measured HBM traffic of the per-trajectory working set (PMC), not an algorithmic minimum: the "
                        "subproblem data (148 KB per trajectory) would fit on chip, the solver state (613 KB) does not",
            })(k4_measured_traffic(B), prof["socp"] / max(nprof, 1)),
            "solver_stats_last_step": {"ipm_iters_mean": float(np.mean(its)), "ipm_iters_max": int(np.max(its)),
                                       "status_optimal_frac": float(np.mean(st == 0)), "merit_max": float(np.max(merit))},
        }
        if world == 1 and not args.no_k1_sweep:
            line["roofline_k1_by_npts"] = k1_by_npts(cache, batch, torch, K, B, args.npts)
        if world == 1 and not args.no_traj_check:
            line["traj_linf_vs_oracle"] = traj_linf_vs_oracle(IntegratorCache, ScvxBatch, p, args.npts)
        if not args.no_cpu_baseline and world == 1:  # reported on rank 0 at N=1 only
            line["cpu_baseline"] = cpu_baseline(args.npts, args.seed)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
