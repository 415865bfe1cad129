#!/usr/bin/env python3
"""bench.py — 6-DoF K=50 SCvx iterations/sec (batch-aggregated) on N MI355X GPUs of one node.

A *step* is one Rocketland.solve_step (rocketland.jl:226-321) applied to every trajectory of the
per-GPU batch: conic subproblem (K4) -> candidate -> K predict_state (K2) -> trust-region update (K5)
-> re-linearisation (K1), all enqueued on one HIP stream.  The metric counts trajectory-iterations:
B trajectories each advancing one solve_step count B.

Workload (BASELINE.json configs[3] shape): SampleProblems.base_prob (exo) normalised,
K=50, Monte-Carlo dispersed initial conditions rIi*(1+0.1U), vIi*(1+0.1U), Philox seed 20261004 with
trajectory b on stream b.  The timed steps are the reference's own workload mix: after every imax-1 = 14
solve_steps (one Rocketland.solve_problem, rocketland.jl:432-443) the batch is put back to create_initial
on the device (scvx_batch_reset: straight-line guess + linearisation, enqueued on the same stream and
inside the timed region, not counted as a step), so rejection runs do not pile up beyond what solve_problem sees.
Timed region: `--warmup` untimed steps, the batch put back to create_initial, then WHOLE solve_problem periods -- `--steps`
rounded up to a multiple of imax-1 = 14 (the JSON line carries both `steps`, what was timed, and `steps_requested`): the cost of a
step varies 10x along a period (cold solves early, one-iteration warm solves after rejected steps), so only whole periods
starting at position 0 give a figure that does not depend on the window (`--no-reset` / `--exact-steps` time exactly `--steps`).
Scaling: at N = 1 the batch is 8192.  At N > 1 the default is BASELINE configs[3] as written -- STRONG scaling of the global batch
8192 (1024 per GPU at N = 8) -- with the weak figure (8192 per GPU) timed beside it in the same line (`weak_scaling`);
`--batch b` alone = weak scaling only, `--global-batch G` = strong scaling only.
Arithmetic: fp64 throughout.

Launch:  python bench.py --gpus N --steps K --warmup W      (N > 1 without WORLD_SIZE in the environment: this process starts N
                                                             rank processes itself, before anything touches the GPU, and
                                                             forwards rank 0's line)
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
    # the round's own passes are over the bench MIX (tools/final_profiles.sh): there a K1 launch inside a step skips the trajectories whose
    # step was rejected, so its per-launch mean is below a full launch's -- reported beside the full-launch figure, not instead of it
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_mix_B%d.json" % B))):
        try:
            ks = json.load(open(f))["kernels"]
            k = next(v for name, v in ks.items() if "linearize" in name)
            n = k["launches_in_fetch_pass"]
            if best is not None:
                best = dict(best, mix_bytes=(2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0 / n, mix_source=os.path.basename(f))
        except Exception:
            pass
    return best


def k4_measured_traffic(B):
    """HBM bytes per socp_kernel launch from the same committed PMC passes: hi = 2 x FETCH_SIZE + WRITE_SIZE, the calibrated figure
    (profiles/r03_stream_ceiling.md: on this kernel's access pattern FETCH_SIZE reports 0.502 of the bytes read, WRITE_SIZE 1.005 of
    the bytes written); lo = FETCH not doubled, kept for comparison with the rounds that carried the bracket."""
    import glob
    best = None
    # *_pmc_mix_*: passes over the bench's own timed mix (per-launch mean over cold and warm steps, like avg_launch_ms); the
    # single-cold-step passes (*_pmc_B*) are the fallback
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_B%d.json" % B))) + sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_mix_B%d.json" % B))):
        try:
            k = json.load(open(f))["kernels"]["scvx::socp_kernel"]
            n = k["launches_in_fetch_pass"]
            best = {"lo": (k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0 / n,
                    "hi": (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0 / n, "source": os.path.basename(f)}
        except Exception:
            pass
    return best


from successiveconvexification_amd.montecarlo import disperse_ics  # noqa: E402  (SURVEY.md 8d law; tests import it from here)


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


def k0_threedof(B, seed):
    """NOT the headline: the 3-DoF lossless-convexification initialiser (FirstRound.solve_initial, initial_solve.jl:17-110;
    BASELINE configs[0] problem class at K = 30) for B dispersed initial conditions through scvx_threedof_solve (host arrays in
    and out), wall clock of the second call."""
    from successiveconvexification_amd import first_round
    from successiveconvexification_amd.defns import DescentProblem
    from successiveconvexification_amd.dynamics import IntegratorCache
    from successiveconvexification_amd.montecarlo import disperse_ics
    p = DescentProblem()
    p.K, p.tf_guess, p.mdry, p.mwet, p.alpha = 30, 6.0, 1.0, 2.0, 0.05
    p.rIi, p.vIi = np.array([4.0, 2.0, 0.0]), np.array([-0.5, -0.5, 0.3])
    c = IntegratorCache(p)
    ic = disperse_ics(p, 0, B, seed)
    first_round.solve_initial_batch(c, ic)
    t0 = time.perf_counter()
    sol, st, info = first_round.solve_initial_batch(c, ic)
    t = time.perf_counter() - t0
    c.close()
    return {"workload": "3-DoF landing SOCP, K=30, flyable instance, 10% dispersed (rIi, vIi), tol 1e-9", "B": int(B), "ms": 1e3 * t,
            "solves_per_s": B / t, "ipm_iters_mean": float(info[:, 0].mean()), "optimal_frac": float(np.mean(st == 0))}


def cpu_baseline(npts, seed, steps, reps=5):
    """The CPU twin (oracle/scvx_port.cpp = the device solver core compiled for the host, + oracle/scvx_oracle.c,
    OpenMP over trajectories) on a bounded sample of the SAME workload: `steps` solve_steps from create_initial (one
    Rocketland.solve_problem when steps = imax-1 = 14, the mix the device is timed on), median of `reps` repetitions,
    once on all host cores of this box's share and once single-threaded."""
    import oracle
    from oracle import model, port
    oracle.use_native(True)   # -O3 -march=native builds of the two oracle sources, compiled on this host (oracle/__init__.py)
    po = model.base_prob_scaled()
    cores = int(os.environ.get("SCVX_CPU_THREADS", min(host_cores(), 16)))  # one GPU's share of a pool host is 16 cores

    def timed(B, threads):
        ic = model.disperse_ics(po, B, seed)
        port.scvx_steps(po, ic[:min(B, threads)], 1, nsub=npts, nthreads=threads, warm_start=True)   # warm the libraries
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            o = port.scvx_steps(po, ic, steps, nsub=npts, nthreads=threads, warm_start=True)   # same solver options as the device
            ts.append(time.perf_counter() - t0)
        its = float(np.mean(np.concatenate(o["iters"])))
        return B * steps / float(np.median(ts)), float(np.median(ts)), its
    # ~100 solves/s per core: 8 trajectories per core x 14 steps ~ 1 s per repetition
    vall, tall, its = timed(8 * cores, cores)
    v1, t1, _ = timed(8, 1)
    oracle.use_native(False)
    return {"value": vall, "unit": "traj-iter/s", "cores": int(cores), "kind": "port",
            "flags": "gcc/g++ " + oracle.NATIVE_FLAGS + " (built on this host; the parity build of the same sources is -O2 -ffp-contract=off)",
            "single_thread": {"value": v1, "cores": 1, "sample": f"8 trajectories x {steps} solve_steps, median of {reps} reps of {t1:.1f} s"},
            "ipm_iters_mean": its,
            "sample": f"{8 * cores} dispersed trajectories x {steps} solve_steps from create_initial (same seed / law / step mix as "
                      f"the device run), OpenMP over trajectories, median of {reps} reps of {tall:.1f} s; same algorithm as the "
                      f"device path (scvx_ipm_core.hpp incl. its warm start after rejected steps + RK4 npts={npts}); the Julia reference itself cannot run here"}


def oracle_full_run_fixture(tol):
    """the oracle's recorded solve_problem at solver tolerance `tol` (1e-9, the oracle's default, is the unsuffixed file)"""
    return os.path.join(ROOT, "tests", "golden", "oracle_scvx_full.npz" if abs(tol - 1e-9) < 1e-24 else "oracle_scvx_full_tol%g.npz" % tol)


def traj_linf_vs_oracle(cache_cls, batch_cls, prob, npts, tol=None):
    """Second half of the headline metric ("traj L-inf vs ref"): a complete Rocketland.solve_problem of the sample
    problem (B = 1, imax-1 = 14 solve_steps) on the device against the oracle's recorded run, a committed fixture
    (tests/golden/oracle_scvx_full*.npz — data; generated by tests/golden/make_oracle_full_run.py).  Outside the timed region.
    tol = None: the device at its default (1e-8) against the oracle at ITS default (1e-9) -- rounds 1-4's figure.
    An explicit tol: device and oracle at that same tolerance."""
    f = oracle_full_run_fixture(1e-9) if tol is None else oracle_full_run_fixture(tol)
    if not os.path.exists(f) or npts != 10:
        return None
    g = np.load(f)
    c = cache_cls(prob, npts=npts)
    b = (batch_cls(c, 1) if tol is None else batch_cls(c, 1, tol=tol)).init(None)
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
            "solver_tol": {"device": 1e-8 if tol is None else tol, "oracle": 1e-9 if tol is None else tol},
            "ref": "oracle (IPM on the exact build_model rows + RK4 npts=10); parity with the Julia reference itself is unpinned"}


def traj_linf_vs_oracle_batch32(cache, batch_cls, B, ic, seed):
    """The same figure on the HEADLINE batch (round 6): all B = 8192 dispersed trajectories run a complete solve_problem (14 solve_steps) on the
    device and 32 of them, indices spread over 0 ... 8191, are compared with the independent oracle's recorded runs at the same solver tolerance
    (tests/golden/oracle_scvx_batch32_tol1e-08.npz, made by tests/golden/make_oracle_batch_runs.py) at EVERY step: radius schedule (= accept /
    reject / grow decisions) and iterates, by component group.  Outside the timed region."""
    f = os.path.join(ROOT, "tests", "golden", "oracle_scvx_batch32_tol1e-08.npz")
    if not os.path.exists(f):
        return None
    g = np.load(f)
    if B != int(g["B"]) or seed != int(g["seed"]) or not np.array_equal(ic[g["index"]], g["ic"]):
        return None
    idx, log = g["index"], g["log"]
    b = batch_cls(cache, B).init(ic)
    worst = {"m_r_v": 0.0, "q_omega": 0.0, "u": 0.0, "sigma": 0.0}
    same = True
    for n in range(log.shape[1]):
        b.solve_step()
        x, u, s = b.trajectory()
        rk, _, _ = b.scalars()
        same = same and bool(np.array_equal(rk[idx], log[:, n, 3]))
        d = np.abs(x[idx] - g["xs"][:, n])
        worst["m_r_v"] = max(worst["m_r_v"], float(d[..., :7].max()))
        worst["q_omega"] = max(worst["q_omega"], float(d[..., 7:].max()))
        worst["u"] = max(worst["u"], float(np.abs(u[idx] - g["us"][:, n]).max()))
        worst["sigma"] = max(worst["sigma"], float(np.abs(s[idx] - log[:, n, 5]).max()))
    b.close()
    return dict(worst, x=max(worst["m_r_v"], worst["q_omega"]), trajectories=int(len(idx)), solve_steps=int(log.shape[1]),
                distinct_radius_schedules=int(len({tuple(l[:, 3]) for l in log})), same_accept_reject_sequence_all=same,
                solver_tol={"device": 1e-8, "oracle": 1e-8},
                note="worst over 32 trajectories x 14 steps.  Mass / position / velocity agree to a few 1e-6; the quaternion / body-rate path is "
                     "the flat direction of these subproblems (two runs of the SAME solver at tol 1e-8 and 1e-10 end 5e-4 apart in omega), so "
                     "the undispersed sample's 3.6e-5 does not carry over to the batch; with both sides at 1e-10 the worst of four re-run "
                     "trajectories is 3.3e-5 (q, omega) / 2.4e-7 (m, r, v): distance ~ sqrt(tol), as for any minimiser whose cost is flat to second "
                     "order in those components (they enter it only through the trust-region term)")


def k1_by_npts(cache, batch, torch, K, B, default_npts, sweep=(1, 2, 4, 10), with_f32=True):
    nu, npc = cache.nu, cache.np   # control_dim and columns of a derivative tile (3 / 21; 5 / 25 with the fin extension)
    """K1 alone on the batch's current trajectories for rk4 npts in (1, 2, 4, 10): the HBM fraction of the discretisation
    kernel depends on how much FP64 work a segment carries (SURVEY.md 8d), so the bench states it per npts.  Device
    pointers through the C ABI (scvx_linearize_f64), HIP events on the stream the kernel runs on; outside the timed region."""
    import ctypes as C
    x, u, s = batch.trajectory()
    xd, ud, sd = (torch.tensor(np.ascontiguousarray(a), device="cuda") for a in (x, u, s))
    e = torch.empty((B, K, 14), dtype=torch.float64, device="cuda")
    d = torch.empty((B, K, npc, 14), dtype=torch.float64, device="cuda")
    L, out = cache._L, {}
    ts = torch.cuda.Stream()            # torch's events only see kernels on a torch stream: run K1 on one for this leg
    cache.set_stream(ts.cuda_stream)
    torch.cuda.synchronize()
    for npts in sweep:
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
        out[str(npts)] = {"ms": ms, "achieved_GBps": k1_alg_bytes(K, nu) * B / (ms * 1e-3) / 1e9,
                          "frac": k1_alg_bytes(K, nu) * B / (ms * 1e-3) / HBM_PEAK}
    # fp32 entry point (scvx_linearize_f32: column-per-lane kernel in float arithmetic, float arrays): SURVEY 8d's fp32 row
    xf, uf, sf = xd.float(), ud.float(), sd.float()
    ef = torch.empty((B, K, 14), dtype=torch.float32, device="cuda")
    df = torch.empty((B, K, npc, 14), dtype=torch.float32, device="cuda")
    out32 = {}
    for npts in (sweep if with_f32 else ()):
        cache.set_npts(npts)

        def call32():
            return L.scvx_linearize_f32(cache.handle, B, K, C.c_void_p(xf.data_ptr()), C.c_void_p(uf.data_ptr()),
                                        C.c_void_p(sf.data_ptr()), C.c_float(1.0 / (K + 1)), C.c_void_p(ef.data_ptr()),
                                        C.c_void_p(df.data_ptr()))
        for _ in range(2):
            assert call32() == 0
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record(ts)
        for _ in range(5):
            call32()
        t1.record(ts)
        torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / 5
        out32[str(npts)] = {"ms": ms, "achieved_GBps": k1_alg_bytes(K, nu, s=4) * B / (ms * 1e-3) / 1e9,
                            "frac": k1_alg_bytes(K, nu, s=4) * B / (ms * 1e-3) / HBM_PEAK}
    cache.set_npts(default_npts)
    cache.set_stream(None)   # back to the context's own stream
    return out, out32


def k1_error_by_npts(cache_cls, prob, sweep=(1, 2, 4, 10)):
    """SURVEY H3's accuracy study next to the timings: K1 on the 50 segments of the sample problem's final trajectory (sigma = 7.39,
    dt = 1/51) against their exact discretisation -- state + variational equations by DOP853 at 1e-13, a committed fixture
    (tests/golden/oracle_k1_accuracy.npz, made by tests/golden/make_k1_accuracy_fixture.py).  Max abs error of endpoint / derivative."""
    f = os.path.join(ROOT, "tests", "golden", "oracle_k1_accuracy.npz")
    if not os.path.exists(f):
        return None
    from successiveconvexification_amd.dynamics import linearize_batch
    g = np.load(f)
    out = {}
    for n in sweep:
        c = cache_cls(prob, npts=n)
        e, d = linearize_batch(c, g["x"][None], g["u"][None], np.array([float(g["sigma"])]), float(g["dt"]))
        out[str(n)] = {"endpoint": float(np.abs(e[0] - g["endpoint_dop853"]).max()), "derivative": float(np.abs(d[0] - g["deriv_dop853"]).max())}
        c.close()
    out["note"] = ("vs DOP853 @1e-13 on the final trajectory of the sample problem; derivative entries reach 1e2.  The device/oracle PARITY "
                   "tolerance (1e-12 / 1e-11) holds at every npts -- both run the same scheme; what npts buys is fidelity to the exact flow: "
                   "SURVEY 8c's 1e-9-class endpoint figure needs the reference's npts = 10 (dynamics.jl:112), npts = 4 gives 5e-8, and one "
                   "substep per segment (the only regime in which K1 is HBM-bound) is 1.7e-5 off, outside the 1e-5 trajectory tolerance")
    return out


def k4_traffic_model(tstats, launches):
    """HBM bytes per socp_kernel launch MODELLED for this run: bytes per interior-point iteration as calibrated by PMC passes
    (profiles/r*_k4_traffic_model.json, the newest round on file: 2 x FETCH_SIZE + WRITE_SIZE over launches with known iteration
    counts, written by tools/final_profiles.sh together with the hash of the library sources it measured) times THIS run's own
    device-side counters.  `calibration_matches_running_library` is False when the sources have changed since the calibration --
    the modelled traffic is then a stale figure and says so.  None when no calibration is on file."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_k4_traffic_model.json")))
    if not files or launches <= 0:
        return None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import lib_hash
        running = lib_hash.lib_source_hash()
    except Exception:
        running = None
    # the calibration taken on THIS library if there is one, else the last one on file (flagged as not matching)
    f = next((g for g in reversed(files) if running and json.load(open(g)).get("lib_source_hash") == running), files[-1])
    m = json.load(open(f))
    total = m["bytes_per_ipm_iteration"] * tstats["ipm_iters"] + m.get("bytes_per_solve", 0.0) * tstats["solves"]
    cal = m.get("lib_source_hash")
    # the sources say nothing about -D switches or another LIB_PATH: the LOADED library's own report of its compile-time switches is
    # compared too when the calibration carries one
    try:
        from successiveconvexification_amd import _lib as _l
        running_bin = lib_hash.lib_build_switches(_l.LIB_PATH)
    except Exception:
        running_bin = None
    cal_bin = m.get("lib_build_switches")
    same = bool(cal and running and cal == running) and (cal_bin is None or running_bin is None or cal_bin == running_bin)
    return {"bytes_per_launch": total / launches, "bytes_per_ipm_iteration": m["bytes_per_ipm_iteration"],
            "bytes_per_solve": m.get("bytes_per_solve", 0.0), "calibrated_on_lib_source_hash": cal, "calibrated_on_git_head": m.get("git_head"),
            "running_lib_source_hash": running, "calibrated_on_build_switches": cal_bin, "running_build_switches": running_bin,
            "calibration_matches_running_library": same,
            "source": "profiles/%s: %s" % (os.path.basename(f), m.get("source", ""))}


SOCP_ALG_BYTES = 137 * 1024  # K4 reads the linearisation and the iterate, writes the solution (per trajectory)
FP64_VECTOR_PEAK = 78.6e12    # FLOP/s, AMD's public MI355X figure (not in the microarchitecture guide)
K1_FLOP_PER_SEG_SUBSTEP = 12.0e3  # sparse count of the variational RK4 substep (DESIGN.md kernel table)


def self_launch(n):
    """`bench.py --gpus N` from a plain shell: this parent starts N fresh rank processes (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* in their environment, 127.0.0.1 rendezvous) and waits for them.  It never imports torch and never initialises the GPU;
    nothing re-execs.  Rank 0's stdout is forwarded; a failing rank ends the others and makes the exit code non-zero."""
    import socket
    import subprocess
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SCVX_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    out0 = None
    rc = 0
    try:
        out0, _ = procs[0].communicate()
        for pr in procs:
            pr.wait()
            rc = rc or pr.returncode
    except BaseException:
        rc = rc or 1
        raise
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.terminate()
    if out0:
        sys.stdout.write(out0)
        sys.stdout.flush()
    if rc != 0:
        raise SystemExit("bench.py: a rank process failed (exit codes %s)" % [pr.returncode for pr in procs])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=14)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=0, help="trajectories per GPU (weak scaling); default 8192 (32768 with --config5)")
    ap.add_argument("--global-batch", type=int, default=0, help="strong scaling: total trajectories, split over the ranks")
    ap.add_argument("--npts", type=int, default=10, help="RK4 substeps per segment (Dynamics.rk4 npts)")
    ap.add_argument("--seed", type=int, default=20261004)
    ap.add_argument("--aero", action="store_true", help="SampleProblems.base_prob_aero_scaled (lift_drag tables): BASELINE configs[2] model; NOT the headline workload")
    ap.add_argument("--config5", action="store_true", help="BASELINE configs[4] as named: 6-DoF + aero tables + the fin extension "
                    "(control_dim = 5, a BUILD-DEFINED model: the reference only sketches it in comments), K = 100, seed 20261005; NOT the headline workload")
    ap.add_argument("--no-reset", action="store_true", help="do not return to create_initial every imax-1 steps")
    ap.add_argument("--exact-steps", action="store_true", help="time exactly --steps solve_steps instead of whole solve_problem periods")
    ap.add_argument("--tol", type=float, default=0.0, help="solver tolerance of the headline run (default: the library's 1e-8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traj-check", action="store_true", help="skip the B=1 full-solve parity figure (profiling runs)")
    ap.add_argument("--no-k1-sweep", action="store_true", help="skip the K1-by-npts leg (profiling runs)")
    ap.add_argument("--dump-gathered", default=None, help="rank 0 saves the gathered trajectory records [world*B][(K+1)*(14+NU)+1] as .npy (tests)")
    args = ap.parse_args()
    explicit_batch = bool(args.batch)
    if not args.batch:
        args.batch = 32768 if args.config5 else 8192
    if args.config5 and args.seed == 20261004:
        args.seed = 20261005          # config 5's dispersion seed
    if args.aero and not args.config5 and args.batch == 256 and args.seed == 20261004:
        args.seed = 20261003          # configs[2]'s (aero tables, batch 256) dispersion seed

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus)     # before torch is imported: the parent never touches the GPU

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    backend = os.environ.get("SCVX_DIST_BACKEND", "nccl")  # "gloo": dry-run of the N>1 logic with ranks sharing one GPU
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from successiveconvexification_amd import montecarlo as mc, sample_problems as sp
    from successiveconvexification_amd.batch import ScvxBatch
    from successiveconvexification_amd.dynamics import IntegratorCache

    if args.config5:
        from dataclasses import replace
        from successiveconvexification_amd.defns import AtmosphericData
        z = np.load(os.path.join(ROOT, "tests", "golden", "lift_drag_tables.npz"))   # the reference's aero/lift_drag.csv, repacked
        p = replace(sp.base_prob_fin_scaled(AtmosphericData(z["drag"], z["lift"], z["torque"])), K=100)
        args.aero = True    # not the headline: the legs that belong to the headline workload are skipped
    elif args.aero:
        from successiveconvexification_amd.defns import AtmosphericData
        z = np.load(os.path.join(ROOT, "tests", "golden", "lift_drag_tables.npz"))   # the reference's aero/lift_drag.csv, repacked
        p = sp.base_prob_aero_scaled(AtmosphericData(z["drag"], z["lift"], z["torque"]))
    else:
        p = sp.base_prob_scaled
    K = p.K
    # N = 1: the batch.  N > 1: BASELINE configs[3] as written = STRONG scaling of the global batch (8192 -> 1024 per GPU at N = 8) is
    # the headline, and the weak figure (--batch per GPU) is timed beside it; an explicit --batch / --global-batch picks one mode.
    both_modes = world > 1 and not args.global_batch and not explicit_batch and not args.config5
    if args.global_batch or both_modes:
        scaling, total = "strong", (args.global_batch or args.batch)
    else:
        scaling, total = "weak", args.batch
    shard = mc.Shard(p, total, args.seed, rank, world, scaling)
    B = shard.B
    cache = IntegratorCache(p, device=local_rank, npts=args.npts)  # kernels run on the context's own HIP stream
    solver_kw = {"tol": args.tol} if args.tol > 0 else {}
    batch = ScvxBatch(cache, B, **solver_kw)
    batch.init(shard.ic)  # inputs resident in HBM from here on
    gather_how = None
    if dist is not None:
        if backend != "nccl":
            why = "gloo dry-run"
        elif os.environ.get("SCVX_BENCH_NATIVE_COMM", "1") == "0":
            why = "disabled by SCVX_BENCH_NATIVE_COMM=0"
        else:
            why = mc.bootstrap_comm(cache, dist, rank, world)
        gather_how = (f"scvx_allgather_trajectories: RCCL (ncclAllGather on the library's own communicator), {world} ranks" if why is None
                      else f"torch.distributed {backend} all_gather, {world} ranks ({why})")
        native = why is None

    def barrier():
        cache.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    period = max(p.imax - 1, 1)
    # whole solve_problem periods from position 0 (see the module docstring)
    whole = not args.no_reset and not args.exact_steps
    steps = ((args.steps + period - 1) // period) * period if whole else args.steps

    def timed_steps(bt, on_timed_start=None):
        """warm-up, back to create_initial, barrier, `steps` solve_steps (create_initial again after every period), barrier"""
        cnt = 0
        for _ in range(args.warmup):
            if not args.no_reset and cnt and cnt % period == 0:
                bt.reset()
            bt.solve_step_async()
            cnt += 1
        if whole and cnt:
            bt.reset()               # the timed region starts at position 0 of a solve_problem
            cnt = 0
        barrier()
        if on_timed_start is not None:
            on_timed_start()
        barrier()
        first = cnt
        t0 = time.perf_counter()
        for _ in range(steps):
            if not args.no_reset and cnt and cnt % period == 0:
                bt.reset()            # create_initial again (device-side), as the next solve_problem would
            bt.solve_step_async()
            cnt += 1
        barrier()
        return time.perf_counter() - t0, first

    def start_counters():
        batch.set_profiling(True)
        batch.step_stats(reset=True)

    elapsed, first_timed = timed_steps(batch, start_counters)
    prof, nprof = batch.profile()
    batch.set_profiling(False)
    tstats = batch.step_stats(reset=True)   # what the timed region executed (rank 0's shard)
    st_f, act_f, _ = batch.flags()
    rec64 = batch.trajectory_record() if world == 1 and not args.no_traj_check else None   # for the f32_linearization leg
    done = B * steps  # every trajectory is stepped by every solve_step (failed ones are reported below, not hidden)

    # final trajectories: the only exchange step of the path -- one all-gather over RCCL
    gathered = None
    if dist is not None:
        ptr, n = batch.trajectory_dev()

        class _Dev:  # zero-copy view of the library's HBM buffer
            __cuda_array_interface__ = {"shape": (B, n // B), "typestr": "<f8", "data": (ptr, False), "version": 2}

        mine = torch.as_tensor(_Dev(), device="cuda")
        dev = "cuda"
        if native:
            import ctypes as C
            out = torch.empty((world, B, n // B), dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()
            rc = cache._L.scvx_allgather_trajectories(batch.handle, C.c_void_p(out.data_ptr()))
            if rc == 0:
                cache.synchronize()
            # every rank learns whether the collective was enqueued everywhere: a rank that failed must not leave the others
            # without the line -- all of them then repeat the gather through torch.distributed and the line says so
            rcs = [None] * world
            dist.all_gather_object(rcs, int(rc))
            if any(r != 0 for r in rcs):
                native = False
                gather_how = (f"torch.distributed {backend} all_gather, {world} ranks (scvx_allgather_trajectories failed on rank(s) "
                              f"{[i for i, r in enumerate(rcs) if r != 0]}: {cache._L.scvx_last_error(cache.handle).decode(errors='replace') if rc != 0 else 'ok here'})")
        if not native:
            if dist.get_backend() != "nccl":
                mine, dev = mine.cpu(), "cpu"
            out = mc.gather_records(mine, dist)
        gathered = tuple(out.shape)
        gathered_arr = out
        # every rank sees every shard: rank r's first record must be what rank r holds
        assert torch.equal(out[rank].to(mine.device), mine), "all-gather returned a different local shard"
        elapsed, done_all = mc.reduce_clock(elapsed, done, dist, dev)
    else:
        done_all = done
    weak = None
    if both_modes:
        # the weak-scaling figure beside the strong headline: `--batch` trajectories on EVERY GPU, same timed region, same clock rule
        wshard = mc.Shard(p, args.batch, args.seed, rank, world, "weak")
        wb = ScvxBatch(cache, wshard.B, **solver_kw).init(wshard.ic)
        w_el, _ = timed_steps(wb)
        w_el, w_done = mc.reduce_clock(w_el, wshard.B * steps, dist, dev)
        wb.close()
        weak = {"value": w_done / w_el, "unit": "traj-iter/s", "scaling": "weak", "batch_per_gpu": wshard.B, "global_batch": wshard.global_batch,
                "ms_per_step": 1e3 * w_el / steps, "steps": steps}
    if args.dump_gathered and rank == 0:
        rec_all = gathered_arr.reshape(-1, gathered_arr.shape[-1]).cpu().numpy() if dist is not None else batch.trajectory_record()
        np.save(args.dump_gathered, rec_all)

    if rank == 0:
        st, its, merit, pobj = batch.solver_stats()
        k1_step_ms = prof["linearize"] / max(nprof, 1)   # in the step: launches skip the trajectories whose step was rejected
        # `roofline_k1` is priced on FULL launches (every trajectory linearised): K1 alone, 5 back-to-back launches at the
        # default npts on the batch's trajectories, HIP events on the stream it runs on
        k1_full, _ = k1_by_npts(cache, batch, torch, K, B, args.npts, sweep=(args.npts,), with_f32=False)
        k1_ms = k1_full[str(args.npts)]["ms"]
        k4_ms = prof["socp"] / max(nprof, 1)
        alg = k1_alg_bytes(K, p.nu) * B
        achieved = alg / (k1_ms * 1e-3) if k1_ms > 0 else 0.0
        traffic = k1_measured_traffic(B)
        k4t = k4_traffic_model(tstats, nprof) if B == 8192 and K == 50 and p.nu == 3 else None
        k1_flops = K1_FLOP_PER_SEG_SUBSTEP * K * args.npts * B
        # K4's algorithmic bytes per trajectory: the linearisation + the iterate in, the solution out = 137 KB at K = 50, NU = 3
        socp_alg = SOCP_ALG_BYTES if (K == 50 and p.nu == 3) else k1_alg_bytes(K, p.nu) + 8 * ((K + 1) * (14 + p.nu) + 1 + 14 * K)
        line = {
            "metric": "6-DoF K=50 SCvx iterations/sec (batch)" if not args.config5 else "6-DoF + fin aero K=100 SCvx iterations/sec (batch) -- BASELINE configs[4], NOT the headline metric",
            "value": done_all / elapsed,
            "unit": "traj-iter/s",
            "n_gpus": world,
            "steps": steps,
            "steps_requested": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / steps,
            # what `value` counts, made first-class: a solve_step that is REJECTED (rocketland.jl:299-301) keeps the
            # reference point, so the trajectories ADVANCE at accepted_steps_per_s; a whole Rocketland.solve_problem (imax - 1 solve_steps
            # from create_initial) completes at solve_problems_per_s.  Rank 0's shard x world (every shard draws from the same law).
            "accepted_steps_per_s": world * (tstats["traj_steps"] - tstats["rejected"] - tstats["failed"]) / elapsed,
            "solve_problems_per_s": (done_all / period) / elapsed if not args.no_reset else None,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": ("6-DoF K=%d SCvx, Monte-Carlo dispersed ICs (BASELINE configs[%d] shape, SURVEY 8d law, seed %d), "
                             "SampleProblems.%s, fp64; solve_problem mix: create_initial again every "
                             "%d steps%s" % (K, 4 if args.config5 else (2 if (args.aero and B == 256) else 3), args.seed,
                                             "base_prob_aero normalised + FIN EXTENSION (control_dim = 5): a BUILD-DEFINED model -- the reference carries "
                                             "the fin force only as commented-out code (dynamics.jl:60-69, rocketland.jl:203-209; SURVEY N2), include/scvx.h "
                                             "states what was enabled; parity is against this build's own oracle of the same stated model" if args.config5
                                             else ("base_prob_aero normalised (lift_drag tables)" if args.aero else "base_prob normalised (exo)"),
                                             period, " (disabled)" if args.no_reset else "")),
                "control_dim": p.nu,
                "K": K, "batch_per_gpu": B, "global_batch": shard.global_batch, "rk4_npts": args.npts,
                "solver": "interior-point (NT scaling): optimal = merit < 1e-8, anything else freezes the trajectory as the "
                          "reference's error() does (accept_tol = tol, the default); the solve that "
                          "follows a REJECTED step (same subproblem, radius halved) starts from the previous solve's optimum while "
                          "that point lies inside the new radius, and still ends at 1e-8 (cold_start_only = the same loop without it); "
                          "a solve that ends on its numerical floor is re-run under other step rules (retries = 5) before it counts as failed",
                "parity_contract": {
                    "solver_tol": args.tol if args.tol > 0 else 1e-8,
                    "traj_linf_tol": {"sample_problem_x_u": 1e-4, "dispersed_batch_m_r_v": 2e-5, "dispersed_batch_q_omega": 1e-3, "dispersed_batch_u": 5e-4},
                    "statement": "`value` is measured at solver tolerance 1e-8 (every conic solve to max(pres, dres, relgap) < 1e-8: the "
                                 "tolerance class of the reference's own solver defaults, rocketland.jl:58-59) and its parity figures are taken at "
                                 "the SAME setting on both sides.  (1) traj_linf_vs_oracle: the undispersed sample problem's complete solve_problem stays "
                                 "within 1e-4 of the oracle's run (measured 3.6e-5 in x).  (2) traj_linf_vs_oracle_batch32 (round 6): 32 dispersed "
                                 "trajectories of THIS batch, every one of 14 steps: the radius schedule (accept / reject / grow) equals the oracle's "
                                 "for all 32 x 14, mass / position / velocity within 2e-5 (measured 4.4e-6), and the quaternion / body-rate "
                                 "components -- the flat direction of these subproblems -- within 1e-3 (measured 4.5e-4; u 1.9e-4): the 1e-4 bound "
                                 "of (1) does NOT hold for them on the batch.  SURVEY 8c's proposed 1e-5 is met by the sample problem from tol 3e-10 "
                                 "down (value_at_traj_linf_1e-5: a secondary figure, run with the full retry ladder, under which no solve_step of the "
                                 "timed region fails; with the default ladder 4 of 229,376 end on the solver's numerical floor -- the method eliminates "
                                 "dz through W^-2, cond ~ 16 v0^4, HISTORY.md section 2.2 -- and at 1e-10 no ladder rescues the last 72) and by m / r / v of the dispersed trajectories at 1e-10 "
                                 "(2.4e-7; q / omega 3.3e-5).  The oracle itself is unpinned (no reference-held vectors exist)"},
                "parallelism": f"batch-sharded x{world}, {scaling} scaling", "traj_iters_timed": done_all, "all_gather_shape": gathered,
                "all_gather": gather_how,
            },
            "roofline_k1": {
                "kernel": "scvx::linearize_pcp_kernel (K1, the discretisation kernel SURVEY 8d names; linearize_pc_kernel at npts <= 2; 2 % of the step)",
                "bound": "fp64" if args.npts >= 3 else "hbm",
                "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": achieved / HBM_PEAK, "traffic": traffic["bytes"] if traffic else None,
                "traffic_source": traffic["source"] if traffic else None,
                "traffic_per_launch_of_the_mix": ({"bytes": traffic.get("mix_bytes"), "source": traffic.get("mix_source"),
                                                   "note": "newest PMC passes over the bench mix: in-step launches skip rejected trajectories"}
                                                  if traffic and traffic.get("mix_bytes") else None),
                "alg_bytes_per_launch": alg, "avg_launch_ms": k1_ms, "avg_ms_in_step": k1_step_ms,
                "fp64_frac": (k1_flops / (k1_ms * 1e-3) / FP64_VECTOR_PEAK) if k1_ms > 0 else None,
                "note": "avg_launch_ms = full launches (all trajectories); inside a step K1 skips trajectories whose step was rejected "
                        "(avg_ms_in_step).  K1 at rk4_npts=%d is FP64-FMA-bound, not HBM-bound (SURVEY 8d): fp64_frac = sparse flop count "
                        "(12 kflop per segment per substep) / 78.6 TFLOP/s vector peak; traffic from PMC in profiles/" % args.npts,
            },
            "kernel_ms_per_step": {k: v / max(nprof, 1) for k, v in prof.items()},
            "timed_region": {
                "solve_step_indices": [first_timed, first_timed + steps - 1],
                "indices_within_solve_problem": [(first_timed + i) % period for i in range(steps)] if not args.no_reset else None,
                "period": period, "whole_periods": (steps // period) if whole else None,
                "traj_steps": int(tstats["traj_steps"]), "conic_solves": int(tstats["solves"]),
                "ipm_iters_mean": tstats["ipm_iters"] / max(tstats["solves"], 1.0),
                "warm_started_frac": tstats["warm_started"] / max(tstats["solves"], 1.0),
                "skipped_solves_frac": tstats["skipped"] / max(tstats["traj_steps"], 1.0),
                "rejected_frac": tstats["rejected"] / max(tstats["traj_steps"], 1.0),
                "failed_steps": int(tstats["failed"]), "converged_steps": int(tstats["converged"]),
                "note": "totals over the timed solve_steps of rank 0's shard (device-side counters, scvx_batch_get_step_stats): the "
                        "throughput depends on this mix -- a solve after a rejected step starts from the kept optimum of the same subproblem and, while that "
                        "point lies inside the halved radius, its first residual evaluation (of the new problem) already meets the tolerance: 1 iteration, no factorisation; "
                        "cold_start_only below is the figure with every solve started from scratch",
            },
            # the dominant kernel: K4 (97 % of the step).  (Rounds 1-2 printed K1 here and K4 as roofline_socp.)
            "roofline": None if k4_ms <= 0 else {
                "kernel": "scvx::socp_kernel (K4, the conic solve: the dominant kernel, 97 % of the step)", "bound": "hbm", "unit": "GB/s",
                "peak": HBM_PEAK / 1e9, "avg_launch_ms": k4_ms,
                "alg_bytes_per_launch": socp_alg * B,
                "achieved": socp_alg * B / (k4_ms * 1e-3) / 1e9,
                "frac": socp_alg * B / (k4_ms * 1e-3) / HBM_PEAK,
                "traffic": k4t["bytes_per_launch"] if k4t else None,
                "traffic_kind": "modelled: PMC-calibrated bytes per interior-point iteration / per solve x this run's own iteration and solve "
                                "counts (timed_region), per launch" if k4t else None,
                "traffic_model": k4t,
                "bandwidth_used_frac": (k4t["bytes_per_launch"] / (k4_ms * 1e-3) / HBM_PEAK) if k4t else None,
                "note": "frac is ALGORITHMIC bytes (137 KB per trajectory per solve, SURVEY 8d) over time.  The interior-point iterations "
                        "stream the per-trajectory solver state from HBM several times each (timed_region.ipm_iters_mean iterations per solve "
                        "on average, ~17 for a cold one), so the real traffic is a few hundred times the algorithmic bytes and the kernel runs "
                        "near the streaming rate of the memory system (4.9 TB/s for a 2-reads-1-write stream of this shape, "
                        "profiles/r03_stream_ceiling.md; 4.8-5.1 TB/s re-measured in round 6 at 2 / 4 / 8 wavefronts per SIMD and as per-pass grid-wide "
                        "launches, 5.36 TB/s for a flat grid-stride stream at K4's 4 : 1 read : write mix: profiles/r06_phase_kernel_gate.md).  `traffic` is never a stored byte count divided by another run's time: it is the "
                        "calibrated per-iteration figure times the iterations THIS run executed",
            },
            "solver_stats_last_step": {"ipm_iters_mean": float(np.mean(its)), "ipm_iters_max": int(np.max(its)),
                                       "optimal_frac": float(np.mean(st == 0)), "almost_optimal_frac": float(np.mean(st == 4)),
                                       "failed_frac": float(np.mean((st != 0) & (st != 4))), "merit_max": float(np.max(merit)),
                                       "frozen_trajectories": int(np.sum(act_f == 0))},
        }
        if world == 1 and not args.no_k1_sweep:
            line["roofline_k1_by_npts"], line["roofline_k1_f32_by_npts"] = k1_by_npts(cache, batch, torch, K, B, args.npts)
        if weak is not None:
            line["weak_scaling"] = dict(weak, note="the same timed region with --batch trajectories on EVERY GPU (the headline `value` of an N > 1 "
                                        "run is the STRONG scaling of the global batch, BASELINE configs[3] as written)")
        if world == 1 and not args.no_k1_sweep and not args.aero:
            line["k1_error_vs_dop853_by_npts"] = k1_error_by_npts(IntegratorCache, p)
        if world == 1 and not args.no_traj_check and not args.aero:
            # device and oracle at the SAME solver tolerance (the headline's: 1e-8 unless --tol); rounds 1-4 compared the device at 1e-8
            # with the oracle at its own default 1e-9
            line["traj_linf_vs_oracle"] = traj_linf_vs_oracle(IntegratorCache, ScvxBatch, p, args.npts, tol=(args.tol if args.tol > 0 else 1e-8))
            # ... and rounds 1-4's figure for continuity: the device at its default 1e-8 against the oracle at ITS default 1e-9
            line["traj_linf_vs_oracle_default_tols"] = traj_linf_vs_oracle(IntegratorCache, ScvxBatch, p, args.npts)
            if args.tol <= 0 and not args.config5:
                line["traj_linf_vs_oracle_batch32"] = traj_linf_vs_oracle_batch32(cache, ScvxBatch, B, shard.ic, args.seed)
        if world == 1 and not args.no_traj_check:
            # NOT the headline: the same loop (a) with every solve started cold, as the reference's solver does, and (b) with
            # scvx_solver_opts.reuse_inactive_tr (a conic solve whose optimum is provably unchanged after a rejected step is
            # skipped).  Reported beside `value`, never instead of it.
            batch.close()

            def variant(lin32=False, **kw):
                b2 = ScvxBatch(cache, B, **kw)
                if lin32:
                    b2.set_linearization_f32(True)
                b2.init(shard.ic)
                t2, _ = timed_steps(b2, lambda: b2.step_stats(reset=True))
                ts2 = b2.step_stats(reset=True)
                out = {"value": B * steps / t2, "unit": "traj-iter/s", "ms_per_step": 1e3 * t2 / steps,
                       "ipm_iters_mean": ts2["ipm_iters"] / max(ts2["solves"], 1.0), "failed_steps": int(ts2["failed"])}
                if lin32:   # how far the mixed-precision iterates are from the fp64 run's after the same steps
                    s2, i2, m2, _ = b2.solver_stats()
                    out["optimal_frac"] = float(np.mean(s2 == 0))
                    out["merit_max"] = float(np.max(m2))
                    if rec64 is not None:
                        # per trajectory: after a whole solve_problem a few trajectories take a different accept / reject decision at a
                        # borderline ratio test and end elsewhere (the max); the median / 99th percentile say how close the rest stay
                        dd = np.abs(b2.trajectory_record() - rec64).max(axis=1)
                        out["traj_linf_vs_f64_run"] = {"median": float(np.median(dd)), "p99": float(np.quantile(dd, 0.99)), "max": float(dd.max())}
                b2.close()
                return out
            line["f32_linearization"] = dict(variant(lin32=True), dtype="f64 arithmetic, f32 derivative tiles",
                                             note="NOT the headline: scvx_batch_set_linearization_f32 -- K1 integrates in double and "
                                             "stores dynam[k].derivative as float, the conic solve widens on load and keeps its "
                                             "workspace, norms and pivots in double (BASELINE configs[3-4] 'fp32', SURVEY H7); "
                                             "same loop, same seed as `value`")
            if not args.aero and args.tol <= 0:
                # SURVEY 8c proposed a converged-trajectory L-inf <= 1e-5.  The optimum of each subproblem is flat: with both solvers at the
                # same tolerance the full-run distance is 3.3e-5 ... 3.8e-5 from 1e-8 down to 1e-9 and drops below 1e-5 from 3e-10 on
                # (profiles/r05_tol_sweep.md).  The throughput AT the loosest tolerance that meets 1e-5, with its own parity figure:
                line["value_at_traj_linf_1e-5"] = dict(variant(tol=3e-10, retries=7), solver_tol=3e-10, retries=7,
                                                       traj_linf_vs_oracle=traj_linf_vs_oracle(IntegratorCache, ScvxBatch, p, args.npts, tol=3e-10),
                                                       note="the same timed region with scvx_solver_opts.tol = 3e-10 on the device AND in the oracle "
                                                            "(tests/golden/oracle_scvx_full_tol3e-10.npz): the loosest setting of profiles/r05_tol_sweep.md whose complete "
                                                            "solve_problem of the sample problem stays within 1e-5 of the oracle's.  failed_steps = solve_steps whose conic "
                                                            "solve ended above the tolerance on its numerical floor (the trajectory is frozen, as the reference's error() "
                                                            "would stop it).  At this tolerance 168 of 229,376 solves need a second step rule, 4 are still left after the "
                                                            "default ladder (retries = 5) and NONE after the full one (retries = 7: profiles/r06_failed_steps_by_ladder.md), "
                                                            "so this figure runs the full ladder; the stragglers' extra attempts lengthen every launch they are in (tail), "
                                                            "which is part of the figure.  At the headline's 1e-8 no solve is ever retried")
            line["cold_start_only"] = dict(variant(warm_start=False), note="warm_start = 0: every conic solve starts from the "
                                           "CVXOPT-style cold point, also the re-solve after a rejected step")
            line["with_reuse_inactive_tr"] = dict(variant(reuse_inactive_tr=True), note="opt-in shortcut, off in the headline: after a "
                                                  "rejected step the conic solve is skipped when the optimum just found lies strictly "
                                                  "inside the halved radius (it is then the new optimum too); every solve_step still runs "
                                                  "its propagation, trust-region test and re-linearisation")
        if world == 1 and not args.no_traj_check and not args.aero:
            line["k0_threedof_init"] = k0_threedof(B, args.seed)
        if not args.no_cpu_baseline and world == 1 and not args.aero:  # reported on rank 0 at N=1 only
            line["cpu_baseline"] = cpu_baseline(args.npts, args.seed, period)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
