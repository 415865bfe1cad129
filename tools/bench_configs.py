"""Measurements for the BASELINE.md results table: BASELINE.json configs 1-3 on one MI355X (fp64)."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from successiveconvexification_amd import sample_problems as sp
from successiveconvexification_amd.batch import ScvxBatch
from successiveconvexification_amd.defns import AtmosphericData
from successiveconvexification_amd.dynamics import IntegratorCache
import bench

def run(name, prob, B, seed, steps=6):
    c = IntegratorCache(prob, npts=10)
    b = ScvxBatch(c, B)
    ic = bench.disperse_ics(prob, 0, B, seed) if B > 1 else None
    b.init(ic)
    b.solve_step_async(); c.synchronize()        # warm-up (first SCvx iteration)
    b.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(steps): b.solve_step_async()
    c.synchronize()
    t = time.perf_counter() - t0
    prof, n = b.profile()
    st, its, merit, pobj = b.solver_stats()
    # full solve_problem from scratch for the convergence / status picture
    b.init(ic)
    t1 = time.perf_counter(); sst, sit, nu, dj = b.solve(); t_solve = time.perf_counter() - t1
    out = dict(config=name, B=B, traj_iter_per_s=B * steps / t, ms_per_step=1e3 * t / steps,
               kernel_ms={k: v / n for k, v in prof.items()}, ipm_iters_mean=float(its.mean()),
               solve_problem_s=t_solve, solve_steps=int(sit.max()), nu_norm_median=float(np.median(nu)),
               status_counts={int(k): int(v) for k, v in zip(*np.unique(sst, return_counts=True))})
    print(json.dumps(out), flush=True)
    b.close(); c.close()

z = np.load(os.path.join(ROOT, "tests", "golden", "lift_drag_tables.npz"))
aero = AtmosphericData(z["drag"], z["lift"], z["torque"])
run("2: 6-DoF K=50 B=1 exo", sp.base_prob_scaled, 1, 0)
run("3: 6-DoF+aero K=50 B=256", sp.base_prob_aero_scaled(aero), 256, 20261003)
run("4: Monte-Carlo K=50 B=8192 exo (fp64)", sp.base_prob_scaled, 8192, 20261004, steps=4)
