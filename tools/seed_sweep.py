"""Robustness of the headline workload across dispersion seeds: B = 8192, one solve_problem period (14 solve_steps) per seed, default solver
options; failed / frozen steps, iterations per solve, worst merit.    python tools/seed_sweep.py [seed ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from successiveconvexification_amd import sample_problems as sp
from successiveconvexification_amd.batch import ScvxBatch
from successiveconvexification_amd.dynamics import IntegratorCache
import bench
B = 8192
c = IntegratorCache(sp.base_prob_scaled)
print("| seed | dispersion | ms per step | IPM its per solve | failed steps of %d | rejected | worst merit (last step) |" % (14 * B))
print("|---|---|---|---|---|---|---|")
for arg in sys.argv[1:] or ["20261004"]:
    seed, frac = (arg.split(":") + ["0.1"])[:2]
    seed, frac = int(seed), float(frac)
    from successiveconvexification_amd.montecarlo import disperse_ics
    ic = disperse_ics(sp.base_prob_scaled, 0, B, seed, frac)
    b = ScvxBatch(c, B).init(ic)
    b.step_stats(reset=True)
    c.synchronize(); t0 = time.perf_counter()
    for _ in range(14):
        b.solve_step_async()
    c.synchronize()
    t = time.perf_counter() - t0
    ts = b.step_stats(reset=True)
    st, its, merit, _ = b.solver_stats()
    print("| %d | %.2f | %.2f | %.2f | %d | %.3f | %.2e |" % (seed, frac, 1e3 * t / 14, ts["ipm_iters"] / max(ts["solves"], 1), int(ts["failed"]),
                                                         ts["rejected"] / max(ts["traj_steps"], 1), merit.max()), flush=True)
    b.close()
c.close()
