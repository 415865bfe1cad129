"""Solver-option sweep (refinement steps) on the default library: merit / iteration statistics over 5 SCvx steps."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from successiveconvexification_amd import sample_problems as sp
from successiveconvexification_amd.batch import ScvxBatch
from successiveconvexification_amd.dynamics import IntegratorCache
import bench
B = 8192
ic = bench.disperse_ics(sp.base_prob_scaled, 0, B, 20261004)
for refine in (0, 1, 2):
    c = IntegratorCache(sp.base_prob_scaled)
    b = ScvxBatch(c, B, refine=refine).init(ic)
    its_all, merit_all = [], []
    t0 = time.perf_counter()
    for s in range(5):
        b.solve_step()
        sst, its, merit, pobj = b.solver_stats()
        its_all.append(its.mean()); merit_all.append(merit)
    t = time.perf_counter() - t0
    m = np.concatenate(merit_all)
    print("refine", refine, "traj-it/s %.0f" % (B * 5 / t), "its", np.round(its_all, 2), "merit max %.2e p99.9 %.2e frac>1e-7 %.4f clean(<1e-8) %.3f" % (m.max(), np.quantile(m, 0.999), (m > 1e-7).mean(), (m < 1e-8).mean()), flush=True)
    b.close(); c.close()
