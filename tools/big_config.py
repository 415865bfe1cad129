"""BASELINE configs[4] shape on one GPU:  python tools/big_config.py K B [lin32]   (lin32: float derivative tiles)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dataclasses import replace
from successiveconvexification_amd import sample_problems as sp
from successiveconvexification_amd.batch import ScvxBatch
from successiveconvexification_amd.dynamics import IntegratorCache
import bench
K, B = int(sys.argv[1]), int(sys.argv[2])
p = replace(sp.base_prob_scaled, K=K)
c = IntegratorCache(p, npts=10)
LIN32 = len(sys.argv) > 3 and sys.argv[3] == "lin32"
t = time.perf_counter(); b = ScvxBatch(c, B); b.set_linearization_f32(LIN32); b.init(bench.disperse_ics(p, 0, B, 20261005)); print("init %.1f s" % (time.perf_counter() - t), flush=True)
for i in range(3):
    t = time.perf_counter(); st, nu, dj = b.solve_step(); dt = time.perf_counter() - t
    sst, sit, merit, pobj = b.solver_stats()
    print("step", i + 1, "%.2f s" % dt, "traj-it/s %.0f" % (B / dt), "status", dict(zip(*np.unique(st, return_counts=True))), "ipm its %.1f max %d" % (sit.mean(), sit.max()), "solver ok %.5f" % (sst == 0).mean(), "nu med %.3e" % np.median(nu), flush=True)
x, u, s = b.trajectory(); print("finite", np.isfinite(x).all(), np.isfinite(u).all(), "last traj sigma", s[-1])
