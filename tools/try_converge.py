import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dataclasses import replace
from successiveconvexification_amd import sample_problems as sp
from successiveconvexification_amd.batch import ScvxBatch
from successiveconvexification_amd.dynamics import IntegratorCache
import bench
p = replace(sp.base_prob_scaled, mdry=0.55, nuTol=1e-6, delTol=1e-3, imax=40, tf_guess=8.0)
B = 8
c = IntegratorCache(p, npts=10)
b = ScvxBatch(c, B).init(bench.disperse_ics(p, 0, B, 7))
for it in range(30):
    st, nu, dj = b.solve_step()
    rk, cost, its = b.scalars()
    x, u, s = b.trajectory()
    print(it + 1, "status", st, "nu %.2e" % nu.max(), "dJ", np.array2string(dj, precision=2), "rk", rk[:3], "sigma %.3f" % s[0], "mf %.4f" % x[0, -1, 0], flush=True)
    if (st == 0).all(): break
