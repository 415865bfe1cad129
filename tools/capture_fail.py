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
ic = bench.disperse_ics(p, 0, B, 7)
b = ScvxBatch(c, B).init(ic)
for it in range(2): b.solve_step()
x, u, s = b.trajectory(); e, d = b.linearization(); rk, cost, its = b.scalars()
xs, us, ss, nu = b.socp_solve(); st, sit, merit, pobj = b.solver_stats()
print("status", st, "iters", sit, "merit", merit)
os.makedirs("gpurun_out", exist_ok=True)
np.savez("gpurun_out/fail_case.npz", x=x, u=u, s=s, e=e, d=d, rk=rk, ic=ic)
