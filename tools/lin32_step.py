"""Two solve_steps on the same 8,192 dispersed trajectories -- derivative tiles in double, then in float -- for the PMC
passes of tools/pmc_lin32.sh (kernels scvx::socp_kernel / scvx::socp_lin32_kernel).    python tools/lin32_step.py [B]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from successiveconvexification_amd import montecarlo as mc, sample_problems as sp
from successiveconvexification_amd.batch import ScvxBatch
from successiveconvexification_amd.dynamics import IntegratorCache

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
p = sp.base_prob_scaled
c = IntegratorCache(p, npts=10)
ic = mc.disperse_ics(p, 0, B, 20261004)
for lin32 in (False, True):
    b = ScvxBatch(c, B)
    b.set_linearization_f32(lin32)
    b.init(ic)
    b.set_profiling(True)
    for _ in range(2):
        b.solve_step_async()
    c.synchronize()
    prof, n = b.profile()
    st, it, merit, _ = b.solver_stats()
    print("lin32" if lin32 else "f64  ", "socp ms/step %.2f  ipm its %.2f  merit max %.2e" % (prof["socp"] / n, it.mean(), merit.max()), flush=True)
    b.close()
