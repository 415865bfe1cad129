"""Latency of ONE cold solve_step (create_initial -> solve_step, every conic solve from the cold start) by batch size:
median of 7 repetitions, wall clock around the synchronous call.    python tools/cold_step_latency.py [B ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from successiveconvexification_amd import montecarlo as mc, sample_problems as sp
from successiveconvexification_amd.batch import ScvxBatch
from successiveconvexification_amd.dynamics import IntegratorCache

p = sp.base_prob_scaled
c = IntegratorCache(p, npts=10)
print("| B | cold solve_step ms (median of 7) | IPM iterations |")
print("|---|---|---|")
for B in [int(v) for v in sys.argv[1:]] or [1, 64, 256, 512, 1024]:
    ic = mc.disperse_ics(p, 0, B, 20261004)
    b = ScvxBatch(c, B).init(ic)
    ts = []
    for rep in range(8):
        b.reset(); c.synchronize()
        t = time.perf_counter(); b.solve_step(); ts.append(time.perf_counter() - t)
    it = b.solver_stats()[1]
    print("| %d | %.2f | %.1f |" % (B, 1e3 * np.median(ts[1:]), it.mean()), flush=True)
    b.close()
