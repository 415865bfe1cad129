"""What a solve_step costs when no trajectory of the batch is live (masked, not compacted: every block of every kernel returns before
its first load).  python tools/dead_step_cost.py [B]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from successiveconvexification_amd import montecarlo as mc, sample_problems as sp
from successiveconvexification_amd.batch import ScvxBatch
from successiveconvexification_amd.dynamics import IntegratorCache
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
c = IntegratorCache(sp.base_prob_scaled)
b = ScvxBatch(c, B).init(mc.disperse_ics(sp.base_prob_scaled, 0, B, 20261004))
b.solve_step()
st, act, live = b.flags()
for frac in (1.0, 0.5, 0.1, 0.01, 0.0):
    a = np.zeros(B, np.int32); a[:int(round(frac * B))] = 1
    b.set_flags(st, a, a)
    b.solve_step_async(); c.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        b.solve_step_async()
    c.synchronize()
    print("B = %d, %5.1f %% of the trajectories active: %.3f ms per solve_step" % (B, 100 * frac, 1e3 * (time.perf_counter() - t0) / 5), flush=True)
