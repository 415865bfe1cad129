"""How many solve_steps of the headline batch end on the solver's numerical floor at a tight tolerance, by ladder length:
B = 8192, two solve_problem periods (229,376 solve_steps), tol and retries from the command line.
    python tools/failed_steps_at_tol.py 3e-10 5 7 [max_iter]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from successiveconvexification_amd import sample_problems as sp
from successiveconvexification_amd.batch import ScvxBatch
from successiveconvexification_amd.dynamics import IntegratorCache
import bench
tol = float(sys.argv[1])
B = 8192
ic = bench.disperse_ics(sp.base_prob_scaled, 0, B, 20261004)
c = IntegratorCache(sp.base_prob_scaled)
print("| tol | retries | failed steps of %d | frozen trajectories | ms per step | IPM iterations per solve |" % (2 * 14 * B))
print("|---|---|---|---|---|---|")
for r in [int(v) for v in sys.argv[2:]] or [5]:
    b = ScvxBatch(c, B, tol=tol, retries=r).init(ic)
    b.step_stats(reset=True)
    c.synchronize(); t0 = time.perf_counter()
    failed = 0; frozen = 0
    for per in range(2):
        if per: b.reset()
        for _ in range(14):
            b.solve_step_async()
        c.synchronize()
        st, its, merit, _ = b.solver_stats()
    t = time.perf_counter() - t0
    ts = b.step_stats(reset=True)
    print("| %g | %d | %d | %d | %.2f | %.2f |" % (tol, r, int(ts["failed"]), int(ts.get("frozen", 0)), 1e3 * t / 28, ts["ipm_iters"] / max(ts["solves"], 1)), flush=True)
    b.close()
c.close()
