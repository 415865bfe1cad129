"""A/B of prebuilt library variants on the bench's own timed region, in ONE process:  python tools/ab_mix.py build/libA.so build/libB.so
For each library: one solve_problem period (14 solve_steps from create_initial, B = 8192 unless B=...) with the default solver
options and with warm_start = 0 (every solve cold), 2 repetitions each; ms per step, the conic kernel's share from the library's own
HIP-event profile, interior-point iterations per solve, optimal fraction."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from successiveconvexification_amd import _lib, sample_problems as sp
import bench
B = int(os.environ.get("B", "8192"))
REPS = int(os.environ.get("REPS", "2"))
ic = bench.disperse_ics(sp.base_prob_scaled, 0, B, 20261004)
for path in sys.argv[1:]:
    _lib._LIB = None
    _lib.LIB_PATH = os.path.join(ROOT, path)
    from successiveconvexification_amd.batch import ScvxBatch
    from successiveconvexification_amd.dynamics import IntegratorCache
    c = IntegratorCache(sp.base_prob_scaled)
    for label, kw in (("mix ", {}), ("cold", {"warm_start": False})):
        b = ScvxBatch(c, B, **kw).init(ic)
        b.solve_step_async(); b.solve_step_async()
        out = []
        for rep in range(REPS):
            b.reset()
            c.synchronize()
            b.set_profiling(True); b.step_stats(reset=True)
            t0 = time.perf_counter()
            for _ in range(14):
                b.solve_step_async()
            c.synchronize()
            t = time.perf_counter() - t0
            prof, n = b.profile(); b.set_profiling(False)
            ts = b.step_stats(reset=True)
            out.append((1e3 * t / 14, prof["socp"] / max(n, 1), ts["ipm_iters"] / max(ts["solves"], 1), int(ts["failed"])))
        st, its, merit, _ = b.solver_stats()
        print("%-40s %s  ms/step %s  socp ms %s  traj-it/s %.0f  ipm its/solve %.2f  failed %d  last-step optimal %.4f merit max %.2e" % (
            path, label, [round(o[0], 2) for o in out], [round(o[1], 2) for o in out], B / (1e-3 * min(o[0] for o in out)), out[-1][2], out[-1][3],
            (st == 0).mean(), merit.max()), flush=True)
        b.close()
    c.close()
