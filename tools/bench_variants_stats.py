"""Like bench_variants.py, with the solver's merit / status distribution: python tools/bench_variants_stats.py libA.so libB.so"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from successiveconvexification_amd import _lib, sample_problems as sp
import bench
B = int(os.environ.get("B", "8192")); STEPS = int(os.environ.get("STEPS", "5"))
ic = bench.disperse_ics(sp.base_prob_scaled, 0, B, 20261004)
for path in sys.argv[1:]:
    _lib._LIB = None
    _lib.LIB_PATH = os.path.join(ROOT, path)
    from successiveconvexification_amd.batch import ScvxBatch
    from successiveconvexification_amd.dynamics import IntegratorCache
    c = IntegratorCache(sp.base_prob_scaled)
    b = ScvxBatch(c, B).init(ic)
    its_all, merit_all, ok = [], [], []
    t0 = time.perf_counter()
    for s in range(STEPS):
        st, nu, dj = b.solve_step()
        sst, its, merit, pobj = b.solver_stats()
        its_all.append(its.mean()); merit_all.append(merit); ok.append((sst == 0).mean())
    t = time.perf_counter() - t0
    m = np.concatenate(merit_all)
    x, u, sg = b.trajectory()
    print(path, "traj-it/s %.0f" % (B * STEPS / t), "ipm its/step", np.round(its_all, 2), "optimal", min(ok),
          "merit max %.2e p99.9 %.2e p99 %.2e mean %.2e  frac>1e-7 %.4f  clean(<1e-8) %.3f" % (m.max(), np.quantile(m, 0.999), np.quantile(m, 0.99), m.mean(), (m > 1e-7).mean(), (m < 1e-8).mean()),
          "cksum x %.12e" % np.abs(x).sum(), flush=True)
    b.close(); c.close()
