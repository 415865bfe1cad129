"""A/B of prebuilt library variants in ONE process (cdna guide rule 24): python tools/bench_variants.py build/libA.so build/libB.so"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from successiveconvexification_amd import _lib, sample_problems as sp
import bench
B = int(os.environ.get("B", "8192"))
ic = bench.disperse_ics(sp.base_prob_scaled, 0, B, 20261004)
res = {}
for path in sys.argv[1:]:
    _lib._LIB = None
    _lib.LIB_PATH = os.path.join(ROOT, path)
    from successiveconvexification_amd.batch import ScvxBatch
    from successiveconvexification_amd.dynamics import IntegratorCache
    c = IntegratorCache(sp.base_prob_scaled)
    b = ScvxBatch(c, B).init(ic)
    ts = []
    for rep in range(3):
        b.init(ic)
        c.synchronize()
        t0 = time.perf_counter()
        b.solve_step_async(); b.solve_step_async()
        c.synchronize()
        ts.append((time.perf_counter() - t0) / 2)
    st, its, merit, pobj = b.solver_stats()
    print(path, "ms/step", [round(1e3 * t, 1) for t in ts], "traj-it/s %.0f" % (B / min(ts)), "ipm its %.1f" % its.mean(), "opt %.4f" % (st == 0).mean(), flush=True)
    b.close(); c.close()
