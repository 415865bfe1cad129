"""Wall clock of Rocketland.solve_problem for a dispersed batch (scvx_solve: every trajectory until converged / failed / imax).
    python tools/solve_timing.py [--B 8192] [--lib path.so ...]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=8192)
    ap.add_argument("--lib", nargs="*", default=[None])
    a = ap.parse_args()
    from successiveconvexification_amd import _lib, sample_problems as sp
    from successiveconvexification_amd.montecarlo import disperse_ics
    ic = disperse_ics(sp.base_prob_scaled, 0, a.B, 20261004)
    for path in a.lib:
        if path:
            _lib._LIB = None
            _lib.LIB_PATH = os.path.join(ROOT, path)
        from successiveconvexification_amd.batch import ScvxBatch
        from successiveconvexification_amd.dynamics import IntegratorCache
        c = IntegratorCache(sp.base_prob_scaled)
        b = ScvxBatch(c, a.B).init(ic)
        b.solve()
        ts = []
        for _ in range(3):
            b.init(ic)
            c.synchronize()
            t0 = time.perf_counter()
            st, it, nu, dj = b.solve()
            ts.append(time.perf_counter() - t0)
        print("%s B %d solve_problem %.1f ms (min of 3), converged %.4f, failed %.4f, steps mean %.2f max %d" % (
            path or "default", a.B, 1e3 * min(ts), (st == 0).mean(), ((st >= 3)).mean(), it.mean(), it.max()), flush=True)
        b.close(); c.close()


if __name__ == "__main__":
    main()
