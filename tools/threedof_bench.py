"""Wall-clock of the batched 3-DoF initialiser (K0) through the host-array entry point scvx_threedof_solve (includes the
H2D/D2H of the initial conditions and solutions), next to its CPU twin on all host cores.
    python tools/threedof_bench.py [--K 30] [--B 1,256,2048,8192] [--twin]"""
import argparse
import os
import sys
import time
from dataclasses import replace

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--K", type=int, default=30)
    ap.add_argument("--B", default="1,256,2048,8192")
    ap.add_argument("--twin", action="store_true")
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    from oracle import model
    from successiveconvexification_amd import first_round
    from successiveconvexification_amd.defns import DescentProblem
    from successiveconvexification_amd.dynamics import IntegratorCache
    po = replace(model.DescentProblem(), K=a.K, tf_guess=6.0, rIi=np.array([4.0, 2.0, 0.0]), vIi=np.array([-0.5, -0.5, 0.3]),
                 mdry=1.0, mwet=2.0, alpha=0.05)
    p = DescentProblem()
    for f in ("g", "mdry", "mwet", "Tmin", "Tmax", "thetaMax", "gammaGs", "alpha", "K", "tf_guess"):
        setattr(p, f, getattr(po, f))
    p.rIi, p.vIi = po.rIi.copy(), po.vIi.copy()
    c = IntegratorCache(p)
    first_round.solve_initial_batch(c)   # tables, first launch
    print("| B | ms | solves/s | iterations mean (max) | optimal |")
    print("|---|---|---|---|---|")
    for B in [int(x) for x in a.B.split(",")]:
        ic = model.disperse_ics(po, B, 20261004)
        first_round.solve_initial_batch(c, ic)
        ts = []
        for _ in range(a.reps):
            t0 = time.perf_counter()
            sol, st, info = first_round.solve_initial_batch(c, ic)
            ts.append(time.perf_counter() - t0)
        t = float(np.median(ts))
        print(f"| {B} | {1e3 * t:.2f} | {B / t:.0f} | {info[:, 0].mean():.1f} ({int(info[:, 0].max())}) | {(st == 0).mean():.4f} |", flush=True)
        if a.twin:
            from oracle import port
            n = min(B, 512)
            t0 = time.perf_counter()
            port.threedof(po, ic[:n])
            tt = time.perf_counter() - t0
            print(f"|   twin, {os.cpu_count()} threads, {n} solves | {1e3 * tt:.1f} | {n / tt:.0f} | | |", flush=True)


if __name__ == "__main__":
    main()
