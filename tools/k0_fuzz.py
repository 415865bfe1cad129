"""Random 3-DoF landing instances through K0 on the device and through its CPU twin: statuses, iteration counts and objectives
must agree.  python tools/k0_fuzz.py [--n 24] [--B 32] [--seed 1]"""
import argparse
import os
import sys
from dataclasses import replace

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=24)
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    from oracle import model, port
    from successiveconvexification_amd import first_round
    from successiveconvexification_amd.defns import DescentProblem
    from successiveconvexification_amd.dynamics import IntegratorCache
    rng = np.random.default_rng(a.seed)
    bad = 0
    print("| # | K | tf | mwet/mdry | alpha | Tmin..Tmax | thetaMax | gammaGs | device status counts | twin status counts | its dev / twin | max rel obj diff |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for n in range(a.n):
        K = int(rng.choice([8, 20, 30, 45]))
        tf = float(rng.uniform(2.0, 10.0))
        mdry = 1.0
        mwet = float(rng.uniform(1.2, 3.0))
        alpha = float(rng.uniform(0.01, 0.2))
        Tmax = float(rng.uniform(2.0, 8.0))
        Tmin = float(rng.uniform(0.05, 0.4)) * Tmax
        th = float(rng.choice([30.0, 60.0, 90.0]))
        gs = float(rng.choice([10.0, 20.0, 35.0]))
        r0 = np.array([rng.uniform(2, 6), rng.uniform(-3, 3), rng.uniform(-1, 1)])
        v0 = np.array([rng.uniform(-1.5, 0.2), rng.uniform(-1, 1), rng.uniform(-0.5, 0.5)])
        po = replace(model.DescentProblem(), K=K, tf_guess=tf, rIi=r0, vIi=v0, mdry=mdry, mwet=mwet, alpha=alpha, Tmax=Tmax, Tmin=Tmin,
                     thetaMax=th, gammaGs=gs)
        p = DescentProblem()
        for f in ("g", "mdry", "mwet", "Tmin", "Tmax", "thetaMax", "gammaGs", "alpha", "K", "tf_guess"):
            setattr(p, f, getattr(po, f))
        p.rIi, p.vIi = r0.copy(), v0.copy()
        ic = model.disperse_ics(po, a.B, 100 + n)
        c = IntegratorCache(p)
        sol, st, info = first_round.solve_initial_batch(c, ic)
        tw, tst, tinfo = port.threedof(po, ic)
        c.close()
        both = (st == 0) & (tst == 0)
        rel = np.abs(info[both, 1] - tinfo[both, 1]) / np.maximum(1.0, np.abs(tinfo[both, 1])) if both.any() else np.zeros(1)
        okrow = np.array_equal(st, tst) and rel.max() < 1e-7
        bad += 0 if okrow else 1
        print("| %d | %d | %.1f | %.2f | %.3f | %.2f..%.2f | %.0f | %.0f | %s | %s | %.1f / %.1f | %.1e %s|" % (
            n, K, tf, mwet, alpha, Tmin, Tmax, th, gs, dict(zip(*np.unique(st, return_counts=True))), dict(zip(*np.unique(tst, return_counts=True))),
            info[:, 0].mean(), tinfo[:, 0].mean(), rel.max(), "" if okrow else "**MISMATCH** "), flush=True)
    print("\nrows where device and twin disagree: %d of %d" % (bad, a.n))


if __name__ == "__main__":
    main()
