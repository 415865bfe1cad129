"""The 6-DoF conic solve (K4's solver core, on the CPU twin) on RANDOM problem classes: constraint parameters, mass ratio,
horizon, dispersion and model flags drawn at random around the sample problem; several solve_steps each.
    python tools/k4_fuzz.py [--n 30] [--B 16] [--steps 8] [--seed 1] [--fins] [--lib path/to/liboracle_port.so]
"attempted" = conic solves of trajectories that had not failed before (a failed trajectory is frozen: the reference stops with
an error there, rocketland.jl:273-276); "failed" = the first non-optimal solve of a trajectory."""
import argparse
import os
import sys
from dataclasses import replace

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def draw_class(rng, base, fins=False):
    """One random problem class around `base` (an oracle.model.DescentProblem): the law of this tool."""
    from oracle import model
    K = int(rng.choice([12, 25, 50, 64]))
    p = replace(base, K=K, mdry=float(base.mwet * rng.uniform(0.4, 0.999)), Tmin=float(base.Tmax * rng.uniform(0.05, 0.6)),
                deltaMax=float(rng.uniform(5.0, 30.0)), thetaMax=float(rng.uniform(30.0, 120.0)), gammaGs=float(rng.uniform(5.0, 45.0)),
                omMax=float(rng.uniform(20.0, 120.0)), tf_guess=float(rng.uniform(0.5, 12.0)), enforce_dp=bool(rng.integers(0, 2)))
    if fins:
        p = replace(p, fins=True, rFB=model.base_prob().rFB / 1000.0, finmxf=float(rng.uniform(0.002, 0.02)))
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=30)
    ap.add_argument("--B", type=int, default=16)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--fins", action="store_true", help="every class with the fin extension (control_dim = 5)")
    ap.add_argument("--lib", default=None, help="a variant build of liboracle_port.so")
    ap.add_argument("--accept", type=float, default=0.0, help="accept_tol of the solver (default = tol: OPTIMAL or failure; 1e-5 = MOSEK's "
                    "MSK_DPAR_INTPNT_CO_TOL_NEAR_REL = 1000 rule, under which a stalled solve within 1000 tol is still reported OPTIMAL)")
    a = ap.parse_args()
    import oracle
    if a.lib:
        import ctypes
        oracle._PORT = ctypes.CDLL(a.lib)
    from oracle import model, port
    rng = np.random.default_rng(a.seed)
    base = model.base_prob_scaled()
    tot = {}
    worst = 0.0
    attempted = failed = 0
    print("| # | K | mdry | Tmin/Tmax | deltaMax | thetaMax | gammaGs | omMax | tf_guess | dp | status counts (solver) | IPM its mean / max | merit max |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for n in range(a.n):
        p = draw_class(rng, base, a.fins)
        K = p.K
        ic = model.disperse_ics(p, a.B, 500 + n, 0.3)
        try:
            o = port.scvx_steps(p, ic, a.steps, nsub=4, warm_start=True, accept=a.accept)
        except Exception as e:  # noqa: BLE001
            print("| %d | %d | error: %s |" % (n, K, e))
            continue
        st = np.concatenate(o["status"]); it = np.concatenate(o["iters"]); m = np.concatenate(o["merit"])
        S = np.stack(o["status"])                                    # [steps][B]
        bad = (S != 0) & (S != 4)      # 4 = stalled inside the acceptance band (only with --accept > tol)
        alive = np.vstack([np.ones((1, S.shape[1]), bool), ~np.maximum.accumulate(bad, axis=0)[:-1]])   # not failed before this step
        feas = S[0] != 5                                             # status 5 = infeasible initial condition: not a solver failure
        attempted += int((alive & feas[None, :]).sum())
        failed += int((alive & bad & feas[None, :]).sum())
        for k, v in zip(*np.unique(st, return_counts=True)):
            tot[int(k)] = tot.get(int(k), 0) + int(v)
        okm = m[(st == 0) | (st == 4)]
        worst = max(worst, okm.max() if okm.size else 0.0)
        print("| %d | %d | %.3f | %.2f | %.0f | %.0f | %.0f | %.0f | %.1f | %d | %s | %.1f / %d | %.1e |" % (
            n, K, p.mdry, p.Tmin / p.Tmax, p.deltaMax, p.thetaMax, p.gammaGs, p.omMax, p.tf_guess, int(p.enforce_dp),
            {int(k): int(v) for k, v in zip(*np.unique(st, return_counts=True))}, it.mean(), it.max(), m.max()), flush=True)
    print("\nsolver status totals:", tot, " worst merit among optimal / almost optimal: %.2e" % worst)
    print("attempted solves on live trajectories: %d, first failures: %d (%.3f %%), optimal: %.3f %%" % (
        attempted, failed, 100.0 * failed / max(attempted, 1), 100.0 * (1 - failed / max(attempted, 1))))


if __name__ == "__main__":
    main()
