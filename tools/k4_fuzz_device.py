"""Device fuzz of the conic solve's executors on RANDOM problem classes (the classes of tools/k4_fuzz.py): the four-wavefront
executor (two-ended factorisation) against the one-wavefront executor, several solve_steps each -- solver statuses per step,
iteration counts and the final trajectories.    python tools/k4_fuzz_device.py [--n 20] [--B 16] [--steps 6] [--seed 1] [--fins] [--retries R]"""
import argparse
import os
import sys
from dataclasses import replace

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=20)
    ap.add_argument("--B", type=int, default=16)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--fins", action="store_true", help="every class with the fin extension (control_dim = 5, build-defined)")
    ap.add_argument("--waves", default="4", help="the executor compared with the one-wavefront one: 4 (two-ended, four wavefronts) or 2")
    ap.add_argument("--retries", type=int, default=None, help="scvx_solver_opts.retries (default: the library's, 5)")
    a = ap.parse_args()
    from successiveconvexification_amd import montecarlo as mc, sample_problems as sp
    from successiveconvexification_amd.batch import ScvxBatch
    from successiveconvexification_amd.dynamics import IntegratorCache
    rng = np.random.default_rng(a.seed)
    base = sp.base_prob_fin_scaled() if a.fins else sp.base_prob_scaled
    print("| # | K | status per executor (all steps) | steps with different statuses | IPM its (1 / 4 wavefronts) | max final x difference (same-status trajectories) |")
    print("|---|---|---|---|---|---|")
    worst = 0.0
    ndiff = 0
    att = {"1": 0, a.waves: 0}; fail = {"1": 0, a.waves: 0}
    for n in range(a.n):
        K = int(rng.choice([12, 25, 31, 50, 64]))
        p = replace(base, K=K, mdry=float(base.mwet * rng.uniform(0.4, 0.999)), Tmin=float(base.Tmax * rng.uniform(0.05, 0.6)),
                    deltaMax=float(rng.uniform(5.0, 30.0)), thetaMax=float(rng.uniform(30.0, 120.0)), gammaGs=float(rng.uniform(5.0, 45.0)),
                    omMax=float(rng.uniform(20.0, 120.0)), tf_guess=float(rng.uniform(0.5, 12.0)))
        if a.fins:
            p = replace(p, finmxf=float(rng.uniform(0.002, 0.02)))
        ic = mc.disperse_ics(p, 0, a.B, 500 + n, frac=0.3)
        res = {}
        for waves in ("1", a.waves):
            os.environ["SCVX_K4_WAVES"] = waves
            c = IntegratorCache(p, npts=4)
            b = ScvxBatch(c, a.B, retries=a.retries).init(ic)
            sts, its = [], []
            for _ in range(a.steps):
                b.solve_step()
                s, it, m, _ = b.solver_stats()
                sts.append(s.copy()); its.append(it.copy())
            res[waves] = (np.array(sts), np.array(its), b.trajectory()[0])
            S = np.array(sts); bad = S != 0                       # solver status per step: a failed trajectory is frozen afterwards
            alive = np.vstack([np.ones((1, S.shape[1]), bool), ~np.maximum.accumulate(bad, axis=0)[:-1]])
            feas = S[0] != 5
            att[waves] += int((alive & feas[None, :]).sum()); fail[waves] += int((alive & bad & feas[None, :]).sum())
            b.close(); c.close()
        s1, s4 = res["1"][0], res[a.waves][0]
        d = (s1 != s4).any(axis=1).sum()
        same = (s1 == s4).all(axis=0)
        dx = np.abs(res["1"][2][same] - res[a.waves][2][same]).max() if same.any() else float("nan")
        worst = max(worst, dx if dx == dx else 0.0)
        ndiff += int((s1 != s4).sum())
        cnt = lambda s: {int(k): int(v) for k, v in zip(*np.unique(s, return_counts=True))}
        print("| %d | %d | %s / %s | %d | %.1f / %.1f | %.1e |" % (n, K, cnt(s1), cnt(s4), d, res["1"][1].mean(), res[a.waves][1].mean(), dx), flush=True)
    os.environ.pop("SCVX_K4_WAVES", None)
    print("\nsolves with different status between the executors: %d; worst final-x difference where all statuses agree: %.2e" % (ndiff, worst))
    for w in ("1", a.waves):
        print("%s-wavefront executor: attempted solves on live trajectories %d, first failures %d (%.3f %%), optimal %.3f %%" % (
            w, att[w], fail[w], 100.0 * fail[w] / max(att[w], 1), 100.0 * (1 - fail[w] / max(att[w], 1))))


if __name__ == "__main__":
    main()
