"""Does the 3-DoF initialiser help the 6-DoF loop?  solve_problem from the straight-line guess (linear_points) and from
FirstRound.solve_initial (scvx_batch_init_threedof) on a flyable instance, B dispersed initial conditions.
    python tools/init_compare.py [--B 256] [--tf 6.0] [--imax 30]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=256)
    ap.add_argument("--K", type=int, default=30)
    ap.add_argument("--tf", type=float, default=6.0)
    ap.add_argument("--imax", type=int, default=30)
    a = ap.parse_args()
    from successiveconvexification_amd.batch import ScvxBatch
    from successiveconvexification_amd.defns import DescentProblem
    from successiveconvexification_amd.dynamics import IntegratorCache
    from successiveconvexification_amd.montecarlo import disperse_ics
    p = DescentProblem()
    p.K, p.tf_guess, p.mdry, p.mwet, p.alpha, p.imax = a.K, a.tf, 1.0, 2.0, 0.05, a.imax
    p.rIi, p.vIi = np.array([4.0, 2.0, 0.0]), np.array([-0.5, -0.5, 0.3])
    c = IntegratorCache(p)
    ic = disperse_ics(p, 0, a.B, 20261004)
    print("| start | converged | failed | solve_steps mean (max) of the converged | final sigma mean | final mass mean |")
    print("|---|---|---|---|---|---|")
    for name in ("straight line (linear_points)", "3-DoF optimum (solve_initial), attitude as the reference: e1 -> -T", "3-DoF optimum, attitude e1 -> +T (align_thrust)"):
        b = ScvxBatch(c, a.B)
        if name.startswith("3-DoF"):
            st3 = b.init_threedof(ic, align_thrust="+T" in name)
            assert np.all(st3 == 0), np.unique(st3, return_counts=True)
        else:
            b.init(ic)
        st, it, nu, dj = b.solve()
        x, u, s = b.trajectory()
        conv = st == 0
        fail = (st == 3) | (st == 4) | (st == 5)
        print("| %s | %.3f | %.3f | %.1f (%d) | %.3f | %.4f |" % (name, conv.mean(), fail.mean(), it[conv].mean() if conv.any() else float("nan"),
                                                               it[conv].max() if conv.any() else -1, s[conv].mean() if conv.any() else float("nan"),
                                                               x[conv, -1, 0].mean() if conv.any() else float("nan")), flush=True)
        b.close()


if __name__ == "__main__":
    main()
