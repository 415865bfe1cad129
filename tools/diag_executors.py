"""Where two executors of the conic solve part ways over a full solve_problem run (diagnostic).
    python tools/diag_executors.py [B]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from successiveconvexification_amd.batch import ScvxBatch
from successiveconvexification_amd.defns import DescentProblem
from successiveconvexification_amd.dynamics import IntegratorCache
from successiveconvexification_amd.montecarlo import disperse_ics

p = DescentProblem()
p.K, p.tf_guess, p.mdry, p.mwet, p.alpha, p.imax = 20, 6.0, 1.0, 2.0, 0.05, 12
p.rIi, p.vIi = np.array([4.0, 2.0, 0.0]), np.array([-0.5, -0.5, 0.3])
B = int(sys.argv[1]) if len(sys.argv) > 1 else 560
ic = disperse_ics(p, 0, B, 99, frac=0.3)
res = {}
for waves in ("1", "2", "4"):
    os.environ["SCVX_K4_WAVES"] = waves
    c = IntegratorCache(p, npts=4)
    b = ScvxBatch(c, B).init(ic)
    hist = []
    for step in range(p.imax - 1):
        st, nu, dj = b.solve_step()
        hist.append((np.array(st), np.array(nu), np.array(dj)))
    x, u, s = b.trajectory()
    res[waves] = (hist, x, u, s)
    b.close(); c.close()
for w in ("2", "4"):
    h1, hw = res["1"][0], res[w][0]
    first = np.full(B, -1)
    for n, ((s1, n1, d1), (sw, nw, dw)) in enumerate(zip(h1, hw)):
        d = (s1 != sw) & (first < 0)
        first[d] = n
    idx = np.where(first >= 0)[0]
    print("waves", w, "vs 1:", len(idx), "of", B, "trajectories part ways; final x diff of the rest %.2e" %
          np.abs(res["1"][1][first < 0] - res[w][1][first < 0]).max())
    for i in idx[:12]:
        n = first[i]
        print("  traj %d step %d: status %d vs %d  nu %.3e vs %.3e  dJ %.6e vs %.6e  | final x diff %.2e" % (
            i, n, h1[n][0][i], hw[n][0][i], h1[n][1][i], hw[n][1][i], h1[n][2][i], hw[n][2][i],
            np.abs(res["1"][1][i] - res[w][1][i]).max()))
