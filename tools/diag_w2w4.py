import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from dataclasses import replace
from successiveconvexification_amd import _lib
if len(sys.argv) > 1: _lib.LIB_PATH = os.path.join(ROOT, sys.argv[1])
import k4_fuzz
from oracle import model
from successiveconvexification_amd import sample_problems as sp
from successiveconvexification_amd.batch import ScvxBatch
from successiveconvexification_amd.dynamics import IntegratorCache
rng = np.random.default_rng(1)
base = model.base_prob_scaled()
for _ in range(40):
    po = k4_fuzz.draw_class(rng, base)
pp = replace(sp.base_prob_scaled, K=po.K, mdry=po.mdry, Tmin=po.Tmin, deltaMax=po.deltaMax, thetaMax=po.thetaMax,
             gammaGs=po.gammaGs, omMax=po.omMax, tf_guess=po.tf_guess, model_flags=sp.base_prob_scaled.model_flags | 1)
ic = model.disperse_ics(po, 16, 539, 0.3)
c = IntegratorCache(pp, npts=4)
res = {}
for W in ("2", "4"):
    os.environ["SCVX_K4_WAVES"] = W
    for tol in (1e-8, 1e-10):
        b = ScvxBatch(c, 16, tol=tol).init(ic)
        out = []
        for n in range(2):
            st, nun, dj = b.solve_step()
            sst, sit, merit, pobj = b.solver_stats()
            out.append((b.trajectory()[0].copy(), sst.copy(), sit.copy(), merit.copy(), pobj.copy()))
        res[(W, tol)] = out
        b.close()
for n in range(2):
    a, bb = res[("2", 1e-8)][n], res[("4", 1e-8)][n]
    r2, r4 = res[("2", 1e-10)][n], res[("4", 1e-10)][n]
    print("step", n + 1)
    print("  |x(W2) - x(W4)| tol 1e-8 per traj:", np.array2string(np.abs(a[0] - bb[0]).max((1, 2)), precision=1))
    print("  |x(W2,1e-8) - x(W2,1e-10)|       :", np.array2string(np.abs(a[0] - r2[0]).max((1, 2)), precision=1))
    print("  |x(W4,1e-8) - x(W4,1e-10)|       :", np.array2string(np.abs(bb[0] - r4[0]).max((1, 2)), precision=1))
    print("  |x(W2,1e-10) - x(W4,1e-10)|      :", np.array2string(np.abs(r2[0] - r4[0]).max((1, 2)), precision=1))
    print("  iters W2", a[2].tolist(), "\n  iters W4", bb[2].tolist())
    print("  pobj rel diff W2 vs W4:", np.array2string(np.abs(a[4] / bb[4] - 1), precision=1))
