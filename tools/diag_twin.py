import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dataclasses import replace
from oracle import model, port
from successiveconvexification_amd import sample_problems as sp
from successiveconvexification_amd.batch import ScvxBatch
from successiveconvexification_amd.dynamics import IntegratorCache
for K in (30, 50, 100):
    po = replace(model.base_prob_scaled(), K=K); pp = replace(sp.base_prob_scaled, K=K)
    B = 3; ic = model.disperse_ics(po, B, 20261005)
    for tol in (1e-8, 1e-9):
        b = ScvxBatch(IntegratorCache(pp, npts=4), B, tol=tol).init(ic)
        xb, ub, sg = b.trajectory(); e, d = b.linearization()
        x, u, s, nu = b.socp_solve(); st, its, merit, pobj = b.solver_stats()
        tw = port.socp(po, xb, ub, e, d, 100.0, ic, tol=tol)
        def obj(dx, du, ds, nu): return -dx[:, K, 0] + po.wNu*np.sqrt((nu**2).sum((1,2))) + 0.5*np.sqrt((dx**2).sum((1,2))+(du**2).sum((1,2))) + abs(ds)
        og = obj(x-xb, u-ub, s-sg, nu); ot = obj(tw["dx"], tw["du"], tw["ds"], tw["nu"])
        print("K", K, "tol", tol, "gpu its", its, "merit", merit, "| twin its", tw["iters"], "merit", tw["merit"])
        print("    dx err %.2e du err %.2e ds err %.2e | obj gpu-twin rel %s" % (np.abs(x-xb-tw["dx"]).max(), np.abs(u-ub-tw["du"]).max(), np.abs(s-sg-tw["ds"]).max(), (og-ot)/np.abs(ot)))
