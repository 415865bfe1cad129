"""Per-step L-inf of the B=1 full solve against the oracle's recorded run, for one or more prebuilt libraries."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from successiveconvexification_amd import _lib, sample_problems as sp
g = np.load(os.path.join(ROOT, "tests", "golden", "oracle_scvx_full.npz"))
for path in sys.argv[1:] or ["successiveconvexification_amd/libscvx_hip.so"]:
    _lib._LIB = None
    _lib.LIB_PATH = os.path.join(ROOT, path)
    from successiveconvexification_amd.batch import ScvxBatch
    from successiveconvexification_amd.dynamics import IntegratorCache
    c = IntegratorCache(sp.base_prob_scaled)
    b = ScvxBatch(c, 1).init(None)
    print(path)
    for n in range(len(g["log"])):
        b.solve_step()
        x, u, s = b.trajectory()
        d = np.abs(x[0] - g["xs"][n])
        i = np.unravel_index(d.argmax(), d.shape)
        st, its, merit, pobj = b.solver_stats()
        print("  step %2d  dx %.16e at %s  du %.3e  ipm its %d merit %.2e x[%s]=%.17g" % (n, d.max(), i, np.abs(u[0] - g["us"][n]).max(), its[0], merit[0], i, x[0][i]))
    b.close(); c.close()
