"""Build first:  mkdir -p variants && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSCVX_TD_PROF -I include \
    -I successiveconvexification_amd/csrc -o variants/libscvx_hip_tdprof.so successiveconvexification_amd/csrc/*.hip
In-kernel section shares of the 3-DoF initialiser (K0) for trajectory 0, from a diagnostic build
(variants/libscvx_hip_tdprof.so: -DSCVX_TD_PROF, s_memtime around the sections of scvx_threedof_core.hpp).
    python tools/prof_threedof.py [B]"""
import ctypes as C
import os
import sys
from dataclasses import replace

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from successiveconvexification_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "variants", "libscvx_hip_tdprof.so")
from oracle import model
from successiveconvexification_amd import first_round
from successiveconvexification_amd.defns import DescentProblem
from successiveconvexification_amd.dynamics import IntegratorCache

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
po = replace(model.DescentProblem(), K=30, tf_guess=6.0, rIi=np.array([4.0, 2.0, 0.0]), vIi=np.array([-0.5, -0.5, 0.3]),
             mdry=1.0, mwet=2.0, alpha=0.05)
p = DescentProblem()
for f in ("g", "mdry", "mwet", "Tmin", "Tmax", "thetaMax", "gammaGs", "alpha", "K", "tf_guess"):
    setattr(p, f, getattr(po, f))
p.rIi, p.vIi = po.rIi.copy(), po.vIi.copy()
c = IntegratorCache(p)
ic = model.disperse_ics(po, B, 20261004)
first_round.solve_initial_batch(c, ic)
sol, st, info = first_round.solve_initial_batch(c, ic)
out = np.zeros(16)
L = _lib.lib()
L.scvx_debug_td_prof.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
assert L.scvx_debug_td_prof(c.handle, out.ctypes.data_as(C.POINTER(C.c_double))) == 0
names = {0: "scale: NT + lam", 1: "scale: H blocks", 2: "factor", 3: "border solve (Y)", 4: "condensed: W^-2 bz, rhs", 5: "condensed: band sweeps",
         6: "condensed: border, E du, W^-2", 7: "refinement residual", 8: "residuals + stop test", 9: "step length + corrector rhs",
         10: "step length + update", 15: "TOTAL"}
tot = out[15]
print("B = %d, trajectory 0: %d iterations, %.2f M ticks of s_memtime (100 MHz) = %.2f ms" % (B, int(info[0, 0]), tot / 1e6, tot / 1e5))
print("| section | share |")
print("|---|---|")
for k, v in names.items():
    print("| %s | %.1f %% |" % (v, 100 * out[k] / tot))
print("| (unaccounted: initial point, loop control) | %.1f %% |" % (100 * (tot - sum(out[k] for k in names if k != 15)) / tot))
