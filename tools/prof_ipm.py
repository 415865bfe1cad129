import ctypes as C, numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from successiveconvexification_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "variants", "libscvx_hip_prof.so")
from successiveconvexification_amd import sample_problems as sp
from successiveconvexification_amd.batch import ScvxBatch
from successiveconvexification_amd.dynamics import IntegratorCache
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
c = IntegratorCache(sp.base_prob_scaled)
b = ScvxBatch(c, B).init(bench.disperse_ics(sp.base_prob_scaled, 0, B, 20261004))
b.socp_solve()
out = np.zeros(64)
_lib.lib().scvx_debug_ipm_prof.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
_lib.lib().scvx_debug_ipm_prof(b.handle, out.ctypes.data_as(C.POINTER(C.c_double)))
st, its, merit, pobj = b.solver_stats()
names = {2:"  k: stage+TBp",5:"  k: pivot tile",12:"  k: chol_inv14",13:"  k: Linv/Nf store",14:"  k: swap+TA+So+Wb+Nb",0:"S_solve",1:"E_apply",3:"Hb_inv",4:"node blocks",5:"S assembly",6:"chol loop",7:"border solves",8:"W_all",9:"residuals",10:"newton(total)",11:"scale_pass",15:"TOTAL",16:"  S_solve: Linv pass",17:"  S_solve: fwd chain",18:"  S_solve: Linv' pass",19:"  S_solve: bwd chain",20:"Et_apply",21:"cone_map(+t)",22:"dir+corr_rhs passes",23:"update_pass"}
tot = out[15]
print("ipm iters traj0:", its[0], " total Mcycles(100MHz ticks?) %.1f" % (tot/1e6))
for k,v in names.items(): print("%-14s %10.0f  %5.1f%%" % (v, out[k], 100*out[k]/tot))

if out[32:].any():
    print("two-wavefront factorisation pipeline (cycles per segment-step of one factorisation; wavefront 0 = chain, 1 = assembly):")
    nf = max(int(its[0]) + 1, 1) * (50 + 1)
    for k, (a, b) in {24: ("copy + pivot update", "stage node + TBp"), 25: ("chol_inv14", "Sd gemm"), 26: ("Linv store, Nf", "TA, TBm, So of k+1"), 27: ("Wb gemm", "-"), 28: ("barrier wait", "barrier wait")}.items():
        print("  slot %d  w0 %-22s %8.0f   w1 %-22s %8.0f" % (k, a, out[k] / nf, b, out[32 + k] / nf))
