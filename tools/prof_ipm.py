import ctypes as C, numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from successiveconvexification_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), sys.argv[2] if len(sys.argv) > 2 else "variants/libscvx_hip_prof.so")
from successiveconvexification_amd import sample_problems as sp
from successiveconvexification_amd.batch import ScvxBatch
from successiveconvexification_amd.dynamics import IntegratorCache
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
c = IntegratorCache(sp.base_prob_scaled)
b = ScvxBatch(c, B).init(bench.disperse_ics(sp.base_prob_scaled, 0, B, 20261004))
b.socp_solve()
out = np.zeros(128)
_lib.lib().scvx_debug_ipm_prof.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
_lib.lib().scvx_debug_ipm_prof(b.handle, out.ctypes.data_as(C.POINTER(C.c_double)))
st, its, merit, pobj = b.solver_stats()
names = {2:"  k: prefetch issue + slot shifts",5:"  k: pivot tile",12:"  k: chol_inv14",13:"  k: Linv store",0:"S_solve",1:"E_apply",3:"Hb_inv",4:"node blocks",5:"S assembly",6:"chol loop",7:"border solves",8:"W_all",9:"residuals",10:"newton(total)",11:"scale_pass",15:"TOTAL",24:"  k: Hd/M/TBp build",25:"  k: N_k gemm+store+ct",26:"  k: Bp copy + tile swap",27:"  k: TA,TBm",28:"  k: So elements",29:"  k: Wb gemm",30:"  k: late r_k + plain",31:"  k: fwd subst + t store",16:"  S_solve: Linv pass",17:"  S_solve: fwd chain",18:"  S_solve: Linv' pass",19:"  S_solve: bwd chain",20:"Et_apply",21:"cone_map(+t)",22:"dir+corr_rhs passes",23:"update_pass"}
tot = out[15]
print("ipm iters traj0:", its[0], " total Mcycles(100MHz ticks?) %.1f" % (tot/1e6))
for k,v in names.items(): print("%-14s %10.0f  %5.1f%%" % (v, out[k], 100*out[k]/tot))

if out[32:].any():
    nf = max(int(its[0]) + 1, 1)
    print("factorisation pipeline, cycles per factorisation by wavefront (slot 24 = own stage, 26 = coupling tiles, 27 = middle node, 28 = barrier):")
    for wv in range(4):
        o = out[32 * wv:32 * wv + 32]
        if o[24:29].any():
            print("  wavefront %d: stage %8.0f  tiles %8.0f  middle %8.0f  barrier %8.0f   (per step of 26: %5.0f / %5.0f / - / %5.0f)" % (
                wv, o[24] / nf, o[26] / nf, o[27] / nf, o[28] / nf, o[24] / nf / 26, o[26] / nf / 26, o[28] / nf / 26))
