"""CPU twin on the bench mix: B dispersed trajectories x 14 solve_steps with the warm start (what the device runs), printing iterations per
solve, failures, and a checksum of the final iterate -- the before/after figure for changes to scvx_ipm_core.hpp.
    python tools/twin_mix.py [B=32] [save.npz | cmp.npz]   (SCVX_TWIN_CLASSES=n: also n random problem classes of tools/k4_fuzz.py)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle
oracle.build()
from oracle import model, port
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
p = model.base_prob_scaled()
ic = model.disperse_ics(p, B, 20261004)
t0 = time.time()
o = port.scvx_steps(p, ic, p.imax - 1, warm_start=True)
its = np.array(o["iters"]); st = np.array(o["status"]); rej = np.array(o["rejected"])
print("twin mix B=%d: %.2f its/solve, failed %d, rejected %.3f, merit max %.2e, %.1fs" % (B, its.mean(), int(((st != 0) & (st != 4)).sum()), rej.mean(), np.max(o["merit"]), time.time() - t0))
print("per-step mean its:", np.round(its.mean(1), 2))
if len(sys.argv) > 2:
    f = sys.argv[2]
    if os.path.exists(f):
        g = np.load(f)
        print("vs %s: |dx| %.3e |du| %.3e |dsigma| %.3e  its %.2f -> %.2f  same rejections %s" % (f, np.abs(o["x"] - g["x"]).max(), np.abs(o["u"] - g["u"]).max(),
              np.abs(o["sigma"] - g["sigma"]).max(), g["iters"].mean(), its.mean(), np.array_equal(rej, g["rejected"])))
    else:
        np.savez(f, x=o["x"], u=o["u"], sigma=o["sigma"], iters=its, rejected=rej)
        print("saved", f)
