"""Twin (CPU) search for conic solves that do not reach `tol` on the bench workload: B dispersed trajectories x `steps`
solve_steps; prints (step, trajectory, status, iterations, merit) of every solve whose status is not 0 and saves the
subproblem data of the first few (x, u, endpoint, deriv, rk, ic) to an .npz for replay with SCVX_IPM_DEBUG."""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=8192)
ap.add_argument("--steps", type=int, default=14)
ap.add_argument("--seed", type=int, default=20261004)
ap.add_argument("--threads", type=int, default=8)
ap.add_argument("--out", default="/tmp/stalls.npz")
a = ap.parse_args()
from oracle import model, port, dynamics as od
po = model.base_prob_scaled()
ic = model.disperse_ics(po, a.B, a.seed)
found = []

def on_step(s, r, rej):
    bad = np.nonzero(r["status"] != 0)[0]
    for b in bad:
        print("step %d traj %d status %d iters %d merit %.3e" % (s, b, r["status"][b], r["iters"][b], r["merit"][b]), flush=True)
        found.append((s, int(b)))

o = port.scvx_steps(po, ic, a.steps, nthreads=a.threads, warm_start=True, on_step=on_step)
print("solves %d, non-optimal %d" % (a.B * a.steps, len(found)))
np.save(a.out.replace(".npz", "_list.npy"), np.array(found))
