"""CPU-twin SCvx loop with solver statistics (no GPU): B dispersed trajectories x STEPS solve_steps through
oracle/scvx_port.cpp (the device solver core compiled for the host) + the C discretisation oracle.

    python tools/twin_stats.py [--B 256] [--steps 8] [--K 50] [--aero] [--seed 20261004] [--tol 1e-8] [--lib path.so]

Prints, per step and overall, the distribution of the returned merit and of the solver status — the numbers the
IPM step-rule constants are validated on."""
import argparse, ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle
from oracle import dynamics as od, model, port


def run(p, B, steps, seed, tol, nsub=10, verbose=True, max_iter=60, refine=6, f32=False, accept=1e-6, ret_traj=False, warm=False):
    ic = model.disperse_ics(p, B, seed)
    t0 = [time.perf_counter()]

    def on_step(s, r, rej):
        if verbose:
            m = r["merit"]; t1 = time.perf_counter()
            print("step %2d  its %.2f (max %d)  status %s  merit max %.2e p99 %.2e  <tol %.3f  rej %.2f  %.1f traj-iter/s"
                  % (s, r["iters"].mean(), r["iters"].max(), dict(zip(*np.unique(r["status"], return_counts=True))),
                     m.max(), np.quantile(m, 0.99), (m < tol).mean(), rej.mean(), B / (t1 - t0[0])), flush=True)
            t0[0] = t1
    o = port.scvx_steps(p, ic, steps, nsub=nsub, tol=tol, accept=max(accept, tol), max_iter=max_iter, refine=refine, f32=f32,
                        on_step=on_step, warm_start=warm)
    m = np.concatenate(o["merit"]); st = np.concatenate(o["status"]); it = np.concatenate(o["iters"])
    if ret_traj:
        return m, st, it, (o["x"], o["u"], o["sigma"], o["rk"])
    return m, st, it


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=256); ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--K", type=int, default=50); ap.add_argument("--aero", action="store_true")
    ap.add_argument("--seed", type=int, default=20261004); ap.add_argument("--tol", type=float, default=1e-8)
    ap.add_argument("--lib", default=None); ap.add_argument("--refine", type=int, default=6)
    ap.add_argument("--max-iter", type=int, default=60); ap.add_argument("--f32", action="store_true")
    ap.add_argument("--accept", type=float, default=1e-6); ap.add_argument("--warm", action="store_true")
    a = ap.parse_args()
    if a.lib:
        oracle._PORT = ctypes.CDLL(os.path.abspath(a.lib))
    if a.aero:
        z = np.load(os.path.join(ROOT, "tests", "golden", "lift_drag_tables.npz"))
        p = model.base_prob_scaled(model.AeroData(z["drag"], z["lift"], z["torque"]))
    else:
        p = model.base_prob_scaled()
    if a.K != p.K:
        from dataclasses import replace
        p = replace(p, K=a.K)
    m, st, it = run(p, a.B, a.steps, a.seed, a.tol, max_iter=a.max_iter, refine=a.refine, f32=a.f32, accept=a.accept, warm=a.warm)
    print("ALL  solves %d  its %.2f  status %s  merit max %.2e p99.9 %.2e p99 %.2e  frac<tol %.4f  frac<1e-7 %.4f"
          % (m.size, it.mean(), dict(zip(*np.unique(st, return_counts=True))), m.max(), np.quantile(m, 0.999),
             np.quantile(m, 0.99), (m < a.tol).mean(), (m < 1e-7).mean()))
