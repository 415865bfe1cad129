"""Rocketland — the reference's SCvx driver API (rocketland.jl), one trajectory at a time, over the HIP path.

    create_initial(problem, cache) -> ProblemIteration            rocketland.jl:34-39
    run_iters(iprob, niters, cache) -> (trajectories, tfs)        rocketland.jl:420-430
    plot_solution_data(ip) / dump_solution(ip, path)              rocketland.jl:454-478 (the arrays plot_solution draws)
    solve_step(iteration, cache) -> (ProblemIteration, |nu|, dJ)  rocketland.jl:226-321
    solve_problem(iprob, cache) -> (ProblemIteration, cnu, cdel)  rocketland.jl:432-443
The recipe of rocketland.jl:26-32 reads the same here:
    cache = IntegratorCache(prob, ProbInfo.from_problem(prob), make_dynamics_module(...))
    pi = create_initial(prob, cache); pi, cnu, cdel = solve_step(pi, cache)
`ProblemIteration.model` holds the device batch (B = 1) instead of MOI handles; everything numeric runs in
libscvx_hip.so (batch.py is the batched form the GPU exists for).
"""
import numpy as np

from .batch import ScvxBatch
from .defns import DescentProblem, LinPoint, LinRes, ProblemIteration
from .dynamics import IntegratorCache


def _snapshot(problem, cache, batch) -> ProblemIteration:
    x, u, s = batch.trajectory()
    e, d = batch.linearization()
    rk, cost, it = batch.scalars()
    K = problem.K
    about = [LinPoint(x[0, k].copy(), u[0, k].copy()) for k in range(K + 1)]
    dynam = [LinRes(e[0, k].copy(), d[0, k].T.copy()) for k in range(K)]
    return ProblemIteration(problem, cache, float(s[0]), about, dynam, batch, int(it[0]), float(rk[0]), float(cost[0]))


def create_initial(problem: DescentProblem, linear_cache: IntegratorCache) -> ProblemIteration:
    batch = ScvxBatch(linear_cache, 1).init(None)
    return _snapshot(problem, linear_cache, batch)


def solve_step(iteration: ProblemIteration, linear_cache: IntegratorCache):
    batch = iteration.model
    st, nu, dj = batch.solve_step()
    if st[0] in (3, 4, 5):  # rocketland.jl:273-276
        raise RuntimeError("Non-optimal result %s exiting" % {3: "SLOW_PROGRESS", 4: "NUMERICAL_ERROR", 5: "INFEASIBLE"}[int(st[0])])
    return _snapshot(iteration.problem, linear_cache, batch), float(nu[0]), float(dj[0])


def solve_problem(iprob: DescentProblem, cache: IntegratorCache):
    prob = create_initial(iprob, cache)
    cnu = np.inf
    cdel = np.inf
    it = 1
    while (iprob.nuTol < cnu or iprob.delTol < cdel) and it < iprob.imax:
        prob, cnu, cdel = solve_step(prob, cache)
        it += 1
    return prob, cnu, cdel


def run_iters(iprob: DescentProblem, niters: int, cache: IntegratorCache = None):
    """rocketland.jl:420-430: niters solve_steps, keeping the position history r[3][K+1] and sigma of every iterate
    (the reference's version calls 1-argument create_initial/solve_step that do not exist at HEAD; the cache is
    explicit here)."""
    cache = cache if cache is not None else IntegratorCache(iprob)
    ip = create_initial(iprob, cache)
    trjs, tfs = [], []
    for _ in range(niters):
        ip, _, _ = solve_step(ip, cache)
        tfs.append(ip.sigma)
        trjs.append(np.stack([pt.state[1:4] for pt in ip.about], axis=1))
    return trjs, tfs


# ---- plot_solution (rocketland.jl:454-478) without a plotting library ------------------------------------------------
def body_axis(q):
    """Dynamics.DCM(q) * [1, 0, 0] (dynamics.jl:29-44): first column of the body->inertial rotation, q scalar-first."""
    q0, q1, q2, q3 = (np.asarray(q, float)[..., i] for i in range(4))
    return np.stack([1 - 2 * (q2 * q2 + q3 * q3), 2 * (q1 * q2 + q0 * q3), 2 * (q1 * q3 - q0 * q2)], axis=-1)


def plot_solution_data(ip: ProblemIteration) -> dict:
    """Everything Rocketland.plot_solution computes before it calls Plots.jl, under the names it uses:
        xs [K+1][2] = (r_y, r_z), ys [K+1][2] = (r_up, r_up)      the two trajectory panels (layout = 2)
        xlims = (tmin, tmax), pmin, pmax                           its axis limits (tmin/tmax over r_up and r_y, pmin/pmax over r_up and r_z)
        thr [K+1] = |u| / Tmax                                      the throttle it prints
        dp [K+1] = (C(q) e1) . v / |v|                              the cos(angle of attack) it prints per node
        xls, yls [K+1][2][2]                                        the attitude tick of every node: from r to r + C(q) e1 / 3
    """
    X = np.stack([pt.state for pt in ip.about])
    U = np.stack([pt.control for pt in ip.about])
    up, ry, rz = X[:, 1], X[:, 2], X[:, 3]
    dv = body_axis(X[:, 7:11])
    v = X[:, 4:7]
    with np.errstate(invalid="ignore", divide="ignore"):
        dp = np.sum(dv * v, axis=1) / np.linalg.norm(v, axis=1)
    xls = np.stack([np.stack([ry, rz], 1), np.stack([ry + dv[:, 1] / 3, rz + dv[:, 2] / 3], 1)], axis=1)
    yls = np.stack([np.stack([up, up], 1), np.stack([up + dv[:, 0] / 3, up + dv[:, 0] / 3], 1)], axis=1)
    return dict(xs=np.stack([ry, rz], 1), ys=np.stack([up, up], 1),
                xlims=(float(min(up.min(), ry.min())), float(max(up.max(), ry.max()))),
                pmin=float(min(up.min(), rz.min())), pmax=float(max(up.max(), rz.max())),
                thr=np.linalg.norm(U, axis=1) / ip.problem.Tmax, dp=dp, xls=xls, yls=yls, sigma=float(ip.sigma))


def dump_solution(ip: ProblemIteration, path: str) -> str:
    """Writes plot_solution_data(ip) to `path`: .npz (all arrays) or .csv (one row per node: k, r_up, r_y, r_z, thr, dp and
    the attitude tick end points) -- enough to redraw the reference's figure with any tool."""
    d = plot_solution_data(ip)
    if path.endswith(".npz"):
        np.savez(path, **{k: np.asarray(v) for k, v in d.items()})
    else:
        K1 = d["xs"].shape[0]
        tab = np.column_stack([np.arange(K1), d["ys"][:, 0], d["xs"][:, 0], d["xs"][:, 1], d["thr"], d["dp"],
                               d["yls"][:, 1, 0], d["xls"][:, 1, 0], d["xls"][:, 1, 1]])
        np.savetxt(path, tab, delimiter=",", comments="",
                   header="k,r_up,r_y,r_z,thr,cos_aoa,tick_up,tick_y,tick_z   # sigma=%.17g xlims=%s" % (d["sigma"], d["xlims"]))
    return path
