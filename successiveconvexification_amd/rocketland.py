"""Rocketland — the reference's SCvx driver API (rocketland.jl), one trajectory at a time, over the HIP path.

    create_initial(problem, cache) -> ProblemIteration            rocketland.jl:34-39
    run_iters(iprob, niters, cache) -> (trajectories, tfs)        rocketland.jl:420-430
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
    if st[0] in (3, 4):  # rocketland.jl:273-276
        raise RuntimeError(f"Non-optimal result {'NUMERICAL_ERROR' if st[0] == 4 else 'SLOW_PROGRESS'} exiting")
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
