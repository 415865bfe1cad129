"""Rocketland.create_initial / solve_step / solve_problem restated (oracle; rules in oracle/__init__.py).

Follows rocketland.jl:34-39 (create_initial), :226-321 (solve_step), :432-443 (solve_problem) with
    discretisation  = oracle.dynamics (exact RK4 variational equations, npts substeps)
    conic solve     = oracle.ipm on oracle.socp.build
"""
from dataclasses import dataclass
import numpy as np

from . import dynamics as od
from . import ipm, socp
from .model import DescentProblem, linear_points


@dataclass
class Iterate:
    """ProblemIteration (master.jl:122-134)."""
    problem: DescentProblem
    par: od.Params
    sigma: float
    x: np.ndarray        # about: [K+1][14]
    u: np.ndarray        # about: [K+1][3]
    endpoint: np.ndarray  # dynam: [K][14]
    deriv: np.ndarray     # dynam: [K][21][14]
    iter: int
    rk: float
    cost: float
    nsub: int = 10
    last: dict = None


def create_initial(p: DescentProblem, nsub=10, rIi=None, vIi=None) -> Iterate:
    """rocketland.jl:34-39 + initial_solve.jl:131-135."""
    par = od.Params(p)
    x, u = linear_points(p, rIi, vIi)
    e, d = od.linearize(par, x[None], u[None], np.array([p.tf_guess]), 1.0 / (p.K + 1), nsub)
    if rIi is not None or vIi is not None:
        # a dispersed initial condition is a different DescentProblem: its boundary rows use it
        from dataclasses import replace
        p = replace(p, rIi=np.asarray(rIi if rIi is not None else p.rIi, float),
                    vIi=np.asarray(vIi if vIi is not None else p.vIi, float))
    return Iterate(p, par, p.tf_guess, x, u, e[0], d[0], 0, 100.0, np.inf, nsub)


def solve_socp(it: Iterate, tol=1e-9):
    p = it.problem
    c, A, b, G, h, l, q, ix = socp.build(p, it.x, it.u, it.endpoint, it.deriv, it.rk)
    sol = ipm.solve(c, A, b, G, h, l, q, tol=tol)
    return sol, ix


def solve_step(it: Iterate, tol=1e-9):
    """rocketland.jl:226-321.  Returns (Iterate, ||nu||, dJ)."""
    p = it.problem
    K = p.K
    sol, ix = solve_socp(it, tol)
    if sol.status != "optimal":  # rocketland.jl:273-276
        raise RuntimeError(f"Non-optimal result {sol.status} exiting")
    z = sol.x
    xr = z[ix.xv].T.copy()   # [K+1][14]
    ur = z[ix.uv].T.copy()
    dsr = float(z[ix.dsig])
    nur = z[ix.nuv].T.copy()
    dt = 1.0 / (K + 1)
    # jK (:289): nonlinear defect cost
    xn = od.propagate(it.par, xr[None], ur[None], np.array([it.sigma + dsr]), dt, it.nsub)[0]
    defect = xr[1:] - xn
    jK = -xr[K, 0] + p.wNu * np.linalg.norm(defect)
    lK = -xr[K, 0] + p.wNu * np.linalg.norm(nur)          # :290
    info = dict(sol=sol, xr=xr, ur=ur, dsr=dsr, nur=nur, jK=jK, lK=lK, Jtr=float(z[ix.Jtr]))
    if it.rk == np.inf:                                    # :292-293 (never taken: rk starts at 100)
        next_rk = p.ri
        djk = np.nan
    else:
        jKm = it.cost
        djk = jKm - jK
        dlk = jKm - lK
        with np.errstate(invalid="ignore"):
            rhk = djk / dlk
        info["rho"] = rhk
        if rhk < p.rh0:                                    # :299-301 reject
            nxt = Iterate(p, it.par, it.sigma, it.x, it.u, it.endpoint, it.deriv, it.iter + 1,
                          it.rk / p.alph, it.cost, it.nsub, info)
            return nxt, float(np.linalg.norm(nur)), np.inf
        elif rhk < p.rh1:
            next_rk = it.rk / p.alph
        elif p.rh1 <= rhk and rhk < p.rh2:
            next_rk = it.rk
        else:                                              # includes rho = NaN on the first call
            next_rk = p.bet * it.rk
    nsig = it.sigma + dsr
    e, d = od.linearize(it.par, xr[None], ur[None], np.array([nsig]), dt, it.nsub)   # :318
    nxt = Iterate(p, it.par, nsig, xr, ur, e[0], d[0], it.iter + 1, next_rk, jK, it.nsub, info)
    return nxt, float(np.linalg.norm(nur)), float(djk)


def solve_problem(p: DescentProblem, nsub=10, rIi=None, vIi=None, tol=1e-9, log=None):
    """rocketland.jl:432-443."""
    it = create_initial(p, nsub, rIi, vIi)
    cnu = np.inf
    cdel = np.inf
    n = 1
    while (p.nuTol < cnu or p.delTol < cdel) and n < p.imax:
        it, cnu, cdel = solve_step(it, tol)
        if log is not None:
            log.append(dict(iter=it.iter, cnu=cnu, cdel=cdel, rk=it.rk, sigma=it.sigma, cost=it.cost,
                            rho=it.last.get("rho"), ipm_iters=it.last["sol"].iters))
        n += 1
    return it, cnu, cdel
