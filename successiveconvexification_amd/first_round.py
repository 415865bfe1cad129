"""FirstRound — initial_solve.jl of the reference: the straight-line initial guess (:113-135) and the 3-DoF
lossless-convexification initialiser `solve_initial` (:17-110, inside a block comment at HEAD), which runs on the
device as a batched conic solve (csrc/scvx_threedof.hip)."""
import ctypes as C

import numpy as np

from .defns import DescentProblem, LinPoint
from . import dynamics


def rotation_between(a, b):
    """Rotations.rotation_between (third-party) as a scalar-first unit quaternion."""
    a = np.asarray(a, float)
    b = np.asarray(b, float)
    w = np.sqrt(a.dot(a) * b.dot(b)) + a.dot(b)
    if abs(w) < 100 * np.finfo(float).eps:
        e = np.zeros(3)
        e[int(np.argmin(np.abs(a)))] = 1.0
        v = np.cross(a, e)
    else:
        v = np.cross(a, b)
    q = np.array([w, v[0], v[1], v[2]])
    return q / np.linalg.norm(q)


def linear_points(problem: DescentProblem):
    """initial_solve.jl:113-129 -> K+1 LinPoints."""
    K = problem.K
    pts = []
    for k in range(K + 1):
        mk = (K - k) / K * problem.mwet + (k / K) * problem.mdry
        rIk = (K - k) / K * problem.rIi + (k / K) * problem.rIf
        vIk = (K - k) / K * problem.vIi + (k / K) * problem.vIf
        q = rotation_between([1, 0, 0], -vIk)
        ctrl = np.zeros(getattr(problem, "nu", 3))   # fin controls (control_dim = 5) start at zero
        ctrl[0] = mk * problem.g
        pts.append(LinPoint(np.concatenate([[mk], rIk, vIk, q, [0.0, 0, 0]]), ctrl))
    return pts


def linear_initial(problem: DescentProblem, cache):
    """initial_solve.jl:131-135"""
    pts = linear_points(problem)
    return pts, dynamics.linearize_dynamics(pts, problem.tf_guess, 1 / (problem.K + 1), cache)


THREEDOF_STATUS = {0: "optimal", 1: "iteration cap", 2: "stalled", 3: "non-finite", 4: "almost optimal", 5: "infeasible"}


def threedof_opts(L, tol=None, max_iter=None, refine=None, delta=None, align_thrust=None):
    from . import _lib
    o = _lib.ScvxThreedofOpts()
    L.scvx_threedof_default_opts(C.byref(o))
    if tol is not None:
        o.tol = tol
    if max_iter is not None:
        o.max_iter = max_iter
    if refine is not None:
        o.refine = refine
    if delta is not None:
        o.delta = delta
    if align_thrust is not None:   # attitude of the 6-DoF start: False = the reference's rotation_between(e1, -T), True = +T
        o.attitude = 1 if align_thrust else 0
    return o


def solve_initial_batch(cache, ic=None, B=None, **opts):
    """The 3-DoF landing SOCP of initial_solve.jl:17-88 for B initial conditions ic [B][6] = (rIi, vIi) (None: the
    problem's own, B of them).  Returns (sol, status, info): sol = dict of arrays with a leading batch axis in the
    reference's variable shapes -- T, r, v, ar [B][3][K+1]; ma, ga, kaR [B][K+1]; nkaR [B] -- status [B] (0 = optimal,
    THREEDOF_STATUS), info [B][5] = iterations, objective, gap, primal and dual residual."""
    from . import _lib
    L = cache._L
    K = cache.problem.K
    if ic is not None:
        ic = np.ascontiguousarray(ic, np.float64)
        if ic.ndim != 2 or ic.shape[1] != 6:
            raise ValueError("ic must be [B][6] = (rIi, vIi)")
        B = ic.shape[0]
    B = 1 if B is None else int(B)
    n = L.scvx_threedof_record_doubles(K)
    rec = np.zeros((B, n))
    status = np.zeros(B, np.int32)
    info = np.zeros((B, 5))
    o = threedof_opts(L, **opts)
    dp = C.POINTER(C.c_double)
    rc = L.scvx_threedof_solve(cache.handle, B, ic.ctypes.data_as(dp) if ic is not None else None, C.byref(o),
                               rec.ctypes.data_as(dp), status.ctypes.data_as(C.POINTER(C.c_int32)), info.ctypes.data_as(dp))
    _lib.check(cache.handle, rc, "scvx_threedof_solve")
    nodes = rec[:, :-1].reshape(B, K + 1, 15)
    tr = lambda a: np.ascontiguousarray(np.swapaxes(a, 1, 2))
    sol = dict(r=tr(nodes[:, :, 0:3]), v=tr(nodes[:, :, 3:6]), ma=nodes[:, :, 6].copy(), T=tr(nodes[:, :, 7:10]),
               ga=nodes[:, :, 10].copy(), kaR=nodes[:, :, 11].copy(), ar=tr(nodes[:, :, 12:15]), nkaR=rec[:, -1].copy())
    return sol, status, info


def solve_initial(problem: DescentProblem, cache, **opts):
    """initial_solve.jl:17-110: (initial_points, linearisation) from the 3-DoF optimum -- state (ma_k, r_k, v_k,
    rotation_between(e1, -T_k), 0), control (|T_k|, 0, 0) (:90-105), linearised at tf_guess (:107).  Raises if the conic
    solve is not optimal (the reference's Mosek call has no fallback either)."""
    sol, status, _ = solve_initial_batch(cache, np.concatenate([problem.rIi, problem.vIi])[None, :], **opts)
    if status[0] != 0:
        raise RuntimeError("3-DoF initial solve: " + THREEDOF_STATUS.get(int(status[0]), str(int(status[0]))))
    pts = []
    for k in range(problem.K + 1):
        Tk = sol["T"][0, :, k]
        q = rotation_between([1, 0, 0], -Tk)
        state = np.concatenate([[sol["ma"][0, k]], sol["r"][0, :, k], sol["v"][0, :, k], q, [0.0, 0, 0]])
        ctrl = np.zeros(getattr(problem, "nu", 3))
        ctrl[0] = np.linalg.norm(Tk)
        pts.append(LinPoint(state, ctrl))
    return pts, dynamics.linearize_dynamics(pts, problem.tf_guess, 1 / (problem.K + 1), cache)
