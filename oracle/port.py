"""ctypes front-end of oracle/scvx_port.cpp — the CPU twin of the device conic solver
(oracle; rules in oracle/__init__.py).  Consts mirrors scvx::ipm::Consts of scvx_ipm_core.hpp."""
import ctypes as C
import os
import numpy as np

from . import port_lib
from .model import DescentProblem

_dp = C.POINTER(C.c_double)


class Consts(C.Structure):
    _fields_ = [("K", C.c_int), ("max_iter", C.c_int), ("refine", C.c_int), ("pad", C.c_int),
                ("warm", C.c_int), ("retries", C.c_int),
                ("tol", C.c_double), ("accept", C.c_double),
                ("itan", C.c_double), ("sqcm", C.c_double), ("icos", C.c_double), ("Tmax", C.c_double),
                ("Tmin", C.c_double), ("omMax", C.c_double), ("mdry", C.c_double), ("wNu", C.c_double),
                ("mwet", C.c_double), ("finmxf", C.c_double), ("vmax", C.c_double),
                ("rIf", C.c_double * 3), ("vIf", C.c_double * 3), ("qBIf", C.c_double * 4),
                ("wBi", C.c_double * 3), ("wBf", C.c_double * 3)]


def consts(p: DescentProblem, tol=1e-8, max_iter=60, refine=6, accept=0.0, retries=None) -> Consts:
    c = Consts()
    # the solver's ladder of step rules (scvx_solver_opts.retries, default 5); SCVX_PORT_RETRIES overrides the default (tools)
    c.retries = int(os.environ.get("SCVX_PORT_RETRIES", "5")) if retries is None else int(retries)
    c.K, c.max_iter, c.refine, c.tol, c.accept = p.K, max_iter, refine, tol, max(accept, tol)
    c.itan = 1.0 / np.tan(np.radians(p.gammaGs))       # rocketland.jl:63
    c.sqcm = np.sqrt((1 - np.cos(np.radians(p.thetaMax))) / 2)  # :64
    c.icos = 1.0 / np.cos(np.radians(p.deltaMax))      # :65
    c.Tmax, c.Tmin, c.omMax, c.mdry, c.wNu, c.mwet = p.Tmax, p.Tmin, p.omMax, p.mdry, p.wNu, p.mwet
    c.vmax = float(np.sqrt(2.0 * p.dpMax / p.rho)) if getattr(p, "enforce_dp", False) else 0.0
    c.finmxf = float(getattr(p, "finmxf", 0.0))
    c.rIf[:] = list(p.rIf); c.vIf[:] = list(p.vIf); c.qBIf[:] = list(p.qBIf)
    c.wBi[:] = list(p.wBi); c.wBf[:] = list(p.wBf)
    return c


def _p(a):
    return a.ctypes.data_as(_dp)


def socp(p: DescentProblem, xbar, ubar, endpoint, deriv, rk, ic=None, tol=1e-8, max_iter=60, refine=6, nthreads=0, accept=0.0,
         f32=False, work=None, warm=None, lin32=False, retries=None):
    """Batched: xbar [B][K+1][14], ubar [B][K+1][nu], endpoint [B][K][14], deriv [B][K][14+2nu+1][14], rk [B]; nu = 3, or 5
    when p.fins (fin extension).  Returns dict(dx, du, ds, nu, status, iters, merit, pobj)."""
    xbar = np.ascontiguousarray(xbar, float)
    ubar = np.ascontiguousarray(ubar, float)
    endpoint = np.ascontiguousarray(endpoint, float)
    deriv = np.ascontiguousarray(deriv, float)
    B, K1, _ = xbar.shape
    K = K1 - 1
    rk = np.ascontiguousarray(np.broadcast_to(np.asarray(rk, float), (B,)))
    if ic is None:
        ic = np.tile(np.concatenate([p.rIi, p.vIi]), (B, 1))
    ic = np.ascontiguousarray(ic, float)
    c = consts(p, tol, max_iter, refine, accept, retries)
    NU = 5 if getattr(p, "fins", False) else 3
    assert ubar.shape[-1] == NU and deriv.shape[-2] == 14 + 2 * NU + 1, (ubar.shape, deriv.shape)
    sol = np.zeros((B, (K + 1) * (14 + NU) + 1))
    nu = np.zeros((B, K, 14))
    info = np.zeros((B, 4))
    if NU == 5:
        assert not (f32 or lin32), "the fin twin is built for double storage"
        wf = np.ascontiguousarray(warm if warm is not None else np.zeros(B), np.int32)
        port_lib().scvx_port_socp_fin(C.byref(c), C.c_int(B), _p(xbar), _p(ubar), _p(endpoint), _p(deriv), _p(rk), _p(ic), _p(sol), _p(nu),
                                      _p(info), C.c_int(nthreads), _p(work) if work is not None else None,
                                      wf.ctypes.data_as(C.POINTER(C.c_int32)))
    elif work is not None:   # persistent per-trajectory workspace + warm flags (the device's warm start after a rejected step)
        wf = np.ascontiguousarray(warm if warm is not None else np.zeros(B), np.int32)
        # SCVX_PORT_FAC32 = 1 / 0 (tools): the factor of the Schur complement forced to float / double (default: the build's SCVX_FACTOR_T)
        fac = os.environ.get("SCVX_PORT_FAC32")
        (port_lib().scvx_port_socp_fac32 if fac == "1" else port_lib().scvx_port_socp_fac64 if fac == "0" else port_lib().scvx_port_socp_ws)(C.byref(c), C.c_int(B), _p(xbar), _p(ubar), _p(endpoint), _p(deriv), _p(rk), _p(ic), _p(sol), _p(nu),
                                     _p(info), C.c_int(nthreads), _p(work), wf.ctypes.data_as(C.POINTER(C.c_int32)))
    else:
      (port_lib().scvx_port_socp_f32 if f32 else (port_lib().scvx_port_socp_lin32 if lin32 else port_lib().scvx_port_socp))(C.byref(c), C.c_int(B), _p(xbar), _p(ubar), _p(endpoint), _p(deriv), _p(rk), _p(ic),
                              _p(sol), _p(nu), _p(info), C.c_int(nthreads))
    nx = 14 * (K + 1)
    return dict(dx=sol[:, :nx].reshape(B, K + 1, 14), du=sol[:, nx:nx + NU * (K + 1)].reshape(B, K + 1, NU),
                ds=sol[:, -1], nu=nu, status=info[:, 0].astype(int), iters=info[:, 1].astype(int),
                merit=info[:, 2], pobj=info[:, 3])


def scvx_steps(p: DescentProblem, ic, steps, nsub=10, nthreads=0, tol=1e-8, accept=0.0, max_iter=60, refine=6, f32=False,
               on_step=None, warm_start=False):
    """`steps` Rocketland.solve_step calls (rocketland.jl:226-321) on B dispersed trajectories, all on the host: the conic
    solve by the CPU twin of the device solver (scvx_port.cpp), discretisation and propagation by the C oracle
    (scvx_oracle.c), accept / reject and radius update in numpy.  Starts from create_initial (straight-line guess).
    Returns dict(x, u, sigma, rk, cost, merit [steps][B], status, iters, rejected)."""
    from . import dynamics as od
    from .model import linear_points
    par = od.Params(p)
    K = p.K
    ic = np.ascontiguousarray(ic, float)
    B = ic.shape[0]
    x = np.zeros((B, K + 1, 14)); u = np.zeros((B, K + 1, p.nu))
    for t in range(B):
        x[t], u[t] = linear_points(p, ic[t, :3], ic[t, 3:])
    sig = np.full(B, p.tf_guess)
    dt = 1.0 / (K + 1)
    if nthreads > 0:   # the OpenMP runtime is shared by both oracle libraries: this also pins the discretisation
        port_lib().scvx_port_set_threads(C.c_int(nthreads))
    e, d = od.linearize(par, x, u, sig, dt, nsub)
    rk = np.full(B, 100.0); cost = np.full(B, np.inf)
    out = dict(merit=[], status=[], iters=[], rejected=[])
    work = None; was_rej = np.zeros(B, np.int32)
    if warm_start:
        port_lib().scvx_port_work_doubles_nu.restype = C.c_size_t
        work = np.zeros((B, port_lib().scvx_port_work_doubles_nu(C.c_int(K), C.c_int(1 if getattr(p, "enforce_dp", False) else 0), C.c_int(p.nu))))
    for s in range(steps):
        r = socp(p, x, u, e, d.astype(np.float32).astype(np.float64) if f32 else d, rk, ic, tol=tol, max_iter=max_iter,
                 refine=refine, nthreads=nthreads, accept=accept, f32=f32, work=work, warm=was_rej)
        xr = x + r["dx"]; ur = u + r["du"]; ns = sig + r["ds"]
        xn = od.propagate(par, xr, ur, ns, dt, nsub)
        jK = -xr[:, K, 0] + p.wNu * np.sqrt(((xr[:, 1:] - xn) ** 2).sum((1, 2)))      # rocketland.jl:289
        lK = -xr[:, K, 0] + p.wNu * np.sqrt((r["nu"] ** 2).sum((1, 2)))                 # :290
        with np.errstate(invalid="ignore"):
            rho = (cost - jK) / (cost - lK)
        ok = (r["status"] == 0) | (r["status"] == 4)
        rej = (rho < p.rh0) & ok                                                        # :299-301
        acc = ~rej & ok
        nrk = np.where(rej | (rho < p.rh1), rk / p.alph, np.where(rho < p.rh2, rk, p.bet * rk))
        nrk = np.where(np.isnan(rho), p.bet * rk, nrk)                                  # first call: rho = NaN -> grow
        rk = np.where(ok, nrk, rk)
        x[acc] = xr[acc]; u[acc] = ur[acc]; sig[acc] = ns[acc]; cost[acc] = jK[acc]
        e, d = od.linearize(par, x, u, sig, dt, nsub)                                   # :318
        was_rej = rej.astype(np.int32)
        out["merit"].append(r["merit"]); out["status"].append(r["status"]); out["iters"].append(r["iters"])
        out["rejected"].append(rej)
        if on_step is not None:
            on_step(s, r, rej)
    out.update(x=x, u=u, sigma=sig, rk=rk, cost=cost)
    return out


class Problem3(C.Structure):
    """scvx::td::Problem3 of scvx_threedof_core.hpp: the DescentProblem fields solve_initial reads."""
    _fields_ = [("K", C.c_int)] + [(n, C.c_double) for n in
                                   ("alpha", "tf_guess", "mwet", "mdry", "g", "Tmin", "Tmax", "thetaMax", "gammaGs")]


def threedof(p: DescentProblem, ic=None, tol=1e-9, max_iter=60, refine=1, delta=1e-9, nthreads=0):
    """CPU twin of the device 3-DoF initialiser (K0).  ic [B][6] = (rIi, vIi), None = the problem's own.  Returns
    (sol, status, info) in the layout of successiveconvexification_amd.first_round.solve_initial_batch: sol = dict of
    T, r, v, ar [B][3][K+1]; ma, ga, kaR [B][K+1]; nkaR [B]; info [B][5] = iterations, objective, gap, pres, dres."""
    P = Problem3(p.K, p.alpha, p.tf_guess, p.mwet, p.mdry, p.g, p.Tmin, p.Tmax, p.thetaMax, p.gammaGs)
    ic = np.concatenate([p.rIi, p.vIi])[None, :] if ic is None else ic
    ic = np.ascontiguousarray(ic, float)
    B, K = ic.shape[0], p.K
    rec = np.zeros((B, (K + 1) * 15 + 1))
    raw = np.zeros((B, 6))
    rc = port_lib().scvx_port_threedof(C.byref(P), C.c_int(B), _p(ic), _p(rec), _p(raw), C.c_double(tol), C.c_int(max_iter),
                                       C.c_int(refine), C.c_double(delta), C.c_int(nthreads))
    if rc != 0:
        raise RuntimeError("scvx_port_threedof failed")
    nodes = rec[:, :-1].reshape(B, K + 1, 15)
    tr = lambda a: np.ascontiguousarray(np.swapaxes(a, 1, 2))
    sol = dict(r=tr(nodes[:, :, 0:3]), v=tr(nodes[:, :, 3:6]), ma=nodes[:, :, 6].copy(), T=tr(nodes[:, :, 7:10]),
               ga=nodes[:, :, 10].copy(), kaR=nodes[:, :, 11].copy(), ar=tr(nodes[:, :, 12:15]), nkaR=rec[:, -1].copy())
    return sol, raw[:, 0].astype(np.int32), raw[:, 1:].copy()
