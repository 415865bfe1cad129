"""ctypes front-end of oracle/scvx_port.cpp — the CPU twin of the device conic solver
(oracle; rules in oracle/__init__.py).  Consts mirrors scvx::ipm::Consts of scvx_ipm_core.hpp."""
import ctypes as C
import numpy as np

from . import port_lib
from .model import DescentProblem

_dp = C.POINTER(C.c_double)


class Consts(C.Structure):
    _fields_ = [("K", C.c_int), ("max_iter", C.c_int), ("refine", C.c_int), ("pad", C.c_int),
                ("tol", C.c_double), ("accept", C.c_double),
                ("itan", C.c_double), ("sqcm", C.c_double), ("icos", C.c_double), ("Tmax", C.c_double),
                ("Tmin", C.c_double), ("omMax", C.c_double), ("mdry", C.c_double), ("wNu", C.c_double),
                ("mwet", C.c_double),
                ("rIf", C.c_double * 3), ("vIf", C.c_double * 3), ("qBIf", C.c_double * 4),
                ("wBi", C.c_double * 3), ("wBf", C.c_double * 3)]


def consts(p: DescentProblem, tol=1e-8, max_iter=60, refine=6, accept=1e-6) -> Consts:
    c = Consts()
    c.K, c.max_iter, c.refine, c.tol, c.accept = p.K, max_iter, refine, tol, max(accept, tol)
    c.itan = 1.0 / np.tan(np.radians(p.gammaGs))       # rocketland.jl:63
    c.sqcm = np.sqrt((1 - np.cos(np.radians(p.thetaMax))) / 2)  # :64
    c.icos = 1.0 / np.cos(np.radians(p.deltaMax))      # :65
    c.Tmax, c.Tmin, c.omMax, c.mdry, c.wNu, c.mwet = p.Tmax, p.Tmin, p.omMax, p.mdry, p.wNu, p.mwet
    c.rIf[:] = list(p.rIf); c.vIf[:] = list(p.vIf); c.qBIf[:] = list(p.qBIf)
    c.wBi[:] = list(p.wBi); c.wBf[:] = list(p.wBf)
    return c


def _p(a):
    return a.ctypes.data_as(_dp)


def socp(p: DescentProblem, xbar, ubar, endpoint, deriv, rk, ic=None, tol=1e-8, max_iter=60, refine=6, nthreads=0, accept=1e-6,
         f32=False):
    """Batched: xbar [B][K+1][14], ubar [B][K+1][3], endpoint [B][K][14], deriv [B][K][21][14], rk [B].
    Returns dict(dx, du, ds, nu, status, iters, merit, pobj)."""
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
    c = consts(p, tol, max_iter, refine, accept)
    sol = np.zeros((B, (K + 1) * 17 + 1))
    nu = np.zeros((B, K, 14))
    info = np.zeros((B, 4))
    (port_lib().scvx_port_socp_f32 if f32 else port_lib().scvx_port_socp)(C.byref(c), C.c_int(B), _p(xbar), _p(ubar), _p(endpoint), _p(deriv), _p(rk), _p(ic),
                              _p(sol), _p(nu), _p(info), C.c_int(nthreads))
    nx = 14 * (K + 1)
    return dict(dx=sol[:, :nx].reshape(B, K + 1, 14), du=sol[:, nx:nx + 3 * (K + 1)].reshape(B, K + 1, 3),
                ds=sol[:, -1], nu=nu, status=info[:, 0].astype(int), iters=info[:, 1].astype(int),
                merit=info[:, 2], pobj=info[:, 3])
