"""ctypes front-end of oracle/scvx_oracle.c (oracle; see oracle/__init__.py for the rules).

Also holds the cubic B-spline prefilter restating Interpolations.jl
``interpolate(A, BSpline(Cubic(Line(OnGrid()))))`` as used by Aerodynamics.load_aerodata
(aerodynamics.jl:17-21): coefficients c on a grid padded by one node per side such that
(c[i-1] + 4 c[i] + c[i+1]) / 6 = A[i] at every grid node and the second difference of c vanishes at
the two end nodes (natural spline), applied separably along both axes.
"""
import ctypes as C
import numpy as np

from . import lib
from .model import DescentProblem

NX, NU, NP = 14, 3, 21
_dp = C.POINTER(C.c_double)


class OracleParams(C.Structure):
    _fields_ = [
        ("alpha", C.c_double), ("g0", C.c_double), ("sos", C.c_double),
        ("J", C.c_double * 9), ("Jinv", C.c_double * 9),
        ("rTB", C.c_double * 3), ("rFB", C.c_double * 3),
        ("aero_kind", C.c_int32), ("n_aoa", C.c_int32), ("n_mach", C.c_int32), ("pad", C.c_int32),
        ("aoa0", C.c_double), ("daoa", C.c_double), ("mach0", C.c_double), ("dmach", C.c_double),
        ("force_scalar", C.c_double), ("length_scalar", C.c_double),
        ("cdrag", _dp), ("clift", _dp),
        ("nu", C.c_int32), ("pad2", C.c_int32),
    ]


def prefilter_1d(n: int) -> np.ndarray:
    """(n+2)x(n+2) system matrix of the Cubic(Line(OnGrid())) prefilter along one axis."""
    M = np.zeros((n + 2, n + 2))
    M[0, 0:3] = [1.0, -2.0, 1.0]
    M[n + 1, n - 1:n + 2] = [1.0, -2.0, 1.0]
    for i in range(1, n + 1):
        M[i, i - 1:i + 2] = [1.0 / 6.0, 4.0 / 6.0, 1.0 / 6.0]
    return M


def prefilter_table(tab: np.ndarray) -> np.ndarray:
    """tab [n_mach][n_aoa] -> coefficients [(n_mach+2)][(n_aoa+2)] (aoa fastest)."""
    nm, na = tab.shape
    Ma, Mm = prefilter_1d(na), prefilter_1d(nm)
    rhs = np.zeros((nm, na + 2))
    rhs[:, 1:na + 1] = tab
    ca = np.linalg.solve(Ma, rhs.T).T  # along aoa: [nm][na+2]
    rhs2 = np.zeros((nm + 2, na + 2))
    rhs2[1:nm + 1, :] = ca
    return np.ascontiguousarray(np.linalg.solve(Mm, rhs2))


class Params:
    """ProbInfo (master.jl:73-83) as the C oracle wants it; keeps the coefficient arrays alive."""

    def __init__(self, p: DescentProblem):
        s = OracleParams()
        s.alpha, s.g0, s.sos = p.alpha, p.g, p.sos
        J = np.asarray(p.jB, float)
        Ji = np.linalg.inv(J)
        s.J[:] = list(J.flatten(order="F"))
        s.Jinv[:] = list(Ji.flatten(order="F"))
        s.rTB[:] = list(p.rTB)
        s.rFB[:] = list(p.rFB)
        s.nu = 5 if getattr(p, "fins", False) else 3     # control_dim (fin extension: build-defined)
        self.nu, self.np = int(s.nu), 14 + 2 * int(s.nu) + 1
        self._keep = []
        if p.aero is None:
            s.aero_kind = 0
        else:
            a = p.aero
            s.aero_kind = 1
            s.n_mach, s.n_aoa = a.drag.shape
            s.aoa0, s.daoa, s.mach0, s.dmach = a.aoa0, a.daoa, a.mach0, a.dmach
            s.force_scalar, s.length_scalar = a.force_scalar, a.length_scalar
            cd, cl = prefilter_table(a.drag), prefilter_table(a.lift)
            self._keep = [cd, cl]
            s.cdrag = cd.ctypes.data_as(_dp)
            s.clift = cl.ctypes.data_as(_dp)
        self.c = s


def _p(a):
    return a.ctypes.data_as(_dp)


def rhs(par: Params, x, u):
    x = np.ascontiguousarray(x, float)
    u = np.ascontiguousarray(u, float)
    g = np.zeros(NX)
    lib().scvx_oracle_rhs(C.byref(par.c), _p(x), _p(u), _p(g))
    return g


def jac(par: Params, x, u):
    x = np.ascontiguousarray(x, float)
    u = np.ascontiguousarray(u, float)
    A = np.zeros((NX, NX))
    Bu = np.zeros((NX, par.nu))
    lib().scvx_oracle_jac(C.byref(par.c), _p(x), _p(u), _p(A), _p(Bu))
    return A, Bu


def segment(par: Params, inp, dt, nsub=10, with_deriv=True):
    inp = np.ascontiguousarray(inp, float)
    e = np.zeros(NX)
    d = np.zeros((par.np, NX)) if with_deriv else None
    lib().scvx_oracle_segment(C.byref(par.c), _p(inp), C.c_double(dt), C.c_int(nsub), _p(e),
                              _p(d) if with_deriv else None)
    return (e, d.T.copy()) if with_deriv else e  # derivative returned as a 14 x np matrix


def linearize(par: Params, x, u, sigma, dt, nsub=10):
    """x [B][K+1][14], u [B][K+1][nu], sigma [B] -> endpoint [B][K][14], deriv [B][K][np][14]."""
    x = np.ascontiguousarray(x, float)
    u = np.ascontiguousarray(u, float)
    sigma = np.ascontiguousarray(sigma, float)
    B, K1, _ = x.shape
    K = K1 - 1
    e = np.zeros((B, K, NX))
    assert u.shape[-1] == par.nu, (u.shape, par.nu)
    d = np.zeros((B, K, par.np, NX))
    lib().scvx_oracle_linearize(C.byref(par.c), C.c_int(B), C.c_int(K), _p(x), _p(u), _p(sigma),
                                C.c_double(dt), C.c_int(nsub), _p(e), _p(d))
    return e, d


def propagate(par: Params, x, u, sigma, dt, nsub=10):
    x = np.ascontiguousarray(x, float)
    u = np.ascontiguousarray(u, float)
    sigma = np.ascontiguousarray(sigma, float)
    B, K1, _ = x.shape
    K = K1 - 1
    assert u.shape[-1] == par.nu, (u.shape, par.nu)
    e = np.zeros((B, K, NX))
    lib().scvx_oracle_propagate(C.byref(par.c), C.c_int(B), C.c_int(K), _p(x), _p(u), _p(sigma),
                                C.c_double(dt), C.c_int(nsub), _p(e))
    return e


def table_eval(par: Params, which, aoa, mach):
    out = np.zeros(3)
    lib().scvx_oracle_table_eval(C.byref(par.c), C.c_int(which), C.c_double(aoa), C.c_double(mach), _p(out))
    return out


def aero_force(par: Params, q, v):
    q = np.ascontiguousarray(q, float)
    v = np.ascontiguousarray(v, float)
    F = np.zeros(3)
    dF = np.zeros((3, 7))
    lib().scvx_oracle_aero_force(C.byref(par.c), _p(q), _p(v), _p(F), _p(dF))
    return F, dF


def fin_dirs(q, v):
    """fd1 = normalize((C(q) e2) x v), fd2 = fd1 x v (dynamics.jl:60-62) and their derivatives w.r.t. (q, v), 3x7 each."""
    q = np.ascontiguousarray(q, float)
    v = np.ascontiguousarray(v, float)
    f1, f2, d1, d2 = np.zeros(3), np.zeros(3), np.zeros((3, 7)), np.zeros((3, 7))
    lib().scvx_oracle_fin_dirs(_p(q), _p(v), _p(f1), _p(f2), _p(d1), _p(d2))
    return f1, f2, d1, d2
