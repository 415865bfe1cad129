"""The SCvx trust-region SOCP, row for row as Rocketland.build_model assembles it
(oracle; see oracle/__init__.py for the rules).

Follows rocketland.jl:53-219 (variables :71-81,142,155,163,184,215; objective :84-86; rows cited inline)
with the per-iteration data of solve_step (rocketland.jl:245-269).  Sizes at K=50: 2,654 variables,
1,742 equalities, 204 linear inequalities, 204 second-order cones of total dimension 2,289
 — checked by tests/test_oracle_socp.py.

FIN EXTENSION (p.fins, control_dim = 5; build-defined): u and du get two more rows, and the commented rows
of rocketland.jl:203-209 are enabled -- finmxf[n] pinned to p.finmxf (the Zeros row, :205) and SOC(3) on
[finmxf[n]; u[4:5, n]] at n = 1..K+1 (:208).  (The Nonpositives row :207 on u[4:5] is a second, contradictory sketch of the
same bound and stays out.)

Cone-program form handed to oracle.ipm:  min c'z  s.t.  A z = b,  G z + s = h,  s in R+^l x Q...
"""
from dataclasses import dataclass
import numpy as np
import scipy.sparse as sp

from .model import DescentProblem

NX = 14


@dataclass
class Index:
    K: int
    n: int
    xv: np.ndarray   # [14][K+1] variable indices
    uv: np.ndarray   # [3][K+1]
    dxv: np.ndarray
    duv: np.ndarray
    dsig: int
    nuv: np.ndarray
    Jvnu: int
    Jtr: int
    Jsig: int
    gshelp: np.ndarray
    aoa_help: np.ndarray
    ang_sp_help: np.ndarray
    mtk: np.ndarray
    rK: int
    finmxf: np.ndarray = None   # [K+1], fin extension only


def index(K: int, NU: int = 3) -> Index:
    pos = 0

    def take(shape):
        nonlocal pos
        cnt = int(np.prod(shape))
        a = np.arange(pos, pos + cnt).reshape(shape, order="F")  # Julia reshape is column-major
        pos += cnt
        return a

    xv = take((NX, K + 1))
    uv = take((NU, K + 1))
    dxv = take((NX, K + 1))
    duv = take((NU, K + 1))
    dsig = int(take((1,))[0])
    nuv = take((NX, K + 1))
    Jvnu = int(take((1,))[0])
    Jtr = int(take((1,))[0])
    Jsig = int(take((1,))[0])
    gshelp = take((K,))
    aoa_help = take((K,))
    ang_sp_help = take((K,))
    mtk = take((K + 1,))
    rK = int(take((1,))[0])
    finmxf = take((K + 1,)) if NU == 5 else None
    return Index(K, pos, xv, uv, dxv, duv, dsig, nuv, Jvnu, Jtr, Jsig, gshelp, aoa_help, ang_sp_help, mtk, rK, finmxf)


class _Rows:
    def __init__(self, n):
        self.n = n
        self.r, self.c, self.v, self.rhs = [], [], [], []
        self.m = 0

    def add(self, cols, vals, rhs):
        for cc, vv in zip(cols, vals):
            self.r.append(self.m)
            self.c.append(int(cc))
            self.v.append(float(vv))
        self.rhs.append(float(rhs))
        self.m += 1

    def mat(self):
        return sp.csc_matrix((self.v, (self.r, self.c)), shape=(self.m, self.n)), np.array(self.rhs)


def build(p: DescentProblem, xbar, ubar, endpoint, deriv, rk):
    """xbar [K+1][14], ubar [K+1][3] (iterAbout), endpoint [K][14], deriv [K][21][14] (iterDynam:
    column-major 14x21 per segment), rk = trust radius.  Returns (c, A, b, G, h, l, q, idx)."""
    K = p.K
    NU = 5 if getattr(p, "fins", False) else 3
    ix = index(K, NU)
    n = ix.n
    tggs = np.tan(np.radians(p.gammaGs))                       # rocketland.jl:63
    sqcm = np.sqrt((1 - np.cos(np.radians(p.thetaMax))) / 2)   # :64
    delMax = np.cos(np.radians(p.deltaMax))                    # :65

    c = np.zeros(n)                                            # objective :84-86
    c[ix.xv[0, K]] = -1.0
    c[ix.Jvnu] = p.wNu
    c[ix.Jtr] = 0.5
    c[ix.Jsig] = 1.0

    E = _Rows(n)
    # state_base: about + dx - x = 0  (:92-94)
    for k in range(K + 1):
        for i in range(NX):
            E.add([ix.dxv[i, k], ix.xv[i, k]], [1.0, -1.0], -xbar[k, i])
    # control_base (:95-97)
    for k in range(K + 1):
        for i in range(NU):
            E.add([ix.duv[i, k], ix.uv[i, k]], [1.0, -1.0], -ubar[k, i])
    # boundary conditions (:109-115)
    bvars = ([ix.xv[0, 0]] + list(ix.xv[1:4, 0]) + list(ix.xv[4:7, 0]) + list(ix.xv[11:14, 0])
             + list(ix.xv[1:4, K]) + list(ix.xv[4:7, K]) + list(ix.xv[7:11, K]) + list(ix.xv[11:14, K])
             + list(ix.uv[1:3, K]))
    bvals = np.concatenate([[p.mwet], p.rIi, p.vIi, p.wBi, p.rIf, p.vIf, p.qBIf, p.wBf, [0.0, 0.0]])
    for var, val in zip(bvars, bvals):
        E.add([var], [1.0], val)
    # linearised dynamics (:117-133):
    # derivative_n [dx_n; du_n; du_{n+1}; dsig] + nu_{n+1} - dx_{n+1} + (endpoint_n - xbar_{n+1}) = 0
    for k in range(K):
        D = deriv[k].T  # 14 x (14 + 2 NU + 1)
        cols = list(ix.dxv[:, k]) + list(ix.duv[:, k]) + list(ix.duv[:, k + 1]) + [ix.dsig]
        for i in range(NX):
            cc = cols + [ix.nuv[i, k + 1], ix.dxv[i, k + 1]]
            vv = list(D[i, :]) + [1.0, -1.0]
            E.add(cc, vv, -(endpoint[k, i] - xbar[k + 1, i]))
    # helper equalities: glideslope (:142-144), tilt (:155-156), rate (:163-164)
    for k in range(K):
        E.add([ix.gshelp[k], ix.xv[1, k]], [1.0, -1.0 / tggs], 0.0)
    for k in range(K):
        E.add([ix.aoa_help[k]], [1.0], sqcm)
    for k in range(K):
        E.add([ix.ang_sp_help[k]], [1.0], p.omMax)
    if NU == 5:   # finmxf[n] - p.finmxf = 0 (rocketland.jl:205 as commented there)
        for k in range(K + 1):
            E.add([ix.finmxf[k]], [1.0], p.finmxf)
    A, b = E.mat()

    L = _Rows(n)  # G z + s = h with s >= 0
    # mdry <= m_k, k = 2..K+1 (:137)
    for k in range(1, K + 1):
        L.add([ix.xv[0, k]], [-1.0], -p.mdry)
    # mtk <= Tmax (:186)
    for k in range(K + 1):
        L.add([ix.mtk[k]], [1.0], p.Tmax)
    # mtk <= u1 / cos(deltaMax) (:188)
    for k in range(K + 1):
        L.add([ix.mtk[k], ix.uv[0, k]], [1.0, -1.0 / delMax], 0.0)
    # linearised thrust lower bound (:199-201 and solve_step :261-265)
    for k in range(K + 1):
        un = np.linalg.norm(ubar[k, :3])   # the thrust part of the control (:199 indexes control[1:3])
        L.add(list(ix.duv[:3, k]), list(-ubar[k, :3] / un), -(p.Tmin - un))
    # hard trust region Jtr - rk <= 0 (:215-216, :269)
    L.add([ix.Jtr], [1.0], rk)
    l = L.m

    q = []

    def soc(vars_):
        for var in vars_:
            L.add([var], [-1.0], 0.0)
        q.append(len(vars_))

    soc([ix.Jvnu] + list(ix.nuv.flatten(order="F")))                                     # :100
    soc([ix.Jtr] + list(ix.dxv.flatten(order="F")) + list(ix.duv.flatten(order="F")))    # :101
    soc([ix.Jsig, ix.dsig])                                                              # :102
    for k in range(K):
        soc([ix.gshelp[k], ix.xv[2, k], ix.xv[3, k]])                                    # :146-148
    for k in range(K):
        soc([ix.aoa_help[k], ix.xv[9, k], ix.xv[10, k]])                                 # :158-160
    for k in range(K):
        soc([ix.ang_sp_help[k]] + list(ix.xv[11:14, k]))                                 # :165-167
    for k in range(K + 1):
        soc([ix.mtk[k]] + list(ix.uv[:3, k]))                                            # :190-192
    if NU == 5:
        for k in range(K + 1):
            soc([ix.finmxf[k]] + list(ix.uv[3:5, k]))                                    # :208 as commented there
    # dynamic pressure 1/2 rho |v_k|^2 <= dpMax, k = 1..K, as the cone (vmax; v_k) -- fields master.jl:27,30, constraint a
    # "todo" at rocketland.jl:211-212; only when the problem enables it (build extension)
    if getattr(p, "enforce_dp", False):
        vmax = float(np.sqrt(2.0 * p.dpMax / p.rho))
        for k in range(K):
            L.add([], [], vmax)
            for i in range(4, 7):
                L.add([ix.xv[i, k]], [-1.0], 0.0)
            q.append(4)
    G, h = L.mat()
    return c, A, b, G, h, l, q, ix
