"""BASELINE configs[0]: the 3-DoF point-mass landing SOCP of the reference's commented-out initialiser
(oracle; rules in oracle/__init__.py).  CPU only — "plumbing, no GPU".

Follows initial_solve.jl:17-110 (`solve_initial`, inside a #= =# block at HEAD), written for an old JuMP:
    variables   T[3,N+1], r[3,N+1], v[3,N+1], ma[N+1], ga[N+1], kaR[N+1], ar[3,N+1], nkaR      (:49-58; `s` unused,
                kaRr is an alias of kaR :71)
    objective   min -ma[N+1] + wkar * nkaR,  [nkaR; kaR] in SOC(N+2), wkar = 100                 (:68-70, :39)
    boundary    r1 = rIi, v1 = vIi, ma1 = mwet, r_{N+1} = 0, v_{N+1} = 0, T[2:3,N+1] = 0          (:60-66)
    dynamics    ma_{i+1} = ma_i - alpha (ga_i + ga_{i+1}) dt/2                                    (:73)
                a_i = T_i/mu_i + ar_i + [-g,0,0]   with the fixed linear mass profile mu            (:24, :74-75)
                r_{i+1} = r_i + v_i dt + 1/3 (a_i + a_{i+1}/2) dt^2,  v_{i+1} = v_i + (a_i + a_{i+1}) dt/2   (:76-77)
    per node    mdry <= ma, [r1/tan(gs); r2; r3] in SOC3, [ga; T] in SOC4, Tmin <= ga <= Tmax,
                ga cos(thetaMax) <= T1, [kaR; ar] in SOC4                                          (:80-88)
"""
import numpy as np
import scipy.sparse as sp

from . import ipm
from .model import DescentProblem


def build(p: DescentProblem):
    N = p.K
    dt = p.tf_guess / N
    mu = np.array([((N - k) / N) * p.mwet + (k / N) * p.mdry for k in range(N + 1)])
    tggs = np.tan(np.radians(p.gammaGs))
    cth = np.cos(np.radians(p.thetaMax))
    wkar = 100.0
    pos = 0

    def take(shape):
        nonlocal pos
        n = int(np.prod(shape))
        a = np.arange(pos, pos + n).reshape(shape, order="F")
        pos += n
        return a

    T, r, v = take((3, N + 1)), take((3, N + 1)), take((3, N + 1))
    ma, ga, kaR = take((N + 1,)), take((N + 1,)), take((N + 1,))
    ar = take((3, N + 1))
    nkaR = int(take((1,))[0])
    n = pos
    c = np.zeros(n)
    c[ma[N]] = -1.0
    c[nkaR] = wkar
    Ar, Ac, Av, b = [], [], [], []

    def eq(cols, vals, rhs):
        row = len(b)
        for cc, vv in zip(cols, vals):
            Ar.append(row); Ac.append(int(cc)); Av.append(float(vv))
        b.append(float(rhs))

    for i in range(3):
        eq([r[i, 0]], [1.0], p.rIi[i]); eq([v[i, 0]], [1.0], p.vIi[i])
        eq([r[i, N]], [1.0], 0.0); eq([v[i, N]], [1.0], 0.0)
    eq([ma[0]], [1.0], p.mwet)
    eq([T[1, N]], [1.0], 0.0); eq([T[2, N]], [1.0], 0.0)
    gvec = np.array([-p.g, 0.0, 0.0])
    for i in range(N):
        eq([ma[i + 1], ma[i], ga[i], ga[i + 1]], [1.0, -1.0, p.alpha * dt / 2, p.alpha * dt / 2], 0.0)
        for j in range(3):
            # r_{i+1} - r_i - v_i dt - dt^2/3 (T_i/mu_i + ar_i) - dt^2/6 (T_{i+1}/mu_{i+1} + ar_{i+1}) = dt^2/2 g_j
            eq([r[j, i + 1], r[j, i], v[j, i], T[j, i], ar[j, i], T[j, i + 1], ar[j, i + 1]],
               [1.0, -1.0, -dt, -dt**2 / 3 / mu[i], -dt**2 / 3, -dt**2 / 6 / mu[i + 1], -dt**2 / 6], dt**2 / 2 * gvec[j])
            eq([v[j, i + 1], v[j, i], T[j, i], ar[j, i], T[j, i + 1], ar[j, i + 1]],
               [1.0, -1.0, -dt / 2 / mu[i], -dt / 2, -dt / 2 / mu[i + 1], -dt / 2], dt * gvec[j])
    A = sp.csc_matrix((Av, (Ar, Ac)), shape=(len(b), n))
    Gr, Gc, Gv, h = [], [], [], []

    def row(cols, vals, rhs):
        rr = len(h)
        for cc, vv in zip(cols, vals):
            Gr.append(rr); Gc.append(int(cc)); Gv.append(float(vv))
        h.append(float(rhs))

    for i in range(N + 1):  # s >= 0 rows
        row([ma[i]], [-1.0], -p.mdry)
        row([ga[i]], [-1.0], -p.Tmin)
        row([ga[i]], [1.0], p.Tmax)
        row([ga[i], T[0, i]], [cth, -1.0], 0.0)
    l = len(h)
    q = []

    def soc(items):
        for cols, vals in items:
            row(cols, [-x for x in vals], 0.0)
        q.append(len(items))

    soc([([nkaR], [1.0])] + [([kaR[i]], [1.0]) for i in range(N + 1)])
    for i in range(N + 1):
        soc([([r[0, i]], [1.0 / tggs]), ([r[1, i]], [1.0]), ([r[2, i]], [1.0])])
        soc([([ga[i]], [1.0])] + [([T[j, i]], [1.0]) for j in range(3)])
        soc([([kaR[i]], [1.0])] + [([ar[j, i]], [1.0]) for j in range(3)])
    G = sp.csc_matrix((Gv, (Gr, Gc)), shape=(len(h), n))
    idx = dict(T=T, r=r, v=v, ma=ma, ga=ga, kaR=kaR, ar=ar, nkaR=nkaR, dt=dt, mu=mu)
    return c, A, np.array(b), G, np.array(h), l, q, idx


def solve_initial(p: DescentProblem, tol=1e-9):
    c, A, b, G, h, l, q, idx = build(p)
    sol = ipm.solve(c, A, b, G, h, l, q, tol=tol)
    z = sol.x
    out = {k: z[idx[k]] for k in ("T", "r", "v", "ma", "ga", "kaR", "ar")}
    out["nkaR"] = z[idx["nkaR"]]
    return sol, out, idx
