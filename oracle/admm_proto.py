"""numpy prototype of the batched operator-splitting SOCP solver (design study for the HIP kernel;
oracle-side test infrastructure, see oracle/__init__.py).

Reduced form of the Rocketland.build_model SOCP (rocketland.jl:53-219): x = xbar + dx, u = ubar + du and
the helper variables are eliminated, leaving w = (dx[K+1][14], du[K+1][3], dsigma, nu[K][14]) with

    minimise  -dx[K][0] + wNu ||nu|| + 0.5 ||(dx,du)|| + |dsigma|
    s.t.      E w = -d                      (linearised dynamics, rocketland.jl:117-133)
              boundary components fixed     (rocketland.jl:109-115)
              ||(dx,du)|| <= rk             (:215-216)
              node sets on x_k, u_k         (:137-201)

ADMM (Boyd et al. 2011 §3, scaled form, over-relaxed): f(w) = c'w + I{Ew=-d, fixed comps}, and one
consensus copy per prox-able term.
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

NX, NU = 14, 3


class Problem:
    def __init__(self, p, xbar, ubar, endpoint, deriv, rk):
        self.p = p
        K = p.K
        self.K = K
        self.xbar, self.ubar = xbar, ubar
        self.rk = rk
        self.nxu = 17 * (K + 1)
        self.n = self.nxu + 1 + 14 * K
        self.ix = np.arange(14 * (K + 1)).reshape(K + 1, 14)
        self.iu = 14 * (K + 1) + np.arange(3 * (K + 1)).reshape(K + 1, 3)
        self.isg = self.nxu
        self.inu = self.nxu + 1 + np.arange(14 * K).reshape(K, 14)
        rows, cols, vals = [], [], []
        d = np.zeros(14 * K)
        for k in range(K):
            D = deriv[k].T
            cc = np.concatenate([self.ix[k], self.iu[k], self.iu[k + 1], [self.isg]])
            for i in range(14):
                r = 14 * k + i
                rows += [r] * 21
                cols += list(cc)
                vals += list(D[i])
                rows += [r, r]
                cols += [self.inu[k, i], self.ix[k + 1, i]]
                vals += [1.0, -1.0]
            d[14 * k:14 * k + 14] = endpoint[k] - xbar[k + 1]
        self.E = sp.csr_matrix((vals, (rows, cols)), shape=(14 * K, self.n))
        self.d = d
        # fixed components
        fixed = np.zeros(self.n, bool)
        w0 = np.zeros(self.n)

        def fix(idx, val):
            fixed[idx] = True
            w0[idx] = val

        fix(self.ix[0, 0], p.mwet - xbar[0, 0])
        fix(self.ix[0, 1:4], p.rIi - xbar[0, 1:4])
        fix(self.ix[0, 4:7], p.vIi - xbar[0, 4:7])
        fix(self.ix[0, 11:14], p.wBi - xbar[0, 11:14])
        fix(self.ix[K, 1:4], p.rIf - xbar[K, 1:4])
        fix(self.ix[K, 4:7], p.vIf - xbar[K, 4:7])
        fix(self.ix[K, 7:11], p.qBIf - xbar[K, 7:11])
        fix(self.ix[K, 11:14], p.wBf - xbar[K, 11:14])
        fix(self.iu[K, 1:3], 0.0 - ubar[K, 1:3])
        self.fixed, self.w0 = fixed, w0
        c = np.zeros(self.n)
        c[self.ix[K, 0]] = -1.0
        self.c = c
        self.tggs = np.tan(np.radians(p.gammaGs))
        self.sqcm = np.sqrt((1 - np.cos(np.radians(p.thetaMax))) / 2)
        self.cosd = np.cos(np.radians(p.deltaMax))

    # ---- projections ----
    @staticmethod
    def proj_soc(t, v):
        """project (t, v) onto {||v|| <= t} — batched over leading dims."""
        nv = np.linalg.norm(v, axis=-1)
        t2 = np.where(nv <= t, t, np.where(nv <= -t, 0.0, 0.5 * (t + nv)))
        scale = np.where(nv <= t, 1.0, np.where(nv <= -t, 0.0, 0.5 * (t + nv) / np.maximum(nv, 1e-300)))
        return t2, v * scale[..., None]

    def proj_x(self, X):
        """X [K+1][14] absolute states."""
        p, K = self.p, self.K
        Y = X.copy()
        Y[1:, 0] = np.maximum(Y[1:, 0], p.mdry)
        # glideslope: ||r[2:3]|| <= r1 / tan(gs)  <=> SOC in (r1/tggs, r23) with scaled head
        a = 1.0 / self.tggs
        # project onto {(r1, r23): ||r23|| <= a r1}: scale head: t = a r1 -> cone ||v|| <= t with metric
        # exact Euclidean projection onto cone ||v|| <= a t0:
        t0 = Y[:K, 1]
        v = Y[:K, 2:4]
        nv = np.linalg.norm(v, axis=-1)
        inside = nv <= a * t0
        polar = a * nv <= -t0
        coef = (t0 + a * nv) / (1 + a * a)  # new t
        tn = np.where(inside, t0, np.where(polar, 0.0, coef))
        sc = np.where(inside, 1.0, np.where(polar, 0.0, a * coef / np.maximum(nv, 1e-300)))
        Y[:K, 1] = tn
        Y[:K, 2:4] = v * sc[:, None]
        # tilt ball
        v = Y[:K, 9:11]
        nv = np.linalg.norm(v, axis=-1)
        Y[:K, 9:11] = v * np.minimum(1.0, self.sqcm / np.maximum(nv, 1e-300))[:, None]
        v = Y[:K, 11:14]
        nv = np.linalg.norm(v, axis=-1)
        Y[:K, 11:14] = v * np.minimum(1.0, p.omMax / np.maximum(nv, 1e-300))[:, None]
        return Y

    def proj_u_ballcone(self, U):
        """{||u|| <= u1 / cos(dmax)} ∩ {||u|| <= Tmax}: cone projection then radial clip."""
        p = self.p
        c = self.cosd
        # cone: ||u|| <= u1/c  <=> ||u_23|| <= u1 * tan(dmax)
        a = np.sqrt(1 - c * c) / c
        t0 = U[:, 0]
        v = U[:, 1:3]
        nv = np.linalg.norm(v, axis=-1)
        inside = nv <= a * t0
        polar = a * nv <= -t0
        coef = (t0 + a * nv) / (1 + a * a)
        tn = np.where(inside, t0, np.where(polar, 0.0, coef))
        sc = np.where(inside, 1.0, np.where(polar, 0.0, a * coef / np.maximum(nv, 1e-300)))
        Y = np.concatenate([tn[:, None], v * sc[:, None]], axis=1)
        n = np.linalg.norm(Y, axis=-1)
        Y = Y * np.minimum(1.0, p.Tmax / np.maximum(n, 1e-300))[:, None]
        return Y

    def proj_u_half(self, dU):
        """uhat . du >= Tmin - ||ubar||."""
        p = self.p
        un = np.linalg.norm(self.ubar, axis=-1)
        uh = self.ubar / un[:, None]
        b = p.Tmin - un
        viol = b - np.sum(uh * dU, axis=-1)
        return dU + np.maximum(viol, 0.0)[:, None] * uh


def admm(P: Problem, rho=1.0, alpha=1.6, iters=2000, w_ref=None, log_every=100, rho_nu=None, rho_u=None, rho_s=None,
         warm=None, eps=None):
    K, n = P.K, P.n
    nxu = P.nxu
    ixu = np.arange(nxu)
    iX = P.ix.ravel()
    iU = P.iu.ravel()
    inu = P.inu.ravel()
    rho_nu = rho if rho_nu is None else rho_nu
    rho_u = rho if rho_u is None else rho_u
    rho_s = rho if rho_s is None else rho_s
    # weights per copy
    r1 = np.full(nxu, rho)       # TR copy on (dx,du)
    r2 = rho_nu                  # nu
    r3 = rho_s                   # sigma
    r4 = rho                     # X sets
    r5 = rho_u                   # U ball-cone
    r6 = rho_u                   # U halfspace
    Dg = np.zeros(n)
    Dg[ixu] += r1
    Dg[inu] += r2
    Dg[P.isg] += r3
    Dg[iX] += r4
    Dg[iU] += r5 + r6
    Dinv = np.where(P.fixed, 0.0, 1.0 / Dg)
    S = (P.E @ sp.diags(Dinv) @ P.E.T).tocsc()
    lu = spla.splu(S)
    Ew0 = P.E @ P.w0
    if warm is None:
        y1 = np.zeros(nxu); y2 = np.zeros(14 * K); y3 = 0.0
        y4 = P.proj_x(P.xbar).ravel() - P.xbar.ravel(); y5 = P.proj_u_ballcone(P.ubar).ravel(); y6 = np.zeros(3 * (K + 1))
        l1 = np.zeros(nxu); l2 = np.zeros(14 * K); l3 = 0.0; l4 = np.zeros(14 * (K + 1)); l5 = np.zeros(3 * (K + 1)); l6 = np.zeros(3 * (K + 1))
    else:
        y1, y2, y3, y4, y5, y6, l1, l2, l3, l4, l5, l6 = [np.copy(a) for a in warm]
    xb = P.xbar.ravel()
    ub = P.ubar.ravel()
    hist = []
    for it in range(1, iters + 1):
        rhs = -P.c.copy()
        rhs[ixu] += r1 * (y1 - l1)
        rhs[inu] += r2 * (y2 - l2)
        rhs[P.isg] += r3 * (y3 - l3)
        rhs[iX] += r4 * (y4 - l4)
        rhs[iU] += r5 * (y5 - ub - l5) + r6 * (y6 - l6)
        v = Dinv * rhs
        mu = lu.solve(P.E @ v + Ew0 + P.d)
        w = P.w0 + Dinv * (rhs - P.E.T @ mu)
        # relaxed images
        m1 = alpha * w[ixu] + (1 - alpha) * y1
        m2 = alpha * w[inu] + (1 - alpha) * y2
        m3 = alpha * w[P.isg] + (1 - alpha) * y3
        m4 = alpha * w[iX] + (1 - alpha) * y4
        m5 = alpha * (w[iU] + ub) + (1 - alpha) * y5
        m6 = alpha * w[iU] + (1 - alpha) * y6
        y1o, y2o, y3o, y4o, y5o, y6o = y1, y2, y3, y4, y5, y6
        z = m1 + l1
        nz = np.linalg.norm(z)
        y1 = z * (min(max(nz - 0.5 / rho, 0.0), P.rk) / max(nz, 1e-300))
        z = m2 + l2
        nz = np.linalg.norm(z)
        y2 = z * max(0.0, 1.0 - (P.p.wNu / r2) / max(nz, 1e-300))
        z = m3 + l3
        y3 = np.sign(z) * max(abs(z) - 1.0 / r3, 0.0)
        y4 = P.proj_x((m4 + l4 + xb).reshape(K + 1, 14)).ravel() - xb
        y5 = P.proj_u_ballcone((m5 + l5).reshape(K + 1, 3)).ravel()
        y6 = P.proj_u_half((m6 + l6).reshape(K + 1, 3)).ravel()
        l1 = l1 + m1 - y1
        l2 = l2 + m2 - y2
        l3 = l3 + m3 - y3
        l4 = l4 + m4 - y4
        l5 = l5 + m5 - y5
        l6 = l6 + m6 - y6
        if it % log_every == 0 or it == iters or eps is not None:
            rp = np.sqrt(np.sum((w[ixu] - y1) ** 2) + np.sum((w[inu] - y2) ** 2) + (w[P.isg] - y3) ** 2
                         + np.sum((w[iX] - y4) ** 2) + np.sum((w[iU] + ub - y5) ** 2) + np.sum((w[iU] - y6) ** 2))
            sd = np.zeros(n)
            sd[ixu] += r1 * (y1 - y1o)
            sd[inu] += r2 * (y2 - y2o)
            sd[P.isg] += r3 * (y3 - y3o)
            sd[iX] += r4 * (y4 - y4o)
            sd[iU] += r5 * (y5 - y5o) + r6 * (y6 - y6o)
            rd = np.linalg.norm(sd)
            err = np.abs(w - w_ref).max() if w_ref is not None else np.nan
            erru = np.abs(w[iU] - w_ref[iU]).max() if w_ref is not None else np.nan
            obj = P.c @ w + P.p.wNu * np.linalg.norm(w[inu]) + 0.5 * np.linalg.norm(w[ixu]) + abs(w[P.isg])
            hist.append((it, rp, rd, err, erru, obj))
            if eps is not None and rp < eps and rd < eps:
                break
    state = (y1, y2, y3, y4, y5, y6, l1, l2, l3, l4, l5, l6)
    return w, hist, state


def ipm_reference(p, P: Problem, data):
    """solve the same subproblem with the full build_model form + IPM, return it as a w vector."""
    from . import ipm, socp
    c, A, b, G, h, l, q, ix = socp.build(p, data["x"], data["u"], data["e"], data["d"], data["rk"])
    sol = ipm.solve(c, A, b, G, h, l, q)
    z = sol.x
    K = p.K
    w = np.zeros(P.n)
    w[P.ix] = z[ix.dxv].T
    w[P.iu] = z[ix.duv].T
    w[P.isg] = z[ix.dsig]
    w[P.inu] = z[ix.nuv].T[1:]
    return w, sol
