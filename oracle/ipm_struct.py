"""Structure-exploiting interior-point solver for the SCvx subproblem — numpy design twin of the HIP
kernel (oracle-side test infrastructure; rules in oracle/__init__.py).

Same optimisation problem as oracle.socp.build (rocketland.jl:53-219) in reduced variables
    w = (dx[K+1][14], du[K+1][3], nu[K][14], s=dsigma, tnu, ttr, ts)
with x = xbar + dx, u = ubar + du substituted, the helper variables gshelp/aoa_help/ang_sp_help/mtk
eliminated (their defining equalities are substituted into the cones), boundary components held
fixed, and the thrust pair  ||u|| <= mtk, mtk <= Tmax, mtk <= u1/cos(dmax)  written as the two cones
(Tmax; u) and (u1/cos(dmax); u).  Same algorithm as oracle.ipm (Mehrotra predictor-corrector,
Nesterov-Todd scaling); what differs is the linear algebra, organised the way the device does it:

    H = G'W^-2 G  =  block-diagonal per node  +  rank-1 per big cone  +  4 global scalars
    KKT [H E'; E 0] solved through S = E_loc Hb^-1 E_loc' (block tridiagonal, 14x14 blocks, K of them)
    with the dense pieces (dsigma column, the two big-cone body vectors, tnu/ttr/ts) as a 6-wide border.
"""
import numpy as np

NX, NU = 14, 3


CAPTURE = None   # set to a list to record (Sd, So) of every factorisation (oracle/bcr_proto.py)


def soc_nt(s, z):
    """Nesterov-Todd scaling of one second-order cone: returns (v, beta) with W = beta (2 v v' - J)."""
    sj = np.sqrt(s[0] * s[0] - s[1:] @ s[1:])
    zj = np.sqrt(z[0] * z[0] - z[1:] @ z[1:])
    sb, zb = s / sj, z / zj
    gam = np.sqrt((1 + sb @ zb) / 2)
    wb = sb.copy()
    wb[0] += zb[0]
    wb[1:] -= zb[1:]
    wb /= 2 * gam
    v = wb.copy()
    v[0] += 1.0
    v /= np.sqrt(2 * (wb[0] + 1.0))
    return v, np.sqrt(sj / zj)


def soc_W(v, beta, x, inverse=False):
    if not inverse:
        y = 2 * (v @ x) * v
        y[0] -= x[0]
        y[1:] += x[1:]
        return beta * y
    vt = v.copy()
    vt[1:] = -vt[1:]
    y = 2 * (vt @ x) * vt
    y[0] -= x[0]
    y[1:] += x[1:]
    return y / beta


def soc_Winv2_parts(v, beta):
    """W^-2 = beta^-2 [[h00, h01 v1'], [h01 v1, I + h11 v1 v1']] -> (h00, h01, h11) already / beta^2."""
    v0 = v[0]
    n1 = v[1:] @ v[1:]
    b2 = 1.0 / (beta * beta)
    return b2 * ((2 * v0 * v0 - 1) ** 2 + 4 * v0 * v0 * n1), b2 * (-4 * v0 * (v0 * v0 + n1)), b2 * 8 * v0 * v0, b2


def soc_prod(a, b):
    out = a[0] * b + b[0] * a
    out[0] = a @ b
    return out


def soc_div(lam, d):
    det = lam[0] ** 2 - lam[1:] @ lam[1:]
    x0 = (lam[0] * d[0] - lam[1:] @ d[1:]) / det
    x = (d - x0 * lam) / lam[0]
    x[0] = x0
    return x


def soc_maxstep(lam, d):
    amax = np.inf
    if d[0] < 0:
        amax = -lam[0] / d[0]
    a = d[0] ** 2 - d[1:] @ d[1:]
    b = 2 * (lam[0] * d[0] - lam[1:] @ d[1:])
    c = lam[0] ** 2 - lam[1:] @ lam[1:]
    disc = b * b - 4 * a * c
    if disc >= 0:
        sq = np.sqrt(disc)
        qq = -0.5 * (b + (sq if b >= 0 else -sq))
        for r in ((c / qq) if qq != 0 else np.inf, (qq / a) if a != 0 else np.inf):
            if r > 0:
                amax = min(amax, r)
    return amax


class Cones:
    """The cone list of the reduced problem.  Each entry: (kind, data) with the affine slack map."""

    def __init__(self, p, K, xbar, ubar, rk):
        self.p, self.K = p, K
        self.xbar, self.ubar, self.rk = xbar, ubar, rk
        self.itan = 1.0 / np.tan(np.radians(p.gammaGs))
        self.sqcm = np.sqrt((1 - np.cos(np.radians(p.thetaMax))) / 2)
        self.icos = 1.0 / np.cos(np.radians(p.deltaMax))
        un = np.linalg.norm(ubar, axis=1)
        self.uhat = ubar / un[:, None]
        self.lb0 = p.Tmin - un
        # cone table: name, count, dim
        self.spec = [("gs", K, 3), ("tilt", K, 3), ("rate", K, 4), ("mass", K, 1), ("tb", K + 1, 4), ("tc", K + 1, 4),
                     ("lb", K + 1, 1), ("nu", 1, 14 * K + 1), ("tr", 1, 17 * (K + 1) + 1), ("sg", 1, 2), ("rk", 1, 1)]
        self.degree = sum(cnt for _, cnt, _ in self.spec)

    def zeros(self):
        return {n: np.zeros((cnt, dim)) for n, cnt, dim in self.spec}

    def identity(self):
        e = self.zeros()
        for n in e:
            e[n][:, 0] = 1.0
        return e

    def affine(self, V):
        """a(w): the value each slack must equal."""
        p, K = self.p, self.K
        x = self.xbar + V["dx"]
        u = self.ubar + V["du"]
        a = self.zeros()
        a["gs"][:, 0] = x[:K, 1] * self.itan
        a["gs"][:, 1:] = x[:K, 2:4]
        a["tilt"][:, 0] = self.sqcm
        a["tilt"][:, 1:] = x[:K, 9:11]
        a["rate"][:, 0] = p.omMax
        a["rate"][:, 1:] = x[:K, 11:14]
        a["mass"][:, 0] = x[1:, 0] - p.mdry
        a["tb"][:, 0] = p.Tmax
        a["tb"][:, 1:] = u
        a["tc"][:, 0] = u[:, 0] * self.icos
        a["tc"][:, 1:] = u
        a["lb"][:, 0] = np.sum(self.uhat * V["du"], axis=1) - self.lb0
        a["nu"][0, 0] = V["tnu"]
        a["nu"][0, 1:] = V["nu"].ravel()
        a["tr"][0, 0] = V["ttr"]
        a["tr"][0, 1:] = np.concatenate([V["dx"].ravel(), V["du"].ravel()])
        a["sg"][0] = [V["ts"], V["s"]]
        a["rk"][0, 0] = self.rk - V["ttr"]
        return a

    def jac_apply(self, dV):
        """J dw (J = d a / d w)."""
        K = self.K
        a = self.zeros()
        dx, du = dV["dx"], dV["du"]
        a["gs"][:, 0] = dx[:K, 1] * self.itan
        a["gs"][:, 1:] = dx[:K, 2:4]
        a["tilt"][:, 1:] = dx[:K, 9:11]
        a["rate"][:, 1:] = dx[:K, 11:14]
        a["mass"][:, 0] = dx[1:, 0]
        a["tb"][:, 1:] = du
        a["tc"][:, 0] = du[:, 0] * self.icos
        a["tc"][:, 1:] = du
        a["lb"][:, 0] = np.sum(self.uhat * du, axis=1)
        a["nu"][0, 0] = dV["tnu"]
        a["nu"][0, 1:] = dV["nu"].ravel()
        a["tr"][0, 0] = dV["ttr"]
        a["tr"][0, 1:] = np.concatenate([dx.ravel(), du.ravel()])
        a["sg"][0] = [dV["ts"], dV["s"]]
        a["rk"][0, 0] = -dV["ttr"]
        return a

    def jact_apply(self, Z):
        """J' z as a variable-shaped dict."""
        K = self.K
        g = dict(dx=np.zeros((K + 1, NX)), du=np.zeros((K + 1, NU)), nu=np.zeros((K, NX)), s=0.0, tnu=0.0, ttr=0.0, ts=0.0)
        g["dx"][:K, 1] += Z["gs"][:, 0] * self.itan
        g["dx"][:K, 2:4] += Z["gs"][:, 1:]
        g["dx"][:K, 9:11] += Z["tilt"][:, 1:]
        g["dx"][:K, 11:14] += Z["rate"][:, 1:]
        g["dx"][1:, 0] += Z["mass"][:, 0]
        g["du"] += Z["tb"][:, 1:]
        g["du"][:, 0] += Z["tc"][:, 0] * self.icos
        g["du"] += Z["tc"][:, 1:]
        g["du"] += Z["lb"][:, 0:1] * self.uhat
        g["tnu"] += Z["nu"][0, 0]
        g["nu"] += Z["nu"][0, 1:].reshape(K, NX)
        g["ttr"] += Z["tr"][0, 0] - Z["rk"][0, 0]
        n = 14 * (K + 1)
        g["dx"] += Z["tr"][0, 1:1 + n].reshape(K + 1, NX)
        g["du"] += Z["tr"][0, 1 + n:].reshape(K + 1, NU)
        g["ts"] += Z["sg"][0, 0]
        g["s"] += Z["sg"][0, 1]
        return g


def solve(p, xbar, ubar, endpoint, deriv, rk, tol=1e-8, max_iter=60, verbose=False, refine=1):
    K = p.K
    C = Cones(p, K, xbar, ubar, rk)
    Dk = np.stack([deriv[k].T for k in range(K)])  # [K][14][21]
    dk = endpoint - xbar[1:]
    A_, Bm, Bp, Sg = Dk[:, :, :14], Dk[:, :, 14:17], Dk[:, :, 17:20], Dk[:, :, 20]
    # fixed components (rocketland.jl:109-115)
    fx = np.zeros((K + 1, NX), bool)
    fu = np.zeros((K + 1, NU), bool)
    fx[0, [0, 1, 2, 3, 4, 5, 6, 11, 12, 13]] = True
    fx[K, 1:14] = True
    fu[K, 1:3] = True
    V = dict(dx=np.zeros((K + 1, NX)), du=np.zeros((K + 1, NU)), nu=np.zeros((K, NX)), s=0.0, tnu=0.0, ttr=0.0, ts=0.0)
    V["dx"][0, [0]] = p.mwet - xbar[0, 0]
    V["dx"][0, 1:4] = p.rIi - xbar[0, 1:4]
    V["dx"][0, 4:7] = p.vIi - xbar[0, 4:7]
    V["dx"][0, 11:14] = p.wBi - xbar[0, 11:14]
    V["dx"][K, 1:4] = p.rIf - xbar[K, 1:4]
    V["dx"][K, 4:7] = p.vIf - xbar[K, 4:7]
    V["dx"][K, 7:11] = p.qBIf - xbar[K, 7:11]
    V["dx"][K, 11:14] = p.wBf - xbar[K, 11:14]
    V["du"][K, 1:3] = 0.0 - ubar[K, 1:3]
    cost = dict(dx=np.zeros((K + 1, NX)), du=np.zeros((K + 1, NU)), nu=np.zeros((K, NX)), s=0.0, tnu=p.wNu, ttr=0.5, ts=1.0)
    cost["dx"][K, 0] = -1.0

    def E_apply(dV):
        """E dw (rows k = 0..K-1)."""
        r = (np.einsum("kij,kj->ki", A_, dV["dx"][:K]) + np.einsum("kij,kj->ki", Bm, dV["du"][:K])
             + np.einsum("kij,kj->ki", Bp, dV["du"][1:]) + Sg * dV["s"] + dV["nu"] - dV["dx"][1:])
        return r

    def Et_apply(y):
        g = dict(dx=np.zeros((K + 1, NX)), du=np.zeros((K + 1, NU)), nu=y.copy(), s=float(np.sum(Sg * y)), tnu=0.0, ttr=0.0, ts=0.0)
        g["dx"][:K] += np.einsum("kij,ki->kj", A_, y)
        g["dx"][1:] -= y
        g["du"][:K] += np.einsum("kij,ki->kj", Bm, y)
        g["du"][1:] += np.einsum("kij,ki->kj", Bp, y)
        return g

    def build_kkt(Wd):
        # ---- H: node blocks, big-cone pieces ----
        Hx = np.zeros((K + 1, NX, NX))
        Hu = np.zeros((K + 1, NU, NU))
        h_tr = soc_Winv2_parts(*Wd["tr"][0])
        h_nu = soc_Winv2_parts(*Wd["nu"][0])
        ptr = Wd["tr"][0][0][1:]
        pnu = Wd["nu"][0][0][1:]
        for k in range(K + 1):
            Hx[k] += h_tr[3] * np.eye(NX)
            Hu[k] += h_tr[3] * np.eye(NU)
        for k in range(K):
            # glideslope: rows (x1*itan, x2, x3)
            v, b = Wd["gs"][k]
            h00, h01, h11, b2 = soc_Winv2_parts(v, b)
            M = np.zeros((3, 3))
            M[0, 0] = h00
            M[0, 1:] = h01 * v[1:]
            M[1:, 0] = h01 * v[1:]
            M[1:, 1:] = b2 * np.eye(2) + h11 * np.outer(v[1:], v[1:])
            Jg = np.diag([C.itan, 1.0, 1.0])
            Hx[k][1:4, 1:4] += Jg @ M @ Jg
            v, b = Wd["tilt"][k]
            h00, h01, h11, b2 = soc_Winv2_parts(v, b)
            Hx[k][9:11, 9:11] += b2 * np.eye(2) + h11 * np.outer(v[1:], v[1:])
            v, b = Wd["rate"][k]
            h00, h01, h11, b2 = soc_Winv2_parts(v, b)
            Hx[k][11:14, 11:14] += b2 * np.eye(3) + h11 * np.outer(v[1:], v[1:])
            Hx[k + 1][0, 0] += 1.0 / Wd["mass"][k] ** 2
        for k in range(K + 1):
            v, b = Wd["tb"][k]
            h00, h01, h11, b2 = soc_Winv2_parts(v, b)
            Hu[k] += b2 * np.eye(3) + h11 * np.outer(v[1:], v[1:])
            v, b = Wd["tc"][k]
            h00, h01, h11, b2 = soc_Winv2_parts(v, b)
            M = np.zeros((4, 4))
            M[0, 0] = h00
            M[0, 1:] = h01 * v[1:]
            M[1:, 0] = h01 * v[1:]
            M[1:, 1:] = b2 * np.eye(3) + h11 * np.outer(v[1:], v[1:])
            Jc = np.zeros((4, 3))
            Jc[0, 0] = C.icos
            Jc[1:, :] = np.eye(3)
            Hu[k] += Jc.T @ M @ Jc
            Hu[k] += np.outer(C.uhat[k], C.uhat[k]) / Wd["lb"][k] ** 2
        # inverses with fixed components masked out
        Hxi = np.zeros_like(Hx)
        Hui = np.zeros_like(Hu)
        for k in range(K + 1):
            fr = ~fx[k]
            if fr.any():
                Hxi[k][np.ix_(fr, fr)] = np.linalg.inv(Hx[k][np.ix_(fr, fr)])
            fr = ~fu[k]
            Hui[k][np.ix_(fr, fr)] = np.linalg.inv(Hu[k][np.ix_(fr, fr)])
        hnui = 1.0 / h_nu[3]
        # ---- S = E_loc Hb^-1 E_loc' (block tridiagonal) ----
        Sd = np.zeros((K, NX, NX))
        So = np.zeros((K - 1, NX, NX))  # block (k+1, k)
        for k in range(K):
            Sd[k] = (A_[k] @ Hxi[k] @ A_[k].T + Bm[k] @ Hui[k] @ Bm[k].T + Bp[k] @ Hui[k + 1] @ Bp[k].T + Hxi[k + 1]
                     + hnui * np.eye(NX))
            if k + 1 < K:
                So[k] = -A_[k + 1] @ Hxi[k + 1] + Bm[k + 1] @ Hui[k + 1] @ Bp[k].T
        if CAPTURE is not None:      # oracle/bcr_proto.py: the block-tridiagonal Schur complements of a real solve, iteration by iteration
            CAPTURE.append((Sd.copy(), So.copy()))
        # block Cholesky
        L = np.zeros_like(Sd)
        Wb = np.zeros_like(So)
        for k in range(K):
            M = Sd[k].copy()
            if k > 0:
                M -= Wb[k - 1] @ Wb[k - 1].T
            L[k] = np.linalg.cholesky(M)
            if k + 1 < K:
                Wb[k] = np.linalg.solve(L[k], So[k].T).T

        def S_solve(r):
            t = np.zeros_like(r)
            for k in range(K):
                rr = r[k].copy()
                if k > 0:
                    rr -= Wb[k - 1] @ t[k - 1]
                t[k] = np.linalg.solve(L[k], rr)
            x = np.zeros_like(r)
            for k in range(K - 1, -1, -1):
                rr = t[k].copy()
                if k + 1 < K:
                    rr -= Wb[k].T @ x[k + 1]
                x[k] = np.linalg.solve(L[k].T, rr)
            return x

        def Hb_inv(g):
            return dict(dx=np.einsum("kij,kj->ki", Hxi, g["dx"]), du=np.einsum("kij,kj->ki", Hui, g["du"]), nu=hnui * g["nu"])

        def Eloc(dl):
            return (np.einsum("kij,kj->ki", A_, dl["dx"][:K]) + np.einsum("kij,kj->ki", Bm, dl["du"][:K])
                    + np.einsum("kij,kj->ki", Bp, dl["du"][1:]) + dl["nu"] - dl["dx"][1:])

        def Eloct(yy):
            g = dict(dx=np.zeros((K + 1, NX)), du=np.zeros((K + 1, NU)), nu=yy.copy())
            g["dx"][:K] += np.einsum("kij,ki->kj", A_, yy)
            g["dx"][1:] -= yy
            g["du"][:K] += np.einsum("kij,ki->kj", Bm, yy)
            g["du"][1:] += np.einsum("kij,ki->kj", Bp, yy)
            return g

        def band_solve(gl, ry_):
            """[Hb E'; E 0][dl; dy] = [gl; ry_]"""
            v = Hb_inv(gl)
            dy = S_solve(Eloc(v) - ry_)
            Ety_ = Eloct(dy)
            dl = Hb_inv({kk: gl[kk] - Ety_[kk] for kk in gl})
            return dl, dy

        def ldot(a_, b_):
            return float(np.sum(a_["dx"] * b_["dx"]) + np.sum(a_["du"] * b_["du"]) + np.sum(a_["nu"] * b_["nu"]))

        # border vectors (local part, E part)
        zl = dict(dx=np.zeros((K + 1, NX)), du=np.zeros((K + 1, NU)), nu=np.zeros((K, NX)))
        n14 = 14 * (K + 1)
        Ptr = dict(dx=np.where(fx, 0.0, ptr[:n14].reshape(K + 1, NX)), du=np.where(fu, 0.0, ptr[n14:].reshape(K + 1, NU)),
                   nu=np.zeros((K, NX)))
        Pnu = dict(dx=np.zeros((K + 1, NX)), du=np.zeros((K + 1, NU)), nu=pnu.reshape(K, NX).copy())
        sol_s = band_solve(zl, -Sg)            # column of s: local 0, E part Sg  -> K [x;y] = [0; -(-Sg)]...
        sol_tr = band_solve(Ptr, np.zeros((K, NX)))
        sol_nu = band_solve(Pnu, np.zeros((K, NX)))
        # global 2x2 from the sigma cone
        vsg, bsg = Wd["sg"][0]
        h00s, h01s, h11s, b2s = soc_Winv2_parts(vsg, bsg)
        Msg = np.array([[h00s, h01s * vsg[1]], [h01s * vsg[1], b2s + h11s * vsg[1] ** 2]])  # on (ts, s)
        hrk = 1.0 / Wd["rk"][0] ** 2

        def kkt_solve(gx, ry_):
            """Full reduced KKT: [H E'; E 0][dw; dy] = [gx; ry_] with gx variable-shaped dict."""
            # border unknowns b = (s, tnu, ttr, ts, atr, anu); local eq: Hb l + E' y + Ptr*(h01tr*ttr + atr) + Pnu*(h01nu*tnu + anu) = gx_l
            # E eq: E_loc l + Sg s = ry_
            gl = dict(dx=gx["dx"], du=gx["du"], nu=gx["nu"])
            l0, y0 = band_solve(gl, ry_)
            # l = l0 - ls*s - ltr*(h01tr ttr + atr) - lnu*(h01nu tnu + anu), same for y
            ls, ys = sol_s  # solves [0; -Sg]: so that contribution of +Sg*s on E rows moves to rhs as -Sg*s => l += ls*s
            ltr, ytr = sol_tr
            lnu, ynu = sol_nu
            # unknown combos: ctr = h01tr*ttr + atr ; cnu = h01nu*tnu + anu
            # border equations:
            #  s  : Msg[1,1]*s + Msg[1,0]*ts + Sg . y = gx_s
            #  ts : Msg[0,0]*ts + Msg[0,1]*s = gx_ts
            #  tnu: h00nu*tnu + h01nu * (Pnu . l) = gx_tnu
            #  ttr: (h00tr + hrk)*ttr + h01tr * (Ptr . l) = gx_ttr
            #  atr: Ptr . l - atr / h11tr = 0
            #  anu: Pnu . l - anu / h11nu = 0
            # with l = l0 + ls*s - ltr*ctr - lnu*cnu ; y = y0 + ys*s - ytr*ctr - ynu*cnu
            M = np.zeros((6, 6))
            r = np.zeros(6)
            # unknown order: s, ts, tnu, ttr, atr, anu
            def lin(vec_l=None, vec_y=None):
                """coefficients of (const, s, ctr, cnu) of P.l or Sg.y"""
                if vec_l is not None:
                    return ldot(vec_l, l0), ldot(vec_l, ls), -ldot(vec_l, ltr), -ldot(vec_l, lnu)
                return float(np.sum(vec_y * y0)), float(np.sum(vec_y * ys)), -float(np.sum(vec_y * ytr)), -float(np.sum(vec_y * ynu))
            c0, cs, ct, cn = lin(vec_y=Sg)
            M[0, 0] = Msg[1, 1] + cs
            M[0, 1] = Msg[1, 0]
            M[0, 3] += ct * h_tr[1]
            M[0, 4] += ct
            M[0, 2] += cn * h_nu[1]
            M[0, 5] += cn
            r[0] = gx["s"] - c0
            M[1, 1] = Msg[0, 0]
            M[1, 0] = Msg[0, 1]
            r[1] = gx["ts"]
            c0, cs, ct, cn = lin(vec_l=Pnu)
            M[2, 2] = h_nu[0] + h_nu[1] * (cn * h_nu[1])
            M[2, 0] = h_nu[1] * cs
            M[2, 3] = h_nu[1] * ct * h_tr[1]
            M[2, 4] = h_nu[1] * ct
            M[2, 5] = h_nu[1] * cn
            r[2] = gx["tnu"] - h_nu[1] * c0
            M[5, 0] = cs
            M[5, 2] = cn * h_nu[1]
            M[5, 3] = ct * h_tr[1]
            M[5, 4] = ct
            M[5, 5] = cn - 1.0 / h_nu[2]
            r[5] = -c0
            c0, cs, ct, cn = lin(vec_l=Ptr)
            M[3, 3] = h_tr[0] + hrk + h_tr[1] * (ct * h_tr[1])
            M[3, 0] = h_tr[1] * cs
            M[3, 2] = h_tr[1] * cn * h_nu[1]
            M[3, 4] = h_tr[1] * ct
            M[3, 5] = h_tr[1] * cn
            r[3] = gx["ttr"] - h_tr[1] * c0
            M[4, 0] = cs
            M[4, 2] = cn * h_nu[1]
            M[4, 3] = ct * h_tr[1]
            M[4, 4] = ct - 1.0 / h_tr[2]
            M[4, 5] = cn
            r[4] = -c0
            b = np.linalg.solve(M, r)
            s_, ts_, tnu_, ttr_, atr_, anu_ = b
            ctr = h_tr[1] * ttr_ + atr_
            cnu = h_nu[1] * tnu_ + anu_
            dw = {kk: l0[kk] + ls[kk] * s_ - ltr[kk] * ctr - lnu[kk] * cnu for kk in l0}
            dy = y0 + ys * s_ - ytr * ctr - ynu * cnu
            dw.update(s=s_, ts=ts_, tnu=tnu_, ttr=ttr_)
            return dw, dy
        return kkt_solve

    # ---- initial point (CVXOPT coneqp initialisation with W = I): least-squares slack, then shift ----
    Wid = {}
    for n, cnt, dim in C.spec:
        if dim == 1:
            Wid[n] = np.ones(cnt)
        else:
            v0 = np.zeros(dim)
            v0[0] = 1.0
            Wid[n] = [(v0, 1.0)] * cnt
    kkt0 = build_kkt(Wid)
    a0 = C.affine(V)
    Jta = C.jact_apply(a0)
    gx0 = {kk: -cost[kk] - Jta[kk] for kk in cost}
    gx0["dx"] = np.where(fx, 0.0, gx0["dx"])
    gx0["du"] = np.where(fu, 0.0, gx0["du"])
    dw0, y = kkt0(gx0, -(E_apply(V) + dk))
    for kk in V:
        V[kk] = V[kk] + dw0[kk]
    a = C.affine(V)
    S = {n: a[n].copy() for n in a}
    Z = {n: -a[n] for n in a}

    def shift(X):
        t = -np.inf
        for n, cnt, dim in C.spec:
            for i in range(cnt):
                xx = X[n][i]
                t = max(t, (np.linalg.norm(xx[1:]) if dim > 1 else 0.0) - xx[0])
        if t >= -1e-8:
            for n in X:
                X[n][:, 0] += 1.0 + t

    shift(S)
    shift(Z)

    def dot(Aa, Bb):
        return sum(float(np.sum(Aa[n] * Bb[n])) for n in Aa)

    status = "max_iter"
    best = (np.inf, None, 0)
    for it in range(1, max_iter + 1):
        a = C.affine(V)
        rz = {n: S[n] - a[n] for n in S}
        JtZ = C.jact_apply(Z)
        Ety = Et_apply(y)
        rx = {kk: cost[kk] - JtZ[kk] + Ety[kk] for kk in cost}
        rx["dx"] = np.where(fx, 0.0, rx["dx"])
        rx["du"] = np.where(fu, 0.0, rx["du"])
        ry = E_apply(V) + dk
        gap = dot(S, Z)
        pobj = -V["dx"][K, 0] + p.wNu * V["tnu"] + 0.5 * V["ttr"] + V["ts"]
        nrx = np.sqrt(sum(float(np.sum(np.square(rx[kk]))) for kk in rx))
        nrz = np.sqrt(sum(float(np.sum(np.square(rz[n]))) for n in rz))
        pres = max(np.linalg.norm(ry), nrz)
        dres = nrx / max(1.0, p.wNu)
        relgap = gap / max(1.0, abs(pobj))
        if verbose:
            print(f"{it:3d} pobj {pobj:+.8e} gap {gap:.2e} pres {pres:.2e} dres {dres:.2e}")
        merit = max(pres, dres, relgap)
        if merit < best[0]:
            best = (merit, {kk: np.copy(V[kk]) for kk in V}, it)
        if pres < tol and dres < tol and relgap < tol:
            status = "optimal"
            break
        if it - best[2] >= 3 and best[0] < 1e-5:   # endgame only: no progress for three iterations = numerical floor
            status = "optimal" if best[0] < 100 * tol else "stalled"
            V = best[1]
            break
        # ---- NT scalings ----
        Wd = {}
        lam = C.zeros()
        for n, cnt, dim in C.spec:
            if dim == 1:
                Wd[n] = np.sqrt(S[n][:, 0] / Z[n][:, 0])
                lam[n][:, 0] = np.sqrt(S[n][:, 0] * Z[n][:, 0])
            else:
                Wd[n] = [soc_nt(S[n][i], Z[n][i]) for i in range(cnt)]
                for i in range(cnt):
                    lam[n][i] = soc_W(*Wd[n][i], Z[n][i])

        def W_apply(X, inverse=False):
            out = C.zeros()
            for n, cnt, dim in C.spec:
                if dim == 1:
                    out[n][:, 0] = X[n][:, 0] / Wd[n] if inverse else X[n][:, 0] * Wd[n]
                else:
                    for i in range(cnt):
                        out[n][i] = soc_W(*Wd[n][i], X[n][i], inverse)
            return out

        try:
            kkt_solve = build_kkt(Wd)
        except np.linalg.LinAlgError:
            status = "optimal" if best[0] < 100 * tol else "kkt_failed"
            V = best[1]
            break

        def newton(ds_rhs):
            # bz = -rz - W (lam \ ds_rhs);  H dw + E' dy = -rx - J' W^-2 bz ; E dw = -ry
            t = C.zeros()
            for n, cnt, dim in C.spec:
                if dim == 1:
                    t[n][:, 0] = ds_rhs[n][:, 0] / lam[n][:, 0]
                else:
                    for i in range(cnt):
                        t[n][i] = soc_div(lam[n][i], ds_rhs[n][i])
            # W^-1 bz = -W^-1 rz - (lam \ ds_rhs): never round-trip through W (keeps digits near the boundary)
            Wirz = W_apply(rz, inverse=True)
            Wibz = {n: -Wirz[n] - t[n] for n in rz}
            JtW = C.jact_apply(W_apply(Wibz, inverse=True))
            gx = {kk: -rx[kk] - JtW[kk] for kk in rx}
            gx["dx"] = np.where(fx, 0.0, gx["dx"])
            gx["du"] = np.where(fu, 0.0, gx["du"])
            dw, dy = kkt_solve(gx, -ry)
            for _ in range(refine):
                # residual of [H E'; E 0][dw; dy] = [gx; -ry] in operator form, then one correction solve
                Hdw = C.jact_apply(W_apply(W_apply(C.jac_apply(dw), inverse=True), inverse=True))
                Etd = Et_apply(dy)
                r1 = {kk: gx[kk] - Hdw[kk] - Etd[kk] for kk in gx}
                r1["dx"] = np.where(fx, 0.0, r1["dx"])
                r1["du"] = np.where(fu, 0.0, r1["du"])
                r2 = -ry - E_apply(dw)
                cw, cy = kkt_solve(r1, r2)
                dw = {kk: dw[kk] + cw[kk] for kk in dw}
                dy = dy + cy
            Jdw = C.jac_apply(dw)
            if verbose > 1:
                Hdw = C.jact_apply(W_apply(W_apply(Jdw, inverse=True), inverse=True))
                Etd = Et_apply(dy)
                res = {kk: Hdw[kk] + Etd[kk] - gx[kk] for kk in gx}
                res["dx"] = np.where(fx, 0.0, res["dx"]); res["du"] = np.where(fu, 0.0, res["du"])
                print("   kkt res:", {kk: float(np.abs(res[kk]).max()) for kk in res}, "E:", float(np.abs(E_apply(dw) + ry).max()))
            WiJ = W_apply(Jdw, inverse=True)
            dz_ = W_apply({n: WiJ[n] + Wibz[n] for n in WiJ}, inverse=True)
            dz = {n: -dz_[n] for n in dz_}
            ds = {n: -rz[n] + Jdw[n] for n in rz}
            return dw, dy, dz, ds

        def maxstep(dS, dZ):
            sds = W_apply(dS, inverse=True)
            sdz = W_apply(dZ)
            amax = np.inf
            for n, cnt, dim in C.spec:
                for dd in (sds, sdz):
                    if dim == 1:
                        neg = dd[n][:, 0] < 0
                        if neg.any():
                            amax = min(amax, np.min(-lam[n][neg, 0] / dd[n][neg, 0]))
                    else:
                        for i in range(cnt):
                            amax = min(amax, soc_maxstep(lam[n][i], dd[n][i]))
            return amax, sds, sdz

        mu = gap / C.degree
        ll = C.zeros()
        for n, cnt, dim in C.spec:
            if dim == 1:
                ll[n][:, 0] = lam[n][:, 0] ** 2
            else:
                for i in range(cnt):
                    ll[n][i] = soc_prod(lam[n][i], lam[n][i])
        dsa = {n: -ll[n] for n in ll}
        try:
            with np.errstate(all="ignore"):
                dw, dy, dz, ds = newton(dsa)
            if not all(np.isfinite(np.asarray(dw[kk])).all() for kk in dw):
                raise np.linalg.LinAlgError("non-finite direction")
        except np.linalg.LinAlgError:
            status = "optimal" if best[0] < 100 * tol else "kkt_failed"
            V = best[1]
            break
        amax, sds, sdz = maxstep(ds, dz)
        alpha = min(1.0, amax)
        sig = (1 - alpha) ** 3
        comb = C.zeros()
        e = C.identity()
        for n, cnt, dim in C.spec:
            if dim == 1:
                comb[n][:, 0] = dsa[n][:, 0] - sds[n][:, 0] * sdz[n][:, 0] + sig * mu
            else:
                for i in range(cnt):
                    comb[n][i] = dsa[n][i] - soc_prod(sds[n][i], sdz[n][i]) + sig * mu * e[n][i]
        try:
            with np.errstate(all="ignore"):
                dw, dy, dz, ds = newton(comb)
            if not all(np.isfinite(np.asarray(dw[kk])).all() for kk in dw):
                raise np.linalg.LinAlgError("non-finite direction")
        except np.linalg.LinAlgError:
            status = "optimal" if best[0] < 100 * tol else "kkt_failed"
            V = best[1]
            break
        amax, sds, sdz = maxstep(ds, dz)
        alpha = min(1.0, 0.99 * amax)
        if verbose > 1:
            print("   alpha", alpha, "sig", sig, "mu", mu, "dw: s", dw["s"], "tnu", dw["tnu"], "ttr", dw["ttr"], "ts", dw["ts"])
        if alpha < 1e-9:
            status = "optimal" if best[0] < 100 * tol else "stalled"
            V = best[1]
            break
        for kk in V:
            V[kk] = V[kk] + alpha * dw[kk]
        y = y + alpha * dy
        for n in S:
            S[n] = S[n] + alpha * ds[n]
            Z[n] = Z[n] + alpha * dz[n]
    V["status"], V["iters"] = status, it
    V["pobj"] = -V["dx"][K, 0] + p.wNu * V["tnu"] + 0.5 * V["ttr"] + V["ts"]
    return V
