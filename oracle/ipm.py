"""Primal-dual interior-point solver for cone programs (oracle; see oracle/__init__.py for the rules).

Plays the role of the conic solver the reference reaches through MathOptInterface
(Mosek at rocketland.jl:58-59, ECOS imported at rocketland.jl:4).  Neither is vendored, pinned or
available here, so this module restates their published algorithm class: an infeasible-start
Mehrotra predictor-corrector method with Nesterov-Todd scaling on

        minimise    c'x
        subject to  A x = b,   G x + s = h,   s in K = R+^l x Q^{q_1} x ... x Q^{q_N}

(Andersen, Roos, Terlaky 2003 — the MOSEK conic optimiser; Domahidi, Chu, Boyd 2013 — ECOS;
Vandenberghe 2010 — "The CVXOPT linear and quadratic cone program solvers", whose notation is used).
The optimum of an SOCP does not depend on which of these produced it, only on the tolerance, which
is why parity is defined against the optimum.

Second-order-cone scalings W = beta (2 v v' - J) are kept sparse in the KKT matrix through the
identity W^2 = beta^2 (I + [v Jv] M [v Jv]') with M = [[4 v'v, -2], [-2, 0]], expanded with two
auxiliary unknowns per cone (the device of ECOS §III-D), so that one sparse LU per iteration serves
the 868- and 715-dimensional trust-region cones of the SCvx subproblem.
"""
from dataclasses import dataclass
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


@dataclass
class ConeSolution:
    x: np.ndarray
    y: np.ndarray
    z: np.ndarray
    s: np.ndarray
    status: str
    iters: int
    pres: float
    dres: float
    gap: float
    pobj: float


class Cone:
    """K = R+^l x Q^{q_1} x ... ; everything vectorised with reduceat over SOC blocks."""

    def __init__(self, l, q):
        self.l = int(l)
        self.q = np.asarray(q, dtype=np.int64)
        self.m = self.l + int(self.q.sum())
        self.nq = len(self.q)
        self.off = self.l + np.concatenate([[0], np.cumsum(self.q)[:-1]]).astype(np.int64) if self.nq else np.zeros(0, np.int64)
        self.blk = np.repeat(np.arange(self.nq), self.q)  # block id of every SOC row
        self.head = np.zeros(self.m - self.l, dtype=bool)
        if self.nq:
            self.head[self.off - self.l] = True
        self.degree = self.l + self.nq

    # ---- helpers on SOC parts (arrays of length m - l) ----
    def _sum(self, a):
        return np.add.reduceat(a, self.off - self.l) if self.nq else np.zeros(0)

    def jdot(self, a, b):
        """a' J b per SOC block."""
        pa = a * b
        tot = self._sum(pa)
        h = pa[self.head]
        return 2 * h - tot

    def e(self):
        v = np.zeros(self.m)
        v[: self.l] = 1.0
        v[self.off] = 1.0
        return v

    def prod(self, a, b):
        """Jordan product a o b."""
        out = np.empty(self.m)
        out[: self.l] = a[: self.l] * b[: self.l]
        if self.nq:
            aq, bq = a[self.l:], b[self.l:]
            dot = self._sum(aq * bq)
            a0, b0 = aq[self.head], bq[self.head]
            r = a0[self.blk] * bq + b0[self.blk] * aq
            r[self.head] = dot
            out[self.l:] = r
        return out

    def div(self, lam, d):
        """solve lam o x = d."""
        out = np.empty(self.m)
        out[: self.l] = d[: self.l] / lam[: self.l]
        if self.nq:
            lq, dq = lam[self.l:], d[self.l:]
            l0, d0 = lq[self.head], dq[self.head]
            l1d1 = self._sum(lq * dq) - l0 * d0
            det = 2 * l0 * l0 - self._sum(lq * lq)  # l0^2 - |l1|^2
            x0 = (l0 * d0 - l1d1) / det
            x = (dq - x0[self.blk] * lq) / l0[self.blk]
            x[self.head] = x0
            out[self.l:] = x
        return out

    def max_step(self, lam, d):
        """largest alpha in (0, inf] with lam + alpha d in K (lam strictly inside)."""
        amax = np.inf
        if self.l:
            neg = d[: self.l] < 0
            if neg.any():
                amax = min(amax, np.min(-lam[: self.l][neg] / d[: self.l][neg]))
        if self.nq:
            lq, dq = lam[self.l:], d[self.l:]
            a = self.jdot(dq, dq)
            b = 2 * self.jdot(lq, dq)
            c = self.jdot(lq, lq)
            l0, d0 = lq[self.head], dq[self.head]
            # head must stay nonnegative
            neg = d0 < 0
            if neg.any():
                amax = min(amax, np.min(-l0[neg] / d0[neg]))
            # smallest positive root of a t^2 + b t + c = 0 (c > 0)
            disc = b * b - 4 * a * c
            with np.errstate(divide="ignore", invalid="ignore"):
                sq = np.sqrt(np.maximum(disc, 0.0))
                qq = -0.5 * (b + np.where(b >= 0, 1.0, -1.0) * sq)
                r1 = np.where(qq != 0, c / qq, np.inf)
                r2 = np.where(a != 0, qq / a, np.inf)
            roots = np.stack([r1, r2])
            roots = np.where((roots > 0) & (disc >= 0)[None, :], roots, np.inf)
            amax = min(amax, roots.min())
        return amax

    def interior_shift(self, s):
        """min t such that s + t e is in K (negative if strictly inside)."""
        t = -np.inf
        if self.l:
            t = max(t, np.max(-s[: self.l]))
        if self.nq:
            sq = s[self.l:]
            s0 = sq[self.head]
            n1 = np.sqrt(np.maximum(self._sum(sq * sq) - s0 * s0, 0.0))
            t = max(t, np.max(n1 - s0))
        return t

    # ---- Nesterov-Todd scaling ----
    def nt(self, s, z):
        W = {}
        W["d"] = np.sqrt(s[: self.l] / z[: self.l])  # W_l = diag(d)
        if self.nq:
            sq, zq = s[self.l:], z[self.l:]
            sj = np.sqrt(self.jdot(sq, sq))
            zj = np.sqrt(self.jdot(zq, zq))
            sb = sq / sj[self.blk]
            zb = zq / zj[self.blk]
            gam = np.sqrt((1 + self._sum(sb * zb)) / 2)
            Jzb = -zb.copy()
            Jzb[self.head] = zb[self.head]
            wb = (sb + Jzb) / (2 * gam[self.blk])
            v = wb.copy()
            v[self.head] += 1.0
            v = v / np.sqrt(2 * (wb[self.head] + 1.0))[self.blk]
            W["v"] = v
            W["beta"] = np.sqrt(sj / zj)
        return W

    def apply_W(self, W, x, inverse=False):
        """W x (or W^{-1} x); W symmetric for both cone types."""
        out = np.empty(self.m)
        out[: self.l] = x[: self.l] / W["d"] if inverse else x[: self.l] * W["d"]
        if self.nq:
            v, beta = W["v"], W["beta"]
            xq = x[self.l:]
            if not inverse:
                vx = self._sum(v * xq)
                Jx = -xq.copy()
                Jx[self.head] = xq[self.head]
                out[self.l:] = beta[self.blk] * (2 * vx[self.blk] * v - Jx)
            else:  # W^{-1} = (1/beta) (2 Jv (Jv)' - J)
                Jv = -v.copy()
                Jv[self.head] = v[self.head]
                vx = self._sum(Jv * xq)
                Jx = -xq.copy()
                Jx[self.head] = xq[self.head]
                out[self.l:] = (2 * vx[self.blk] * Jv - Jx) / beta[self.blk]
        return out


def _kkt_factor(A, G, cone, W, n, p, reg):
    """LU of [[reg I, A', G', 0], [A, -reg I, 0, 0], [G, 0, -D, P], [0, 0, P', E]]."""
    m = cone.m
    D = np.empty(m)
    D[: cone.l] = W["d"] ** 2
    blocks_P = None
    E = None
    if cone.nq:
        v, beta = W["v"], W["beta"]
        b2 = beta**2
        D[cone.l:] = b2[cone.blk]
        a = cone._sum(v * v)
        Jv = -v.copy()
        Jv[cone.head] = v[cone.head]
        # eigen-decomposition of M = [[4a, -2], [-2, 0]]
        rt = np.sqrt(4 * a * a + 4)
        mu1, mu2 = 2 * a + rt, 2 * a - rt
        # eigenvectors (unnormalised): [mu, -2]
        n1 = np.sqrt(mu1**2 + 4)
        n2 = np.sqrt(mu2**2 + 4)
        p1 = (mu1 / n1)[cone.blk] * v + (-2 / n1)[cone.blk] * Jv
        p2 = (mu2 / n2)[cone.blk] * v + (-2 / n2)[cone.blk] * Jv
        rows = np.concatenate([np.arange(cone.l, m), np.arange(cone.l, m)])
        cols = np.concatenate([2 * cone.blk, 2 * cone.blk + 1])
        blocks_P = sp.csc_matrix((np.concatenate([p1, p2]), (rows, cols)), shape=(m, 2 * cone.nq))
        E = np.empty(2 * cone.nq)
        E[0::2] = 1.0 / (b2 * mu1)
        E[1::2] = 1.0 / (b2 * mu2)
    nt = 2 * cone.nq
    In = sp.identity(n, format="csc") * reg
    Ip = sp.identity(p, format="csc") * (-reg)
    rowsK = [
        [In, A.T, G.T, None if nt == 0 else sp.csc_matrix((n, nt))],
        [A, Ip, None, None],
        [G, None, -sp.diags(D), blocks_P],
    ]
    if nt:
        rowsK.append([None, None, blocks_P.T, sp.diags(E)])
    else:
        rowsK = [r[:3] for r in rowsK]
    Kmat = sp.bmat(rowsK, format="csc")
    return spla.splu(Kmat, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.01), Kmat, nt


def _kkt_solve(lu, Kmat, n, p, m, nt, bx, by, bz, refine=2):
    rhs = np.concatenate([bx, by, bz, np.zeros(nt)])
    sol = lu.solve(rhs)
    for _ in range(refine):
        r = rhs - Kmat @ sol
        sol += lu.solve(r)
    return sol[:n], sol[n:n + p], sol[n + p:n + p + m]


def solve(c, A, b, G, h, l, q, tol=1e-9, max_iter=100, verbose=False) -> ConeSolution:
    c = np.asarray(c, float)
    b = np.asarray(b, float)
    h = np.asarray(h, float)
    A = sp.csc_matrix(A)
    G = sp.csc_matrix(G)
    n, p = c.size, b.size
    cone = Cone(l, q)
    m = cone.m
    e = cone.e()
    reg = 1e-10

    # ---- initial point (CVXOPT coneqp §initialisation with P = 0) ----
    W0 = {"d": np.ones(cone.l)}
    if cone.nq:
        v0 = np.zeros(m - cone.l)
        v0[cone.head] = 1.0
        W0["v"] = v0
        W0["beta"] = np.ones(cone.nq)
    lu, Kmat, nt = _kkt_factor(A, G, cone, W0, n, p, reg)
    x, y, z = _kkt_solve(lu, Kmat, n, p, m, nt, -c, b, h)
    s = -z.copy()
    ts = cone.interior_shift(s)
    if ts >= -1e-8 * max(1.0, np.linalg.norm(s)):
        s = s + (1.0 + ts) * e
    tz = cone.interior_shift(z)
    if tz >= -1e-8 * max(1.0, np.linalg.norm(z)):
        z = z + (1.0 + tz) * e

    nrm_c, nrm_b, nrm_h = max(1.0, np.linalg.norm(c)), max(1.0, np.linalg.norm(b)), max(1.0, np.linalg.norm(h))
    status = "max_iter"
    pres = dres = gap = np.inf
    it = 0
    for it in range(1, max_iter + 1):
        rx = A.T @ y + G.T @ z + c
        ry = A @ x - b
        rz = G @ x + s - h
        gap = float(s @ z)
        pobj = float(c @ x)
        dobj = float(-b @ y - h @ z)
        pres = max(np.linalg.norm(ry) / nrm_b, np.linalg.norm(rz) / nrm_h)
        dres = np.linalg.norm(rx) / nrm_c
        relgap = gap / max(1.0, abs(pobj), abs(dobj))
        if verbose:
            print(f"{it:3d} pobj {pobj:+.8e} dobj {dobj:+.8e} gap {gap:.2e} pres {pres:.2e} dres {dres:.2e}")
        if pres < tol and dres < tol and (gap < tol or relgap < tol):
            status = "optimal"
            break
        # numerical floor: the iterate sits on the cone boundary to rounding; accept it if it is a
        # certified near-optimum (what Mosek/ECOS report as OPTIMAL at their default 1e-8 tolerances)
        near = pres < 10 * tol and dres < 10 * tol and relgap < 100 * tol
        with np.errstate(all="ignore"):
            W = cone.nt(s, z)
            lam = cone.apply_W(W, z)
        ok = all(np.isfinite(v).all() for v in W.values()) and np.isfinite(lam).all()
        if ok:
            try:
                lu, Kmat, nt = _kkt_factor(A, G, cone, W, n, p, reg)
            except RuntimeError:
                ok = False
        if not ok:
            status = "optimal" if near else "kkt_singular"
            break

        def newton(ds_rhs):
            # G dx - W'W dz = -rz - W (lam \ ds_rhs)
            t = cone.apply_W(W, cone.div(lam, ds_rhs))
            dx, dy, dz = _kkt_solve(lu, Kmat, n, p, m, nt, -rx, -ry, -rz - t)
            ds = -rz - G @ dx
            return dx, dy, dz, ds

        mu = gap / cone.degree
        # affine (predictor) direction
        dsa_rhs = -cone.prod(lam, lam)
        dx, dy, dz, ds = newton(dsa_rhs)
        sds = cone.apply_W(W, ds, inverse=True)
        sdz = cone.apply_W(W, dz)
        alpha = min(1.0, cone.max_step(lam, sds), cone.max_step(lam, sdz))
        sigma = (1.0 - alpha) ** 3
        # combined direction
        comb = dsa_rhs - cone.prod(sds, sdz) + sigma * mu * e
        dx, dy, dz, ds = newton(comb)
        sds = cone.apply_W(W, ds, inverse=True)
        sdz = cone.apply_W(W, dz)
        alpha = min(1.0, 0.99 * min(cone.max_step(lam, sds), cone.max_step(lam, sdz)))
        if alpha < 1e-8:
            status = "optimal" if near else "stalled"
            break
        x = x + alpha * dx
        y = y + alpha * dy
        z = z + alpha * dz
        s = s + alpha * ds
        if not (np.isfinite(x).all() and np.isfinite(z).all()):
            status = "nonfinite"
            break
    return ConeSolution(x, y, z, s, status, it, float(pres), float(dres), float(gap), float(c @ x))
