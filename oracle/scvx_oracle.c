/*
 * scvx_oracle.c — CPU restatement (plain C, fp64) of the reference's discretisation path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this.  The product (successiveconvexification_amd/, libscvx_hip.so) never links, imports
 * or calls anything in oracle/.
 *
 * PARITY UNPINNED: the reference (Julia) cannot run in this container, ships no tests, golden vectors
 * or recorded outputs for this path, and its numeric engines are un-vendored,
 * un-versioned Julia packages (DifferentialEquations BS3, DiffEqSensitivity, Zygote, Mosek/ECOS).
 * This file restates the reference's own equations and the published algorithms it delegates to;
 * it is validated by mathematical self-consistency (tests/test_oracle_*.py), not by reference output.
 *
 * What follows the reference (file:line into /root/reference):
 *   DCM                      dynamics.jl:29-44
 *   Omega                    dynamics.jl:46-52
 *   dx_static (RHS)          dynamics.jl:54-77  (tau_aero == 0, dynamics.jl:69)
 *   FOH control              dynamics.jl:108-110, 144-150
 *   rk4, npts substeps       dynamics.jl:112-134  WITHOUT its stage bug (:126-128 omit idt)
 *   sensitivity semantics    dynamics.jl:298-305, 321-334; autodiff_dynamics.jl:74-92
 *                            derivative = d x(dt) / d [x_k; u_k; u_{k+1}; sigma]  (14x21, column-major)
 *   aero_force (symbolic)    aerodynamics.jl:60-77 + shims dynamics.jl:162-207
 *   cubic B-spline tables    aerodynamics.jl:17-21 (Interpolations.jl Cubic(Line(OnGrid())), Flat())
 *
 * FIN EXTENSION (control_dim = 5; BUILD-DEFINED).  The reference carries the fin model only as commented-out
 * code: fd1 = normalize((C(q) e2) x v), fd2 = fd1 x v are computed and unused (dynamics.jl:60-62); the force
 * ff = u[4] fd1 + u[5] fd2 (:63), its place in the acceleration (:66 "aerf + ff") and the torque cross(rFB, ff) (:69) sit in
 * #= =# comments.  With p->nu == 5 this file enables exactly those three commented expressions (bdy_trq stays zero, as in
 * the live model); with nu == 3 nothing changes.  fd1 is guarded like the reference's ifnz: (C e2) x v = 0 gives no fin force.
 *
 * The Jacobian of the discrete RK4 map is obtained by integrating the variational equations with the
 * same RK4 tableau, which is identical to differentiating the discrete map (what
 * sensitivity_zygote, dynamics.jl:311-313, asks forward-mode AD to do).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NX 14
#define NU_MAX 5
#define NP_MAX 25 /* 14 + 2 * 5 + 1 */

typedef struct {
    double alpha, g0, sos;
    double J[9], Jinv[9]; /* column-major 3x3 */
    double rTB[3], rFB[3];
    int32_t aero_kind; /* 0 exo, 1 atmospheric */
    int32_t n_aoa, n_mach;
    int32_t pad;
    double aoa0, daoa, mach0, dmach;
    double force_scalar, length_scalar;
    const double *cdrag; /* prefiltered B-spline coefficients, (n_aoa+2) x (n_mach+2), aoa fastest */
    const double *clift;
    int32_t nu;   /* control_dim: 3 (the live model) or 5 (fin extension) */
    int32_t pad2;
} oracle_params;
static int nu_of(const oracle_params *p) { return p->nu == 5 ? 5 : 3; }

/* ---------- small helpers ---------- */
static void dcm(const double q[4], double C[9] /* row-major */) {
    /* dynamics.jl:29-44 */
    double q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    C[0] = 1 - 2 * (q2 * q2 + q3 * q3);
    C[1] = 2 * (q1 * q2 - q0 * q3);
    C[2] = 2 * (q1 * q3 + q0 * q2);
    C[3] = 2 * (q1 * q2 + q0 * q3);
    C[4] = 1 - 2 * (q1 * q1 + q3 * q3);
    C[5] = 2 * (q2 * q3 - q0 * q1);
    C[6] = 2 * (q1 * q3 - q0 * q2);
    C[7] = 2 * (q2 * q3 + q0 * q1);
    C[8] = 1 - 2 * (q1 * q1 + q2 * q2);
}
/* d(C u)/dq : 3x4 row-major */
static void dcm_u_dq(const double q[4], const double u[3], double D[12]) {
    double q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    double u1 = u[0], u2 = u[1], u3 = u[2];
    /* column 0: d/dq0 */
    D[0 * 4 + 0] = 2 * (-q3 * u2 + q2 * u3);
    D[1 * 4 + 0] = 2 * (q3 * u1 - q1 * u3);
    D[2 * 4 + 0] = 2 * (-q2 * u1 + q1 * u2);
    /* d/dq1 */
    D[0 * 4 + 1] = 2 * (q2 * u2 + q3 * u3);
    D[1 * 4 + 1] = 2 * (q2 * u1 - 2 * q1 * u2 - q0 * u3);
    D[2 * 4 + 1] = 2 * (q3 * u1 + q0 * u2 - 2 * q1 * u3);
    /* d/dq2 */
    D[0 * 4 + 2] = 2 * (-2 * q2 * u1 + q1 * u2 + q0 * u3);
    D[1 * 4 + 2] = 2 * (q1 * u1 + q3 * u3);
    D[2 * 4 + 2] = 2 * (-q0 * u1 + q3 * u2 - 2 * q2 * u3);
    /* d/dq3 */
    D[0 * 4 + 3] = 2 * (-2 * q3 * u1 - q0 * u2 + q1 * u3);
    D[1 * 4 + 3] = 2 * (q0 * u1 - 2 * q3 * u2 + q2 * u3);
    D[2 * 4 + 3] = 2 * (q1 * u1 + q2 * u2);
}
static void cross3(const double a[3], const double b[3], double c[3]) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
static void mat3v(const double M[9] /* column-major */, const double v[3], double o[3]) {
    for (int i = 0; i < 3; i++) o[i] = M[i] * v[0] + M[3 + i] * v[1] + M[6 + i] * v[2];
}

/* ---------- cubic B-spline (Interpolations.jl Cubic(Line(OnGrid()))) ---------- */
/* value and gradient of the tensor-product cubic B-spline with coefficient grid c ((na+2) x (nm+2),
 * index (i+1, j+1) is grid node (i, j)), at continuous grid coordinates (ta, tm) in [0,na-1]x[0,nm-1]. */
static void bspline_w(double t, int n, int *i0, double w[4], double dw[4]) {
    int i = (int)floor(t);
    if (i < 0) i = 0;
    if (i > n - 2) i = n - 2;
    double d = t - i;
    double d2 = d * d, d3 = d2 * d;
    /* uniform cubic B-spline basis on nodes i-1..i+2 */
    w[0] = (1 - 3 * d + 3 * d2 - d3) / 6.0;
    w[1] = (4 - 6 * d2 + 3 * d3) / 6.0;
    w[2] = (1 + 3 * d + 3 * d2 - 3 * d3) / 6.0;
    w[3] = d3 / 6.0;
    dw[0] = (-3 + 6 * d - 3 * d2) / 6.0;
    dw[1] = (-12 * d + 9 * d2) / 6.0;
    dw[2] = (3 + 6 * d - 9 * d2) / 6.0;
    dw[3] = (3 * d2) / 6.0;
    *i0 = i; /* coefficient index of node i-1 is i (because of the +1 padding) */
}
/* table lookup with Flat() extrapolation: value, d/d(aoa), d/d(mach) in axis units */
static void table_eval(const oracle_params *p, const double *c, double aoa, double mach, double out[3]) {
    int na = p->n_aoa, nm = p->n_mach;
    double ta = (aoa - p->aoa0) / p->daoa, tm = (mach - p->mach0) / p->dmach;
    int flat_a = 0, flat_m = 0;
    if (ta < 0) { ta = 0; flat_a = 1; }
    if (ta > na - 1) { ta = na - 1; flat_a = 1; }
    if (tm < 0) { tm = 0; flat_m = 1; }
    if (tm > nm - 1) { tm = nm - 1; flat_m = 1; }
    int ia, im;
    double wa[4], dwa[4], wm[4], dwm[4];
    bspline_w(ta, na, &ia, wa, dwa);
    bspline_w(tm, nm, &im, wm, dwm);
    double v = 0, va = 0, vm = 0;
    int lda = na + 2;
    for (int b = 0; b < 4; b++) {
        double s = 0, sa = 0;
        for (int a = 0; a < 4; a++) {
            double cc = c[(size_t)(im + b) * lda + (ia + a)];
            s += wa[a] * cc;
            sa += dwa[a] * cc;
        }
        v += wm[b] * s;
        va += wm[b] * sa;
        vm += dwm[b] * s;
    }
    out[0] = v;
    out[1] = flat_a ? 0.0 : va / p->daoa;
    out[2] = flat_m ? 0.0 : vm / p->dmach;
}

/* ---------- aerodynamic force, live (symbolic-derived) model: aerodynamics.jl:60-77 ---------- */
/* F[3] and, if dF != NULL, dF/d(q[4], v[3]) as 3x7 row-major (cols: q0..q3, v1..v3). */
static void aero_force(const oracle_params *p, const double q[4], const double v[3], double F[3], double *dF) {
    F[0] = F[1] = F[2] = 0;
    if (dF) memset(dF, 0, sizeof(double) * 21);
    if (p->aero_kind == 0) return;
    double C[9];
    dcm(q, C);
    double bv[3] = {C[0], C[3], C[6]}; /* C(q) e1 */
    double vn2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    double vn = sqrt(vn2);
    if (!(vn > 0)) return; /* ifnz(drag_norm, ...) and mach<=0 -> everything 0 (dynamics.jl:162-168,198-204) */
    double c = bv[0] * v[0] + bv[1] * v[1] + bv[2] * v[2];
    double mach = vn / p->sos;
    double arg = c / (mach * p->sos); /* clamp_aoa dynamics.jl:162-168 */
    int clamped = 0;
    if (arg < -1.0) { arg = -1.0; clamped = 1; }
    if (arg > 1.0) { arg = 1.0; clamped = 1; }
    double td[3], tl[3];
    table_eval(p, p->cdrag, arg, mach, td);
    table_eval(p, p->clift, arg, mach, tl);
    double fs = p->force_scalar;
    double drag = td[0] * fs, lift = tl[0] * fs;
    /* trqd = v x bv ; liftd = (-trqd) x v */
    double trqd[3], ntr[3], ld[3];
    cross3(v, bv, trqd);
    ntr[0] = -trqd[0]; ntr[1] = -trqd[1]; ntr[2] = -trqd[2];
    cross3(ntr, v, ld);
    double ln = sqrt(ld[0] * ld[0] + ld[1] * ld[1] + ld[2] * ld[2]);
    int has_lift = (ln > 0);
    for (int i = 0; i < 3; i++) {
        F[i] = drag * v[i] / vn;
        if (has_lift) F[i] += lift * ld[i] / ln;
    }
    if (!dF) return;
    /* ---- derivatives ---- */
    /* d bv / d q (3x4): bv = [1-2(q2^2+q3^2), 2(q1q2+q0q3), 2(q1q3-q0q2)] */
    double dbv[12] = {0, 0, -4 * q[2], -4 * q[3],
                      2 * q[3], 2 * q[2], 2 * q[1], 2 * q[0],
                      -2 * q[2], 2 * q[3], -2 * q[0], 2 * q[1]};
    /* d arg / dq, dv ; d mach / dv */
    double darg[7] = {0}, dmach[7] = {0};
    for (int j = 0; j < 3; j++) dmach[4 + j] = v[j] / (vn * p->sos);
    if (!clamped) {
        for (int j = 0; j < 4; j++) darg[j] = (dbv[0 * 4 + j] * v[0] + dbv[1 * 4 + j] * v[1] + dbv[2 * 4 + j] * v[2]) / vn;
        for (int j = 0; j < 3; j++) darg[4 + j] = bv[j] / vn - c * v[j] / (vn2 * vn);
    }
    double ddrag[7], dlift[7];
    for (int j = 0; j < 7; j++) {
        ddrag[j] = fs * (td[1] * darg[j] + td[2] * dmach[j]);
        dlift[j] = fs * (tl[1] * darg[j] + tl[2] * dmach[j]);
    }
    /* drag term: drag * v/vn */
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 7; j++) dF[i * 7 + j] += ddrag[j] * v[i] / vn;
        for (int j = 0; j < 3; j++) dF[i * 7 + 4 + j] += drag * ((i == j ? 1.0 : 0.0) / vn - v[i] * v[j] / (vn2 * vn));
    }
    if (has_lift) {
        /* ld = (bv x v) x v = v (bv.v) - bv (v.v)  => ld = c v - vn2 bv */
        /* d ld / dq = v * dc/dq - vn2 * dbv/dq ; d ld/dv = c I + v bv^T - 2 bv v^T */
        double dld[21];
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 4; j++) {
                double dc = dbv[0 * 4 + j] * v[0] + dbv[1 * 4 + j] * v[1] + dbv[2 * 4 + j] * v[2];
                dld[i * 7 + j] = v[i] * dc - vn2 * dbv[i * 4 + j];
            }
            for (int j = 0; j < 3; j++)
                dld[i * 7 + 4 + j] = (i == j ? c : 0.0) + v[i] * bv[j] - 2 * bv[i] * v[j];
        }
        /* d (ld/ln) = (I - l l^T)/ln * dld with l = ld/ln */
        double l[3] = {ld[0] / ln, ld[1] / ln, ld[2] / ln};
        for (int j = 0; j < 7; j++) {
            double proj = l[0] * dld[0 * 7 + j] + l[1] * dld[1 * 7 + j] + l[2] * dld[2 * 7 + j];
            for (int i = 0; i < 3; i++) {
                double dl = (dld[i * 7 + j] - l[i] * proj) / ln;
                dF[i * 7 + j] += dlift[j] * l[i] + lift * dl;
            }
        }
    }
}

/* ---------- fin force (build-defined extension; dynamics.jl:60-63 as commented there) ---------- */
/* fd1 = normalize((C e2) x v), fd2 = fd1 x v.  If d != NULL: d fd1 / d(q, v) and d fd2 / d(q, v), each 3x7 row-major. */
static int fin_dirs(const double q[4], const double v[3], const double C[9], double fd1[3], double fd2[3], double *d1, double *d2) {
    double b2[3] = {C[1], C[4], C[7]}; /* C(q) e2 */
    double n[3];
    cross3(b2, v, n);
    double nn = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    if (!(nn > 0)) {
        for (int i = 0; i < 3; i++) fd1[i] = fd2[i] = 0;
        if (d1) { memset(d1, 0, sizeof(double) * 21); memset(d2, 0, sizeof(double) * 21); }
        return 0;
    }
    for (int i = 0; i < 3; i++) fd1[i] = n[i] / nn;
    cross3(fd1, v, fd2);
    if (!d1) return 1;
    /* d b2 / dq (3x4): b2 = [2(q1q2 - q0q3), 1 - 2(q1^2 + q3^2), 2(q2q3 + q0q1)] */
    double db2[12] = {-2 * q[3], 2 * q[2], 2 * q[1], -2 * q[0],
                      0, -4 * q[1], 0, -4 * q[3],
                      2 * q[1], 2 * q[0], 2 * q[3], 2 * q[2]};
    double dn[21]; /* d n / d(q, v) */
    for (int j = 0; j < 4; j++) {
        double col[3] = {db2[j], db2[4 + j], db2[8 + j]}, c[3];
        cross3(col, v, c);
        for (int i = 0; i < 3; i++) dn[i * 7 + j] = c[i];
    }
    for (int j = 0; j < 3; j++) { /* d (b2 x v) / d v_j = b2 x e_j */
        double e[3] = {0, 0, 0}, c[3];
        e[j] = 1;
        cross3(b2, e, c);
        for (int i = 0; i < 3; i++) dn[i * 7 + 4 + j] = c[i];
    }
    for (int j = 0; j < 7; j++) {
        double proj = fd1[0] * dn[j] + fd1[1] * dn[7 + j] + fd1[2] * dn[14 + j];
        for (int i = 0; i < 3; i++) d1[i * 7 + j] = (dn[i * 7 + j] - fd1[i] * proj) / nn;
    }
    for (int j = 0; j < 7; j++) {
        double col[3] = {d1[j], d1[7 + j], d1[14 + j]}, c[3];
        cross3(col, v, c);
        for (int i = 0; i < 3; i++) d2[i * 7 + j] = c[i];
    }
    for (int j = 0; j < 3; j++) { /* + fd1 x e_j */
        double e[3] = {0, 0, 0}, c[3];
        e[j] = 1;
        cross3(fd1, e, c);
        for (int i = 0; i < 3; i++) d2[i * 7 + 4 + j] += c[i];
    }
    return 1;
}

/* ---------- RHS g(x,u) (un-scaled by sigma) and its Jacobians ---------- */
void scvx_oracle_rhs(const oracle_params *p, const double x[NX], const double *u /* [nu] */, double g[NX]) {
    const double *q = x + 7, *w = x + 11, *v = x + 4;
    double C[9], F[3], ff[3] = {0, 0, 0};
    dcm(q, C);
    aero_force(p, q, v, F, 0);
    if (nu_of(p) == 5) { /* ff = u[4] fd1 + u[5] fd2 (dynamics.jl:63); aero_frc = aerf + ff (:66) */
        double fd1[3], fd2[3];
        fin_dirs(q, v, C, fd1, fd2, 0, 0);
        for (int i = 0; i < 3; i++) { ff[i] = u[3] * fd1[i] + u[4] * fd2[i]; F[i] += ff[i]; }
    }
    double un = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    g[0] = -p->alpha * un;
    g[1] = v[0]; g[2] = v[1]; g[3] = v[2];
    for (int i = 0; i < 3; i++) {
        double cu = C[i * 3] * u[0] + C[i * 3 + 1] * u[1] + C[i * 3 + 2] * u[2];
        g[4 + i] = (cu + F[i]) / x[0];
    }
    g[4] -= p->g0;
    /* 0.5 * Omega(w) q, dynamics.jl:46-52 */
    g[7] = 0.5 * (-w[0] * q[1] - w[1] * q[2] - w[2] * q[3]);
    g[8] = 0.5 * (w[0] * q[0] + w[2] * q[2] - w[1] * q[3]);
    g[9] = 0.5 * (w[1] * q[0] - w[2] * q[1] + w[0] * q[3]);
    g[10] = 0.5 * (w[2] * q[0] + w[1] * q[1] - w[0] * q[2]);
    double Jw[3], wxJw[3], rxu[3], t[3], a[3];
    mat3v(p->J, w, Jw);
    cross3(w, Jw, wxJw);
    cross3(p->rTB, u, rxu);
    double rxf[3];
    cross3(p->rFB, ff, rxf); /* aero_trq = cross(rFB, ff) (dynamics.jl:69 as commented there; zero when nu == 3) */
    for (int i = 0; i < 3; i++) t[i] = rxu[i] + rxf[i] - wxJw[i];
    mat3v(p->Jinv, t, a);
    g[11] = a[0]; g[12] = a[1]; g[13] = a[2];
}

/* A = dg/dx (14x14 row-major), Bu = dg/du (14 x nu row-major) */
void scvx_oracle_jac(const oracle_params *p, const double x[NX], const double *u, double A[NX * NX], double *Bu) {
    const int NU = nu_of(p);
    memset(A, 0, sizeof(double) * NX * NX);
    memset(Bu, 0, sizeof(double) * NX * NU);
    const double *q = x + 7, *w = x + 11, *v = x + 4;
    double m = x[0];
    double C[9], D[12], F[3], dF[21];
    dcm(q, C);
    dcm_u_dq(q, u, D);
    aero_force(p, q, v, F, dF);
    double fd1[3] = {0, 0, 0}, fd2[3] = {0, 0, 0}, dff[21];
    memset(dff, 0, sizeof dff);
    if (NU == 5) {
        double d1[21], d2[21];
        fin_dirs(q, v, C, fd1, fd2, d1, d2);
        for (int i = 0; i < 3; i++) F[i] += u[3] * fd1[i] + u[4] * fd2[i];
        for (int i = 0; i < 21; i++) { dff[i] = u[3] * d1[i] + u[4] * d2[i]; dF[i] += dff[i]; }
    }
    double un = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    for (int j = 0; j < 3; j++) Bu[0 * NU + j] = (un > 0) ? -p->alpha * u[j] / un : 0.0;
    for (int i = 0; i < 3; i++) A[(1 + i) * NX + 4 + i] = 1.0;
    for (int i = 0; i < 3; i++) {
        double cu = C[i * 3] * u[0] + C[i * 3 + 1] * u[1] + C[i * 3 + 2] * u[2];
        A[(4 + i) * NX + 0] = -(cu + F[i]) / (m * m);
        for (int j = 0; j < 4; j++) A[(4 + i) * NX + 7 + j] = (D[i * 4 + j] + dF[i * 7 + j]) / m;
        for (int j = 0; j < 3; j++) A[(4 + i) * NX + 4 + j] = dF[i * 7 + 4 + j] / m;
        for (int j = 0; j < 3; j++) Bu[(4 + i) * NU + j] = C[i * 3 + j] / m;
        if (NU == 5) { Bu[(4 + i) * NU + 3] = fd1[i] / m; Bu[(4 + i) * NU + 4] = fd2[i] / m; }
    }
    /* d qdot / dq = 0.5 Omega(w) */
    double Om[16] = {0, -w[0], -w[1], -w[2], w[0], 0, w[2], -w[1], w[1], -w[2], 0, w[0], w[2], w[1], -w[0], 0};
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) A[(7 + i) * NX + 7 + j] = 0.5 * Om[i * 4 + j];
    /* d qdot / dw */
    double Xi[12] = {-q[1], -q[2], -q[3], q[0], -q[3], q[2], q[3], q[0], -q[1], -q[2], q[1], q[0]};
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 3; j++) A[(7 + i) * NX + 11 + j] = 0.5 * Xi[i * 3 + j];
    /* d wdot / dw = -Jinv ( [w]x J - [Jw]x ) */
    double Jw[3];
    mat3v(p->J, w, Jw);
    double M[9]; /* row-major: [w]x J - [Jw]x */
    double wx[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    double Jx[9] = {0, -Jw[2], Jw[1], Jw[2], 0, -Jw[0], -Jw[1], Jw[0], 0};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += wx[i * 3 + k] * p->J[j * 3 + k]; /* J col-major: J(k,j) */
            M[i * 3 + j] = s - Jx[i * 3 + j];
        }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += p->Jinv[k * 3 + i] * M[k * 3 + j];
            A[(11 + i) * NX + 11 + j] = -s;
        }
    /* d wdot / du = Jinv [rTB]x */
    const double *r = p->rTB;
    double rx[9] = {0, -r[2], r[1], r[2], 0, -r[0], -r[1], r[0], 0};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += p->Jinv[k * 3 + i] * rx[k * 3 + j];
            Bu[(11 + i) * NU + j] = s;
        }
    if (NU == 5) { /* wdot += Jinv (rFB x ff): columns q, v of A and u4, u5 of Bu */
        const double *f = p->rFB;
        double fx[9] = {0, -f[2], f[1], f[2], 0, -f[0], -f[1], f[0], 0};
        double G[9]; /* Jinv [rFB]x, row-major */
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                double s = 0;
                for (int k = 0; k < 3; k++) s += p->Jinv[k * 3 + i] * fx[k * 3 + j];
                G[i * 3 + j] = s;
            }
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 7; j++) {
                double s = G[i * 3] * dff[j] + G[i * 3 + 1] * dff[7 + j] + G[i * 3 + 2] * dff[14 + j];
                A[(11 + i) * NX + (j < 4 ? 7 + j : j)] += s; /* dff columns: q0..q3 -> 7..10, v1..v3 -> 4..6 */
            }
            Bu[(11 + i) * NU + 3] = G[i * 3] * fd1[0] + G[i * 3 + 1] * fd1[1] + G[i * 3 + 2] * fd1[2];
            Bu[(11 + i) * NU + 4] = G[i * 3] * fd2[0] + G[i * 3 + 1] * fd2[1] + G[i * 3 + 2] * fd2[2];
        }
    }
}

/* time derivative of the augmented (x | S) system at normalised time fraction lkp = t/dt.
 * S is 14x21 row-major here (converted to Julia column-major on output). */
static void aug_rhs(const oracle_params *p, const double *x, const double *S, const double *uk, const double *up,
                    double sigma, double lkp, double *dx, double *dS) {
    const int NU = nu_of(p), NP = 14 + 2 * NU + 1;
    double lkm = 1.0 - lkp, u[NU_MAX];
    for (int i = 0; i < NU; i++) u[i] = uk[i] * lkm + up[i] * lkp; /* dynamics.jl:108-110,144-150 */
    double g[NX];
    scvx_oracle_rhs(p, x, u, g);
    for (int i = 0; i < NX; i++) dx[i] = sigma * g[i];
    if (!S) return;
    double A[NX * NX], Bu[NX * NU_MAX];
    scvx_oracle_jac(p, x, u, A, Bu);
    for (int i = 0; i < NX; i++)
        for (int j = 0; j < NP; j++) {
            double s = 0;
            for (int k = 0; k < NX; k++) s += A[i * NX + k] * S[k * NP + j];
            if (j >= 14 && j < 14 + NU) s += Bu[i * NU + (j - 14)] * lkm;
            else if (j >= 14 + NU && j < 14 + 2 * NU) s += Bu[i * NU + (j - 14 - NU)] * lkp;
            s *= sigma;
            if (j == NP - 1) s += g[i];
            dS[i * NP + j] = s;
        }
}

/* one segment: inp[np] -> endpoint[14], deriv (column-major 14 x np, may be NULL); np = 14 + 2 nu + 1 */
void scvx_oracle_segment(const oracle_params *p, const double *inp, double dt, int nsub, double *endpoint, double *deriv) {
    const int NU = nu_of(p), NP = 14 + 2 * NU + 1;
    double x[NX], S[NX * NP_MAX];
    const double *uk = inp + 14, *up = inp + 14 + NU;
    double sigma = inp[NP - 1];
    memcpy(x, inp, sizeof(double) * NX);
    int withS = deriv != 0;
    if (withS) {
        memset(S, 0, sizeof(double) * NX * NP);
        for (int i = 0; i < NX; i++) S[i * NP + i] = 1.0;
    }
    double h = dt / nsub;
    double k1x[NX], k2x[NX], k3x[NX], k4x[NX], xt[NX];
    double k1S[NX * NP_MAX], k2S[NX * NP_MAX], k3S[NX * NP_MAX], k4S[NX * NP_MAX], St[NX * NP_MAX];
    for (int s = 0; s < nsub; s++) {
        double f0 = (double)s / nsub, fm = (s + 0.5) / nsub, f1 = (double)(s + 1) / nsub;
        aug_rhs(p, x, withS ? S : 0, uk, up, sigma, f0, k1x, k1S);
        for (int i = 0; i < NX; i++) xt[i] = x[i] + 0.5 * h * k1x[i];
        if (withS) for (int i = 0; i < NX * NP; i++) St[i] = S[i] + 0.5 * h * k1S[i];
        aug_rhs(p, xt, withS ? St : 0, uk, up, sigma, fm, k2x, k2S);
        for (int i = 0; i < NX; i++) xt[i] = x[i] + 0.5 * h * k2x[i];
        if (withS) for (int i = 0; i < NX * NP; i++) St[i] = S[i] + 0.5 * h * k2S[i];
        aug_rhs(p, xt, withS ? St : 0, uk, up, sigma, fm, k3x, k3S);
        for (int i = 0; i < NX; i++) xt[i] = x[i] + h * k3x[i];
        if (withS) for (int i = 0; i < NX * NP; i++) St[i] = S[i] + h * k3S[i];
        aug_rhs(p, xt, withS ? St : 0, uk, up, sigma, f1, k4x, k4S);
        for (int i = 0; i < NX; i++) x[i] += h * (k1x[i] + 2 * k2x[i] + 2 * k3x[i] + k4x[i]) / 6.0;
        if (withS) for (int i = 0; i < NX * NP; i++) S[i] += h * (k1S[i] + 2 * k2S[i] + 2 * k3S[i] + k4S[i]) / 6.0;
    }
    memcpy(endpoint, x, sizeof(double) * NX);
    if (withS)
        for (int i = 0; i < NX; i++)
            for (int j = 0; j < NP; j++) deriv[j * NX + i] = S[i * NP + j];
}

/* Dynamics.linearize_dynamics (dynamics.jl:321-334) over a batch, ABI layouts of include/scvx.h. */
void scvx_oracle_linearize(const oracle_params *p, int B, int K, const double *x, const double *u, const double *sigma,
                           double dt, int nsub, double *endpoint, double *deriv) {
    const int NU = nu_of(p), NP = 14 + 2 * NU + 1;
#pragma omp parallel for schedule(static)
    for (long s = 0; s < (long)B * K; s++) {
        int b = (int)(s / K), k = (int)(s % K);
        double inp[NP_MAX];
        const double *xk = x + ((size_t)b * (K + 1) + k) * NX;
        const double *uk = u + ((size_t)b * (K + 1) + k) * NU;
        memcpy(inp, xk, sizeof(double) * NX);
        memcpy(inp + 14, uk, sizeof(double) * 2 * NU); /* u_k then u_{k+1}: adjacent */
        inp[NP - 1] = sigma[b];
        scvx_oracle_segment(p, inp, dt, nsub, endpoint + (size_t)s * NX, deriv ? deriv + (size_t)s * NX * NP : 0);
    }
}

/* Dynamics.predict_state (dynamics.jl:315-317) over a batch. */
void scvx_oracle_propagate(const oracle_params *p, int B, int K, const double *x, const double *u, const double *sigma,
                           double dt, int nsub, double *xnext) {
    scvx_oracle_linearize(p, B, K, x, u, sigma, dt, nsub, xnext, 0);
}

/* exposed for the spline tests */
void scvx_oracle_table_eval(const oracle_params *p, int which, double aoa, double mach, double out[3]) {
    table_eval(p, which == 0 ? p->cdrag : p->clift, aoa, mach, out);
}
void scvx_oracle_aero_force(const oracle_params *p, const double q[4], const double v[3], double F[3], double dF[21]) {
    aero_force(p, q, v, F, dF);
}
void scvx_oracle_fin_dirs(const double q[4], const double v[3], double fd1[3], double fd2[3], double d1[21], double d2[21]) {
    double C[9];
    dcm(q, C);
    fin_dirs(q, v, C, fd1, fd2, d1, d2);
}
