// Device-side 6-DoF rigid-body dynamics for the SCvx discretisation kernels (gfx950).
//
// What the reference computes here (file:line into BenChung/SuccessiveConvexification):
//   RHS                       Dynamics.dx_static        dynamics.jl:54-77 (DCM :29-44, Omega :46-52)
//   first-order-hold control  current_control           dynamics.jl:108-110, 144-150
//   aerodynamic force         Aerodynamics.aero_force   aerodynamics.jl:60-77 + shims dynamics.jl:162-207
//   table interpolation       load_aerodata             aerodynamics.jl:17-21 (cubic B-spline, Flat)
//   fin force (FIN = true)    the commented expressions of dynamics.jl:60-63, 66, 69: fd1 = normalize((C e2) x v),
//                             fd2 = fd1 x v, ff = u[4] fd1 + u[5] fd2 added to the aerodynamic force, torque rFB x ff
//                             (build-defined model; control_dim = 5)
// The reference obtains Jacobians by forward-mode AD over generated code (dynamics.jl:245-256); here
// they are written out analytically and applied column-wise: a lane never forms the 14x14 matrix, it
// applies the ~48 structural non-zeros of df/dx directly to the sensitivity column it owns.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

namespace scvx {

struct DynParams {
    double alpha, g0, sos;
    double J[9];     // row-major inertia
    double Jinv[9];  // row-major inverse inertia
    double rTB[3];
    double JrT[9];   // row-major Jinv * [rTB]x  (d wdot / du, constant)
    int aero;        // 0 exo, 1 atmospheric
    int n_aoa, n_mach;
    int fin;         // 1: fin extension (control_dim = 5)
    double JrF[9];   // row-major Jinv * [rFB]x  (d wdot / d ff, constant)
    double aoa0, inv_daoa, mach0, inv_dmach, force_scalar;
    const double* cdrag;  // prefiltered coefficients, (n_mach+2) x (n_aoa+2), aoa fastest
    const double* clift;
};

// The device functions below are templates on the arithmetic type R: double is the reference precision (and what the
// SCvx loop uses); float backs the scvx_*_f32 entry points (BASELINE configs[3-4] name fp32).  DynP<R> is DynParams with
// every constant converted once ON THE HOST and passed as the kernel argument (scalar registers), so that no expression
// silently promotes to double in the float build and no vector register holds a constant.
template <typename R>
struct DynP {
    R alpha, g0, sos;
    R J[9], Jinv[9], rTB[3], JrT[9], JrF[9];
    int aero, n_aoa, n_mach;
    R aoa0, inv_daoa, mach0, inv_dmach, force_scalar;
    const double* cdrag;   // prefiltered coefficients stay double in memory (shared by both precisions, cache-resident)
    const double* clift;
    __host__ __device__ __forceinline__ explicit DynP(const DynParams& p)
        : alpha((R)p.alpha), g0((R)p.g0), sos((R)p.sos), aero(p.aero), n_aoa(p.n_aoa), n_mach(p.n_mach), aoa0((R)p.aoa0),
          inv_daoa((R)p.inv_daoa), mach0((R)p.mach0), inv_dmach((R)p.inv_dmach), force_scalar((R)p.force_scalar),
          cdrag(p.cdrag), clift(p.clift) {
#pragma unroll
        for (int i = 0; i < 9; i++) { J[i] = (R)p.J[i]; Jinv[i] = (R)p.Jinv[i]; JrT[i] = (R)p.JrT[i]; JrF[i] = (R)p.JrF[i]; }
#pragma unroll
        for (int i = 0; i < 3; i++) rTB[i] = (R)p.rTB[i];
    }
};

// Everything one RK stage needs about the state trajectory, evaluated once per stage per lane.
template <bool AERO, typename R = double, bool FIN = false>
struct Stage {
    R g[14];      // un-scaled RHS
    R C[9];       // DCM, row-major
    R invm;
    R am[3];      // d vdot / d m
    R Dq[12];     // d vdot / d q   (3x4 row-major)
    R Dv[(AERO || FIN) ? 9 : 1];  // d vdot / d v (aero / fins)
    R Mw[9];      // d wdot / d w
    // fin extension: force directions, d wdot / d q and d v through the fin torque, d wdot / d (u4, u5)
    R fd1[FIN ? 3 : 1], fd2[FIN ? 3 : 1], Wq[FIN ? 12 : 1], Wv[FIN ? 9 : 1], Gf1[FIN ? 3 : 1], Gf2[FIN ? 3 : 1];
};

// fd1 = normalize((C e2) x v), fd2 = fd1 x v (dynamics.jl:60-62); JAC: d fd1, d fd2 / d (q0..q3, v1..v3), 3x7 row-major each.
// (C e2) x v = 0 gives no fin force (the reference's ifnz convention for its other normalised directions).
template <bool JAC, typename R>
__device__ __forceinline__ void fin_dirs(const R* q, const R* v, const R* C, R fd1[3], R fd2[3], R d1[21], R d2[21]) {
    const R b2[3] = {C[1], C[4], C[7]};
    const R n[3] = {b2[1] * v[2] - b2[2] * v[1], b2[2] * v[0] - b2[0] * v[2], b2[0] * v[1] - b2[1] * v[0]};
    const R nn = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    const bool ok = nn > R(0.0);
    const R inn = ok ? R(1.0) / nn : R(0.0);
#pragma unroll
    for (int i = 0; i < 3; i++) fd1[i] = n[i] * inn;
    fd2[0] = fd1[1] * v[2] - fd1[2] * v[1];
    fd2[1] = fd1[2] * v[0] - fd1[0] * v[2];
    fd2[2] = fd1[0] * v[1] - fd1[1] * v[0];
    if (!JAC) return;
    // d b2 / d q (3x4): b2 = [2(q1q2 - q0q3), 1 - 2(q1^2 + q3^2), 2(q2q3 + q0q1)]
    const R db2[12] = {-R(2.0) * q[3], R(2.0) * q[2], R(2.0) * q[1], -R(2.0) * q[0],
                       R(0.0), -R(4.0) * q[1], R(0.0), -R(4.0) * q[3],
                       R(2.0) * q[1], R(2.0) * q[0], R(2.0) * q[3], R(2.0) * q[2]};
    R dn[21];
#pragma unroll
    for (int j = 0; j < 4; j++) {   // (d b2 / d q_j) x v
        const R c0 = db2[j], c1 = db2[4 + j], c2 = db2[8 + j];
        dn[j] = c1 * v[2] - c2 * v[1];
        dn[7 + j] = c2 * v[0] - c0 * v[2];
        dn[14 + j] = c0 * v[1] - c1 * v[0];
    }
    // d (b2 x v) / d v = [b2]x
    dn[4] = R(0.0);  dn[5] = -b2[2]; dn[6] = b2[1];
    dn[11] = b2[2];  dn[12] = R(0.0); dn[13] = -b2[0];
    dn[18] = -b2[1]; dn[19] = b2[0];  dn[20] = R(0.0);
#pragma unroll
    for (int j = 0; j < 7; j++) {
        const R proj = fd1[0] * dn[j] + fd1[1] * dn[7 + j] + fd1[2] * dn[14 + j];
#pragma unroll
        for (int i = 0; i < 3; i++) d1[i * 7 + j] = ok ? (dn[i * 7 + j] - fd1[i] * proj) * inn : R(0.0);
    }
#pragma unroll
    for (int j = 0; j < 7; j++) {   // (d fd1) x v
        const R c0 = d1[j], c1 = d1[7 + j], c2 = d1[14 + j];
        d2[j] = c1 * v[2] - c2 * v[1];
        d2[7 + j] = c2 * v[0] - c0 * v[2];
        d2[14 + j] = c0 * v[1] - c1 * v[0];
    }
    // + fd1 x e_j = [fd1]x columns
    d2[5] -= fd1[2];  d2[6] += fd1[1];
    d2[11] += fd1[2]; d2[13] -= fd1[0];
    d2[18] -= fd1[1]; d2[19] += fd1[0];
}

template <typename R>
__device__ __forceinline__ void bspline_weights(R t, int n, int& i0, R w[4], R dw[4]) {
    int i = (int)floor(t);
    i = i < 0 ? 0 : (i > n - 2 ? n - 2 : i);
    R d = t - R(i);
    R d2 = d * d, d3 = d2 * d;
    const R s = R(1.0) / R(6.0);
    w[0] = (R(1.0) - R(3.0) * d + R(3.0) * d2 - d3) * s;
    w[1] = (R(4.0) - R(6.0) * d2 + R(3.0) * d3) * s;
    w[2] = (R(1.0) + R(3.0) * d + R(3.0) * d2 - R(3.0) * d3) * s;
    w[3] = d3 * s;
    dw[0] = (-R(3.0) + R(6.0) * d - R(3.0) * d2) * s;
    dw[1] = (-R(12.0) * d + R(9.0) * d2) * s;
    dw[2] = (R(3.0) + R(6.0) * d - R(9.0) * d2) * s;
    dw[3] = (R(3.0) * d2) * s;
    i0 = i;
}

// drag and lift tables at (aoa, mach): value, d/daoa, d/dmach each; Flat() extrapolation.
template <typename R>
__device__ __forceinline__ void aero_tables(const DynP<R>& p, R aoa, R mach, R td[3], R tl[3]) {
    const int na = p.n_aoa, nm = p.n_mach;
    R ta = (aoa - p.aoa0) * p.inv_daoa, tm = (mach - p.mach0) * p.inv_dmach;
    bool fa = false, fm = false;
    if (ta < R(0.0)) { ta = R(0.0); fa = true; }
    if (ta > R(na - 1)) { ta = R(na - 1); fa = true; }
    if (tm < R(0.0)) { tm = R(0.0); fm = true; }
    if (tm > (R)(nm - 1)) { tm = (R)(nm - 1); fm = true; }
    int ia, im;
    R wa[4], dwa[4], wm[4], dwm[4];
    bspline_weights(ta, na, ia, wa, dwa);
    bspline_weights(tm, nm, im, wm, dwm);
    const int lda = na + 2;
    R vd = 0, vda = 0, vdm = 0, vl = 0, vla = 0, vlm = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const double* rd = p.cdrag + (size_t)(im + b) * lda + ia;
        const double* rl = p.clift + (size_t)(im + b) * lda + ia;
        R sd = 0, sda = 0, sl = 0, sla = 0;
#pragma unroll
        for (int a = 0; a < 4; a++) {
            const R cd = (R)rd[a], cl = (R)rl[a];
            sd = fma(wa[a], cd, sd);
            sda = fma(dwa[a], cd, sda);
            sl = fma(wa[a], cl, sl);
            sla = fma(dwa[a], cl, sla);
        }
        vd = fma(wm[b], sd, vd);
        vda = fma(wm[b], sda, vda);
        vdm = fma(dwm[b], sd, vdm);
        vl = fma(wm[b], sl, vl);
        vla = fma(wm[b], sla, vla);
        vlm = fma(dwm[b], sl, vlm);
    }
    td[0] = vd;
    td[1] = fa ? R(0.0) : vda * p.inv_daoa;
    td[2] = fm ? R(0.0) : vdm * p.inv_dmach;
    tl[0] = vl;
    tl[1] = fa ? R(0.0) : vla * p.inv_daoa;
    tl[2] = fm ? R(0.0) : vlm * p.inv_dmach;
}

// F[3] and (JAC) dF/d(q0..q3, v1..v3) as 3x7 row-major.
template <bool JAC, typename R>
__device__ __forceinline__ void aero_force(const DynP<R>& p, const R* q, const R* v, const R* C,
                                           R F[3], R dF[21]) {
    F[0] = F[1] = F[2] = R(0.0);
    if (JAC) {
#pragma unroll
        for (int i = 0; i < 21; i++) dF[i] = R(0.0);
    }
    const R bv[3] = {C[0], C[3], C[6]};
    const R vn2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    const R vn = sqrt(vn2);
    if (!(vn > R(0.0))) return;
    const R ivn = R(1.0) / vn;
    const R c = bv[0] * v[0] + bv[1] * v[1] + bv[2] * v[2];
    const R mach = vn / p.sos;
    R arg = c / (mach * p.sos);
    bool clamped = false;
    if (arg < -R(1.0)) { arg = -R(1.0); clamped = true; }
    if (arg > R(1.0)) { arg = R(1.0); clamped = true; }
    R td[3], tl[3];
    aero_tables(p, arg, mach, td, tl);
    const R fs = p.force_scalar;
    const R drag = td[0] * fs, lift = tl[0] * fs;
    // liftd = (bv x v) x v = c v - |v|^2 bv
    R ld[3];
#pragma unroll
    for (int i = 0; i < 3; i++) ld[i] = c * v[i] - vn2 * bv[i];
    const R ln = sqrt(ld[0] * ld[0] + ld[1] * ld[1] + ld[2] * ld[2]);
    const bool has_lift = ln > R(0.0);
    const R iln = has_lift ? R(1.0) / ln : R(0.0);
    R l[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        l[i] = ld[i] * iln;
        F[i] = drag * v[i] * ivn + (has_lift ? lift * l[i] : R(0.0));
    }
    if (!JAC) return;
    const R dbv[12] = {R(0.0), R(0.0), -R(4.0) * q[2], -R(4.0) * q[3],
                            R(2.0) * q[3], R(2.0) * q[2], R(2.0) * q[1], R(2.0) * q[0],
                            -R(2.0) * q[2], R(2.0) * q[3], -R(2.0) * q[0], R(2.0) * q[1]};
    R dc[4];  // d c / d q
#pragma unroll
    for (int j = 0; j < 4; j++) dc[j] = dbv[j] * v[0] + dbv[4 + j] * v[1] + dbv[8 + j] * v[2];
    R darg[7], dmach[7];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        darg[j] = clamped ? R(0.0) : dc[j] * ivn;
        dmach[j] = R(0.0);
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
        darg[4 + j] = clamped ? R(0.0) : (bv[j] * ivn - c * v[j] * ivn * ivn * ivn);
        dmach[4 + j] = v[j] * ivn / p.sos;
    }
    R ddrag[7], dlift[7];
#pragma unroll
    for (int j = 0; j < 7; j++) {
        ddrag[j] = fs * (td[1] * darg[j] + td[2] * dmach[j]);
        dlift[j] = fs * (tl[1] * darg[j] + tl[2] * dmach[j]);
    }
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int j = 0; j < 7; j++) dF[i * 7 + j] = ddrag[j] * v[i] * ivn;
#pragma unroll
        for (int j = 0; j < 3; j++)
            dF[i * 7 + 4 + j] += drag * ((i == j ? ivn : R(0.0)) - v[i] * v[j] * ivn * ivn * ivn);
    }
    if (has_lift) {
        R dld[21];
#pragma unroll
        for (int i = 0; i < 3; i++) {
#pragma unroll
            for (int j = 0; j < 4; j++) dld[i * 7 + j] = v[i] * dc[j] - vn2 * dbv[i * 4 + j];
#pragma unroll
            for (int j = 0; j < 3; j++) dld[i * 7 + 4 + j] = (i == j ? c : R(0.0)) + v[i] * bv[j] - R(2.0) * bv[i] * v[j];
        }
#pragma unroll
        for (int j = 0; j < 7; j++) {
            const R proj = l[0] * dld[j] + l[1] * dld[7 + j] + l[2] * dld[14 + j];
#pragma unroll
            for (int i = 0; i < 3; i++) dF[i * 7 + j] += dlift[j] * l[i] + lift * (dld[i * 7 + j] - l[i] * proj) * iln;
        }
    }
}

template <typename R>
__device__ __forceinline__ void dcm(const R* q, R* C) {
    const R q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    C[0] = R(1.0) - R(2.0) * (q2 * q2 + q3 * q3);
    C[1] = R(2.0) * (q1 * q2 - q0 * q3);
    C[2] = R(2.0) * (q1 * q3 + q0 * q2);
    C[3] = R(2.0) * (q1 * q2 + q0 * q3);
    C[4] = R(1.0) - R(2.0) * (q1 * q1 + q3 * q3);
    C[5] = R(2.0) * (q2 * q3 - q0 * q1);
    C[6] = R(2.0) * (q1 * q3 - q0 * q2);
    C[7] = R(2.0) * (q2 * q3 + q0 * q1);
    C[8] = R(1.0) - R(2.0) * (q1 * q1 + q2 * q2);
}

// RHS only (K2 propagate and the state part of K1).
template <bool AERO, bool FIN = false, typename R>
__device__ __forceinline__ void rhs_only(const DynP<R>& p, const R* x, const R* u, R* g) {
    const R* v = x + 4;
    const R* q = x + 7;
    const R* w = x + 11;
    R C[9], F[3] = {R(0.0), R(0.0), R(0.0)}, ff[3] = {R(0.0), R(0.0), R(0.0)};
    dcm(q, C);
    if (AERO) aero_force<false, R>(p, q, v, C, F, nullptr);
    if (FIN) {
        R fd1[3], fd2[3];
        fin_dirs<false, R>(q, v, C, fd1, fd2, nullptr, nullptr);
#pragma unroll
        for (int i = 0; i < 3; i++) { ff[i] = u[3] * fd1[i] + u[4] * fd2[i]; F[i] += ff[i]; }
    }
    const R invm = R(1.0) / x[0];
    g[0] = -p.alpha * sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    g[1] = v[0]; g[2] = v[1]; g[3] = v[2];
#pragma unroll
    for (int i = 0; i < 3; i++) g[4 + i] = (C[3 * i] * u[0] + C[3 * i + 1] * u[1] + C[3 * i + 2] * u[2] + F[i]) * invm;
    g[4] -= p.g0;
    g[7] = R(0.5) * (-w[0] * q[1] - w[1] * q[2] - w[2] * q[3]);
    g[8] = R(0.5) * (w[0] * q[0] + w[2] * q[2] - w[1] * q[3]);
    g[9] = R(0.5) * (w[1] * q[0] - w[2] * q[1] + w[0] * q[3]);
    g[10] = R(0.5) * (w[2] * q[0] + w[1] * q[1] - w[0] * q[2]);
    R Jw[3], t[3];
#pragma unroll
    for (int i = 0; i < 3; i++) Jw[i] = p.J[3 * i] * w[0] + p.J[3 * i + 1] * w[1] + p.J[3 * i + 2] * w[2];
    t[0] = (p.rTB[1] * u[2] - p.rTB[2] * u[1]) - (w[1] * Jw[2] - w[2] * Jw[1]);
    t[1] = (p.rTB[2] * u[0] - p.rTB[0] * u[2]) - (w[2] * Jw[0] - w[0] * Jw[2]);
    t[2] = (p.rTB[0] * u[1] - p.rTB[1] * u[0]) - (w[0] * Jw[1] - w[1] * Jw[0]);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        g[11 + i] = p.Jinv[3 * i] * t[0] + p.Jinv[3 * i + 1] * t[1] + p.Jinv[3 * i + 2] * t[2];
        if (FIN) g[11 + i] += p.JrF[3 * i] * ff[0] + p.JrF[3 * i + 1] * ff[1] + p.JrF[3 * i + 2] * ff[2];   // Jinv (rFB x ff)
    }
}

// RHS + the structural non-zeros of df/dx at (x,u).
template <bool AERO, bool FIN = false, typename R>
__device__ __forceinline__ void stage_eval(const DynP<R>& p, const R* x, const R* u, Stage<AERO, R, FIN>& s) {
    const R* v = x + 4;
    const R* q = x + 7;
    const R* w = x + 11;
    dcm(q, s.C);
    R F[3] = {R(0.0), R(0.0), R(0.0)};
    R dF[(AERO || FIN) ? 21 : 1];
    if (AERO) aero_force<true>(p, q, v, s.C, F, dF);
    R ff[3] = {R(0.0), R(0.0), R(0.0)};
    if (FIN) {
        R d1[21], d2[21];
        fin_dirs<true, R>(q, v, s.C, s.fd1, s.fd2, d1, d2);
        R dff[21];
#pragma unroll
        for (int i = 0; i < 21; i++) {
            dff[i] = u[3] * d1[i] + u[4] * d2[i];
            dF[i] = AERO ? dF[i] + dff[i] : dff[i];
        }
#pragma unroll
        for (int i = 0; i < 3; i++) {
            ff[i] = u[3] * s.fd1[i] + u[4] * s.fd2[i];
            F[i] += ff[i];
            s.Gf1[i] = p.JrF[3 * i] * s.fd1[0] + p.JrF[3 * i + 1] * s.fd1[1] + p.JrF[3 * i + 2] * s.fd1[2];
            s.Gf2[i] = p.JrF[3 * i] * s.fd2[0] + p.JrF[3 * i + 1] * s.fd2[1] + p.JrF[3 * i + 2] * s.fd2[2];
#pragma unroll
            for (int j = 0; j < 4; j++)
                s.Wq[4 * i + j] = p.JrF[3 * i] * dff[j] + p.JrF[3 * i + 1] * dff[7 + j] + p.JrF[3 * i + 2] * dff[14 + j];
#pragma unroll
            for (int j = 0; j < 3; j++)
                s.Wv[3 * i + j] = p.JrF[3 * i] * dff[4 + j] + p.JrF[3 * i + 1] * dff[11 + j] + p.JrF[3 * i + 2] * dff[18 + j];
        }
    }
    const R invm = R(1.0) / x[0];
    s.invm = invm;
    s.g[0] = -p.alpha * sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    s.g[1] = v[0]; s.g[2] = v[1]; s.g[3] = v[2];
    const R q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    const R u1 = u[0], u2 = u[1], u3 = u[2];
    // d(C u)/dq, 3x4 row-major
    R D[12];
    D[0] = R(2.0) * (-q3 * u2 + q2 * u3);
    D[4] = R(2.0) * (q3 * u1 - q1 * u3);
    D[8] = R(2.0) * (-q2 * u1 + q1 * u2);
    D[1] = R(2.0) * (q2 * u2 + q3 * u3);
    D[5] = R(2.0) * (q2 * u1 - R(2.0) * q1 * u2 - q0 * u3);
    D[9] = R(2.0) * (q3 * u1 + q0 * u2 - R(2.0) * q1 * u3);
    D[2] = R(2.0) * (-R(2.0) * q2 * u1 + q1 * u2 + q0 * u3);
    D[6] = R(2.0) * (q1 * u1 + q3 * u3);
    D[10] = R(2.0) * (-q0 * u1 + q3 * u2 - R(2.0) * q2 * u3);
    D[3] = R(2.0) * (-R(2.0) * q3 * u1 - q0 * u2 + q1 * u3);
    D[7] = R(2.0) * (q0 * u1 - R(2.0) * q3 * u2 + q2 * u3);
    D[11] = R(2.0) * (q1 * u1 + q2 * u2);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const R acc = (s.C[3 * i] * u1 + s.C[3 * i + 1] * u2 + s.C[3 * i + 2] * u3 + F[i]) * invm;
        s.g[4 + i] = acc;
        s.am[i] = -acc * invm;
#pragma unroll
        for (int j = 0; j < 4; j++) s.Dq[4 * i + j] = (D[4 * i + j] + ((AERO || FIN) ? dF[7 * i + j] : R(0.0))) * invm;
        if (AERO || FIN) {
#pragma unroll
            for (int j = 0; j < 3; j++) s.Dv[3 * i + j] = dF[7 * i + 4 + j] * invm;
        }
    }
    s.g[4] -= p.g0;
    s.g[7] = R(0.5) * (-w[0] * q1 - w[1] * q2 - w[2] * q3);
    s.g[8] = R(0.5) * (w[0] * q0 + w[2] * q2 - w[1] * q3);
    s.g[9] = R(0.5) * (w[1] * q0 - w[2] * q1 + w[0] * q3);
    s.g[10] = R(0.5) * (w[2] * q0 + w[1] * q1 - w[0] * q2);
    R Jw[3], t[3];
#pragma unroll
    for (int i = 0; i < 3; i++) Jw[i] = p.J[3 * i] * w[0] + p.J[3 * i + 1] * w[1] + p.J[3 * i + 2] * w[2];
    t[0] = (p.rTB[1] * u3 - p.rTB[2] * u2) - (w[1] * Jw[2] - w[2] * Jw[1]);
    t[1] = (p.rTB[2] * u1 - p.rTB[0] * u3) - (w[2] * Jw[0] - w[0] * Jw[2]);
    t[2] = (p.rTB[0] * u2 - p.rTB[1] * u1) - (w[0] * Jw[1] - w[1] * Jw[0]);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        s.g[11 + i] = p.Jinv[3 * i] * t[0] + p.Jinv[3 * i + 1] * t[1] + p.Jinv[3 * i + 2] * t[2];
        if (FIN) s.g[11 + i] += p.JrF[3 * i] * ff[0] + p.JrF[3 * i + 1] * ff[1] + p.JrF[3 * i + 2] * ff[2];
    }
    // T = [w]x J - [Jw]x ; Mw = -Jinv T
    R T[9];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        T[0 + j] = -w[2] * p.J[3 + j] + w[1] * p.J[6 + j];
        T[3 + j] = w[2] * p.J[0 + j] - w[0] * p.J[6 + j];
        T[6 + j] = -w[1] * p.J[0 + j] + w[0] * p.J[3 + j];
    }
    T[1] += Jw[2]; T[2] -= Jw[1];
    T[3] -= Jw[2]; T[5] += Jw[0];
    T[6] += Jw[1]; T[7] -= Jw[0];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            s.Mw[3 * i + j] = -(p.Jinv[3 * i] * T[j] + p.Jinv[3 * i + 1] * T[3 + j] + p.Jinv[3 * i + 2] * T[6 + j]);
}

// d/dt of one sensitivity column c (14 values):  sigma * (A c + Bu * wc) + gsel * g
// wc[3] = FOH weight of this column's control component (zero for state / sigma columns).
template <bool AERO, typename R>
__device__ __forceinline__ void column_deriv(const DynP<R>& p, const Stage<AERO, R>& s, const R* x, const R* u,
                                             const R* c, const R* wc, R gsel, R sigma, R* dc) {
    const R* q = x + 7;
    const R* w = x + 11;
    const R un = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    const R iun = un > R(0.0) ? R(1.0) / un : R(0.0);
    R a[14];
    a[0] = -p.alpha * iun * (u[0] * wc[0] + u[1] * wc[1] + u[2] * wc[2]);
    a[1] = c[4]; a[2] = c[5]; a[3] = c[6];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        R t = s.am[i] * c[0];
        t = fma(s.Dq[4 * i], c[7], t);
        t = fma(s.Dq[4 * i + 1], c[8], t);
        t = fma(s.Dq[4 * i + 2], c[9], t);
        t = fma(s.Dq[4 * i + 3], c[10], t);
        if (AERO) {
            t = fma(s.Dv[3 * i], c[4], t);
            t = fma(s.Dv[3 * i + 1], c[5], t);
            t = fma(s.Dv[3 * i + 2], c[6], t);
        }
        t = fma((s.C[3 * i] * wc[0] + s.C[3 * i + 1] * wc[1] + s.C[3 * i + 2] * wc[2]), s.invm, t);
        a[4 + i] = t;
    }
    const R cq0 = c[7], cq1 = c[8], cq2 = c[9], cq3 = c[10];
    const R cw0 = c[11], cw1 = c[12], cw2 = c[13];
    a[7] = R(0.5) * (-w[0] * cq1 - w[1] * cq2 - w[2] * cq3 - q[1] * cw0 - q[2] * cw1 - q[3] * cw2);
    a[8] = R(0.5) * (w[0] * cq0 + w[2] * cq2 - w[1] * cq3 + q[0] * cw0 - q[3] * cw1 + q[2] * cw2);
    a[9] = R(0.5) * (w[1] * cq0 - w[2] * cq1 + w[0] * cq3 + q[3] * cw0 + q[0] * cw1 - q[1] * cw2);
    a[10] = R(0.5) * (w[2] * cq0 + w[1] * cq1 - w[0] * cq2 - q[2] * cw0 + q[1] * cw1 + q[0] * cw2);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        R t = s.Mw[3 * i] * cw0;
        t = fma(s.Mw[3 * i + 1], cw1, t);
        t = fma(s.Mw[3 * i + 2], cw2, t);
        t = fma(p.JrT[3 * i], wc[0], t);
        t = fma(p.JrT[3 * i + 1], wc[1], t);
        t = fma(p.JrT[3 * i + 2], wc[2], t);
        a[11 + i] = t;
    }
#pragma unroll
    for (int i = 0; i < 14; i++) dc[i] = fma(sigma, a[i], gsel * s.g[i]);
}

// ---- producer / consumer form of the stage: what a column needs to know about the state trajectory ----
// Coefficient record of one RK stage of one segment (NCOEF doubles, stored [field][segment] in LDS):
//   g[14] | C[9] | invm | am[3] | Dq[12] | Mw[9] | q[4] | w[3] | ku[3] = -alpha u/|u| | (aero / fins) Dv[9]
//   | (fins) fd1[3] fd2[3] Wq[12] Wv[9] Gf1[3] Gf2[3]
template <bool AERO, bool FIN = false> struct StageRec { static constexpr int N = FIN ? 100 : (AERO ? 67 : 58); };

// ---- column-wise force derivatives for the producer ----
// The producer needs d F / d (q0..q3, v1..v3) only to fold it, column by column, into the record's Dq / Dv (and Wq / Wv) entries.
// aero_force<true> and fin_dirs<true> return the whole 3x7 arrays (plus 3x7 temporaries): ~90 values live at once.  Here the
// column-independent part is prepared once (AeroPrep / FinPrep) and column j is produced, used and dropped.
template <typename R>
struct AeroPrep {
    bool on, clamped, has_lift;
    R F[3], bv[3], l[3];
    R ivn, vn2, c, iln, drag, lift, fs_td1, fs_td2, fs_tl1, fs_tl2, isos;
};
template <typename R>
__device__ __forceinline__ void aero_prep(const DynP<R>& p, const R* v, const R* C, AeroPrep<R>& A) {
    A.F[0] = A.F[1] = A.F[2] = R(0.0);
    A.bv[0] = C[0]; A.bv[1] = C[3]; A.bv[2] = C[6];
    A.vn2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    const R vn = sqrt(A.vn2);
    A.on = vn > R(0.0);
    A.clamped = false; A.has_lift = false;
    A.ivn = A.on ? R(1.0) / vn : R(0.0);
    A.c = A.bv[0] * v[0] + A.bv[1] * v[1] + A.bv[2] * v[2];
    A.isos = R(1.0) / p.sos;
    A.iln = R(0.0); A.drag = A.lift = R(0.0);
    A.fs_td1 = A.fs_td2 = A.fs_tl1 = A.fs_tl2 = R(0.0);
    A.l[0] = A.l[1] = A.l[2] = R(0.0);
    if (!A.on) return;
    const R mach = vn * A.isos;
    R arg = A.c / (mach * p.sos);
    if (arg < -R(1.0)) { arg = -R(1.0); A.clamped = true; }
    if (arg > R(1.0)) { arg = R(1.0); A.clamped = true; }
    R td[3], tl[3];
    aero_tables(p, arg, mach, td, tl);
    const R fs = p.force_scalar;
    A.drag = td[0] * fs; A.lift = tl[0] * fs;
    A.fs_td1 = fs * td[1]; A.fs_td2 = fs * td[2]; A.fs_tl1 = fs * tl[1]; A.fs_tl2 = fs * tl[2];
    R ld[3];
#pragma unroll
    for (int i = 0; i < 3; i++) ld[i] = A.c * v[i] - A.vn2 * A.bv[i];
    const R ln = sqrt(ld[0] * ld[0] + ld[1] * ld[1] + ld[2] * ld[2]);
    A.has_lift = ln > R(0.0);
    A.iln = A.has_lift ? R(1.0) / ln : R(0.0);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        A.l[i] = ld[i] * A.iln;
        A.F[i] = A.drag * v[i] * A.ivn + (A.has_lift ? A.lift * A.l[i] : R(0.0));
    }
}
// column j (0..3: q0..q3, 4..6: v1..v3) of d F_aero / d (q, v): the arithmetic of aero_force<true>, one column at a time
template <int J, typename R>
__device__ __forceinline__ void aero_col(const R* q, const R* v, const AeroPrep<R>& A, R d[3]) {
    d[0] = d[1] = d[2] = R(0.0);
    if (!A.on) return;
    R dbv[3] = {R(0.0), R(0.0), R(0.0)};
    if (J == 0) { dbv[1] = R(2.0) * q[3]; dbv[2] = -R(2.0) * q[2]; }
    if (J == 1) { dbv[1] = R(2.0) * q[2]; dbv[2] = R(2.0) * q[3]; }
    if (J == 2) { dbv[0] = -R(4.0) * q[2]; dbv[1] = R(2.0) * q[1]; dbv[2] = -R(2.0) * q[0]; }
    if (J == 3) { dbv[0] = -R(4.0) * q[3]; dbv[1] = R(2.0) * q[0]; dbv[2] = R(2.0) * q[1]; }
    const R ivn = A.ivn;
    R dc = R(0.0), darg, dmach;
    if (J < 4) {
        dc = dbv[0] * v[0] + dbv[1] * v[1] + dbv[2] * v[2];
        darg = A.clamped ? R(0.0) : dc * ivn;
        dmach = R(0.0);
    } else {
        darg = A.clamped ? R(0.0) : (A.bv[J - 4] * ivn - A.c * v[J - 4] * ivn * ivn * ivn);
        dmach = v[J - 4] * ivn * A.isos;
    }
    const R ddrag = A.fs_td1 * darg + A.fs_td2 * dmach;
    const R dlift = A.fs_tl1 * darg + A.fs_tl2 * dmach;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        d[i] = ddrag * v[i] * ivn;
        if (J >= 4) d[i] += A.drag * ((i == J - 4 ? ivn : R(0.0)) - v[i] * v[J - 4] * ivn * ivn * ivn);
    }
    if (A.has_lift) {
        R dld[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            if (J < 4) dld[i] = v[i] * dc - A.vn2 * dbv[i];
            else dld[i] = (i == J - 4 ? A.c : R(0.0)) + v[i] * A.bv[J - 4] - R(2.0) * A.bv[i] * v[J - 4];
        }
        const R proj = A.l[0] * dld[0] + A.l[1] * dld[1] + A.l[2] * dld[2];
#pragma unroll
        for (int i = 0; i < 3; i++) d[i] += dlift * A.l[i] + A.lift * (dld[i] - A.l[i] * proj) * A.iln;
    }
}
template <typename R>
struct FinPrep {
    bool ok;
    R fd1[3], fd2[3], b2[3], inn;
};
template <typename R>
__device__ __forceinline__ void fin_prep(const R* v, const R* C, FinPrep<R>& Fp) {
    Fp.b2[0] = C[1]; Fp.b2[1] = C[4]; Fp.b2[2] = C[7];
    const R n[3] = {Fp.b2[1] * v[2] - Fp.b2[2] * v[1], Fp.b2[2] * v[0] - Fp.b2[0] * v[2], Fp.b2[0] * v[1] - Fp.b2[1] * v[0]};
    const R nn = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    Fp.ok = nn > R(0.0);
    Fp.inn = Fp.ok ? R(1.0) / nn : R(0.0);
#pragma unroll
    for (int i = 0; i < 3; i++) Fp.fd1[i] = n[i] * Fp.inn;
    Fp.fd2[0] = Fp.fd1[1] * v[2] - Fp.fd1[2] * v[1];
    Fp.fd2[1] = Fp.fd1[2] * v[0] - Fp.fd1[0] * v[2];
    Fp.fd2[2] = Fp.fd1[0] * v[1] - Fp.fd1[1] * v[0];
}
// column j of d ff / d (q, v), ff = u4 fd1 + u5 fd2: the arithmetic of fin_dirs<true>, one column at a time
template <int J, typename R>
__device__ __forceinline__ void fin_col(const R* q, const R* v, const FinPrep<R>& Fp, R u4, R u5, R d[3]) {
    R dn[3];
    if (J < 4) {
        R c0, c1, c2;
        if (J == 0) { c0 = -R(2.0) * q[3]; c1 = R(0.0); c2 = R(2.0) * q[1]; }
        if (J == 1) { c0 = R(2.0) * q[2]; c1 = -R(4.0) * q[1]; c2 = R(2.0) * q[0]; }
        if (J == 2) { c0 = R(2.0) * q[1]; c1 = R(0.0); c2 = R(2.0) * q[3]; }
        if (J == 3) { c0 = -R(2.0) * q[0]; c1 = -R(4.0) * q[3]; c2 = R(2.0) * q[2]; }
        dn[0] = c1 * v[2] - c2 * v[1];
        dn[1] = c2 * v[0] - c0 * v[2];
        dn[2] = c0 * v[1] - c1 * v[0];
    } else {   // b2 x e_j
        if (J == 4) { dn[0] = R(0.0); dn[1] = Fp.b2[2]; dn[2] = -Fp.b2[1]; }
        if (J == 5) { dn[0] = -Fp.b2[2]; dn[1] = R(0.0); dn[2] = Fp.b2[0]; }
        if (J == 6) { dn[0] = Fp.b2[1]; dn[1] = -Fp.b2[0]; dn[2] = R(0.0); }
    }
    const R proj = Fp.fd1[0] * dn[0] + Fp.fd1[1] * dn[1] + Fp.fd1[2] * dn[2];
    R d1[3], d2[3];
#pragma unroll
    for (int i = 0; i < 3; i++) d1[i] = Fp.ok ? (dn[i] - Fp.fd1[i] * proj) * Fp.inn : R(0.0);
    d2[0] = d1[1] * v[2] - d1[2] * v[1];
    d2[1] = d1[2] * v[0] - d1[0] * v[2];
    d2[2] = d1[0] * v[1] - d1[1] * v[0];
    if (J == 4) { d2[1] += Fp.fd1[2]; d2[2] -= Fp.fd1[1]; }     // + fd1 x e_j
    if (J == 5) { d2[0] -= Fp.fd1[2]; d2[2] += Fp.fd1[0]; }
    if (J == 6) { d2[0] += Fp.fd1[1]; d2[1] -= Fp.fd1[0]; }
#pragma unroll
    for (int i = 0; i < 3; i++) d[i] = u4 * d1[i] + u5 * d2[i];
}

// ---- producer: evaluate one RK stage and PUBLISH each group of the record as soon as it exists ----
// (stage_eval + stage_publish keep the whole Stage -- 48 to 90 values -- live next to the producer's three copies of the
// state until the publish at the end; here a group is stored to LDS right after it is computed, the force derivatives are
// produced one column at a time, and only g[14] survives, which the state update needs.)  Record layout: StageRec.
// `live` = this lane owns a segment of the group.
template <bool AERO, bool FIN, typename R>
__device__ __forceinline__ void stage_eval_publish(const DynP<R>& p, const R* x, const R* u, R* g, R* rec, int stride, bool live) {
    constexpr int oC = 14, oInvm = 23, oAm = 24, oDq = 27, oMw = 39, oQ = 48, oW = 52, oKu = 55, oDv = 58;
    constexpr int oF1 = 67, oF2 = 70, oWq = 73, oWv = 85, oG1 = 94, oG2 = 97;
    const R* v = x + 4;
    const R* q = x + 7;
    const R* w = x + 11;
    auto PUT = [&](int i, R val) { if (live) rec[i * stride] = val; };
    R C[9];
    dcm(q, C);
#pragma unroll
    for (int i = 0; i < 9; i++) PUT(oC + i, C[i]);
    const R invm = R(1.0) / x[0];
    PUT(oInvm, invm);
    const R q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    const R u1 = u[0], u2 = u[1], u3 = u[2];
    R F[3] = {R(0.0), R(0.0), R(0.0)}, ff[3] = {R(0.0), R(0.0), R(0.0)};
    AeroPrep<R> A;
    FinPrep<R> Fp;
    if (AERO) {
        aero_prep(p, v, C, A);
#pragma unroll
        for (int i = 0; i < 3; i++) F[i] = A.F[i];
    }
    if (FIN) {
        fin_prep(v, C, Fp);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            ff[i] = u[3] * Fp.fd1[i] + u[4] * Fp.fd2[i];
            F[i] += ff[i];
            PUT(oF1 + i, Fp.fd1[i]);
            PUT(oF2 + i, Fp.fd2[i]);
            PUT(oG1 + i, p.JrF[3 * i] * Fp.fd1[0] + p.JrF[3 * i + 1] * Fp.fd1[1] + p.JrF[3 * i + 2] * Fp.fd1[2]);
            PUT(oG2 + i, p.JrF[3 * i] * Fp.fd2[0] + p.JrF[3 * i + 1] * Fp.fd2[1] + p.JrF[3 * i + 2] * Fp.fd2[2]);
        }
    }
    // one column of the velocity / rate rows at a time: d(C u)/dq column + force-derivative column -> Dq | Dv (and Wq | Wv)
    auto column = [&](auto Jt) {
        constexpr int J = decltype(Jt)::value;
        R d[3] = {R(0.0), R(0.0), R(0.0)};
        if (AERO) aero_col<J>(q, v, A, d);
        if (FIN) {
            R df[3];
            fin_col<J>(q, v, Fp, u[3], u[4], df);
#pragma unroll
            for (int i = 0; i < 3; i++) {
                d[i] += df[i];
                PUT((J < 4 ? oWq + 4 * i + J : oWv + 3 * i + (J - 4)), p.JrF[3 * i] * df[0] + p.JrF[3 * i + 1] * df[1] + p.JrF[3 * i + 2] * df[2]);
            }
        }
        if (J < 4) {
            R D[3];   // column J of d(C u)/dq
            if (J == 0) { D[0] = R(2.0) * (-q3 * u2 + q2 * u3); D[1] = R(2.0) * (q3 * u1 - q1 * u3); D[2] = R(2.0) * (-q2 * u1 + q1 * u2); }
            if (J == 1) { D[0] = R(2.0) * (q2 * u2 + q3 * u3); D[1] = R(2.0) * (q2 * u1 - R(2.0) * q1 * u2 - q0 * u3); D[2] = R(2.0) * (q3 * u1 + q0 * u2 - R(2.0) * q1 * u3); }
            if (J == 2) { D[0] = R(2.0) * (-R(2.0) * q2 * u1 + q1 * u2 + q0 * u3); D[1] = R(2.0) * (q1 * u1 + q3 * u3); D[2] = R(2.0) * (-q0 * u1 + q3 * u2 - R(2.0) * q2 * u3); }
            if (J == 3) { D[0] = R(2.0) * (-R(2.0) * q3 * u1 - q0 * u2 + q1 * u3); D[1] = R(2.0) * (q0 * u1 - R(2.0) * q3 * u2 + q2 * u3); D[2] = R(2.0) * (q1 * u1 + q2 * u2); }
#pragma unroll
            for (int i = 0; i < 3; i++) PUT(oDq + 4 * i + J, (D[i] + d[i]) * invm);
        } else if (AERO || FIN) {
#pragma unroll
            for (int i = 0; i < 3; i++) PUT(oDv + 3 * i + (J - 4), d[i] * invm);
        }
    };
    column(std::integral_constant<int, 0>()); column(std::integral_constant<int, 1>());
    column(std::integral_constant<int, 2>()); column(std::integral_constant<int, 3>());
    if (AERO || FIN) {
        column(std::integral_constant<int, 4>()); column(std::integral_constant<int, 5>()); column(std::integral_constant<int, 6>());
    }
    g[0] = -p.alpha * sqrt(u1 * u1 + u2 * u2 + u3 * u3);
    g[1] = v[0]; g[2] = v[1]; g[3] = v[2];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const R acc = (C[3 * i] * u1 + C[3 * i + 1] * u2 + C[3 * i + 2] * u3 + F[i]) * invm;
        g[4 + i] = acc;
        PUT(oAm + i, -acc * invm);
    }
    g[4] -= p.g0;
    g[7] = R(0.5) * (-w[0] * q1 - w[1] * q2 - w[2] * q3);
    g[8] = R(0.5) * (w[0] * q0 + w[2] * q2 - w[1] * q3);
    g[9] = R(0.5) * (w[1] * q0 - w[2] * q1 + w[0] * q3);
    g[10] = R(0.5) * (w[2] * q0 + w[1] * q1 - w[0] * q2);
    {
        R Jw[3], t[3];
#pragma unroll
        for (int i = 0; i < 3; i++) Jw[i] = p.J[3 * i] * w[0] + p.J[3 * i + 1] * w[1] + p.J[3 * i + 2] * w[2];
        t[0] = (p.rTB[1] * u3 - p.rTB[2] * u2) - (w[1] * Jw[2] - w[2] * Jw[1]);
        t[1] = (p.rTB[2] * u1 - p.rTB[0] * u3) - (w[2] * Jw[0] - w[0] * Jw[2]);
        t[2] = (p.rTB[0] * u2 - p.rTB[1] * u1) - (w[0] * Jw[1] - w[1] * Jw[0]);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            g[11 + i] = p.Jinv[3 * i] * t[0] + p.Jinv[3 * i + 1] * t[1] + p.Jinv[3 * i + 2] * t[2];
            if (FIN) g[11 + i] += p.JrF[3 * i] * ff[0] + p.JrF[3 * i + 1] * ff[1] + p.JrF[3 * i + 2] * ff[2];
        }
        // T = [w]x J - [Jw]x ; Mw = -Jinv T
        R T[9];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            T[0 + j] = -w[2] * p.J[3 + j] + w[1] * p.J[6 + j];
            T[3 + j] = w[2] * p.J[0 + j] - w[0] * p.J[6 + j];
            T[6 + j] = -w[1] * p.J[0 + j] + w[0] * p.J[3 + j];
        }
        T[1] += Jw[2]; T[2] -= Jw[1];
        T[3] -= Jw[2]; T[5] += Jw[0];
        T[6] += Jw[1]; T[7] -= Jw[0];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++)
                PUT(oMw + 3 * i + j, -(p.Jinv[3 * i] * T[j] + p.Jinv[3 * i + 1] * T[3 + j] + p.Jinv[3 * i + 2] * T[6 + j]));
    }
#pragma unroll
    for (int i = 0; i < 14; i++) PUT(i, g[i]);
#pragma unroll
    for (int i = 0; i < 4; i++) PUT(oQ + i, q[i]);
#pragma unroll
    for (int i = 0; i < 3; i++) PUT(oW + i, w[i]);
    {
        const R un = sqrt(u1 * u1 + u2 * u2 + u3 * u3);
        const R k = un > R(0.0) ? -p.alpha / un : R(0.0);
        PUT(oKu, k * u1); PUT(oKu + 1, k * u2); PUT(oKu + 2, k * u3);
    }
}

// ---- the producer's stage split over TWO wavefronts (aero / fin models: linearize_pcp2_kernel) ----
// With aerodynamics the producer is the block's bottleneck (its consumers wait 65-80 % of the kernel at the stage barriers,
// profiles/r04_k1_split.md): one stage is the spline tables, the force, seven force-derivative columns and the rate Jacobian, a
// serial chain on one wavefront.  Only the first half of that feeds the state recurrence.  So wavefront P0 runs the STATE path --
// rotation matrix, tables, force, right-hand side -- publishes the part of the record it owns and hands (q, v, w, u, 1/m and the
// prepared aerodynamic / fin quantities: HandRec) to wavefront P1, which one stage later produces the DERIVATIVE part of the
// same record: the seven columns of d(C u + F)/d(q, v) / m, the fin torque columns and the rate Jacobian.  Same arithmetic as
// stage_eval_publish, statement for statement.
template <bool FIN> struct HandRec {
    // q 0..3 | v 4..6 | w 7..9 | u 10..14 | invm 15 | flags 16 | ivn vn2 c iln drag lift 17..22 | fs_td1 fs_td2 fs_tl1 fs_tl2 23..26 |
    // bv 27..29 | l 30..32 | fin: fd1 33..35 | b2 36..38 | inn 39
    static constexpr int N = FIN ? 40 : 33;
};
template <bool AERO, bool FIN, typename R>
__device__ __forceinline__ void stage_state_publish(const DynP<R>& p, const R* x, const R* u, R* g, R* rec, int stride, bool live,
                                                    R* hand, int hstride) {
    constexpr int oC = 14, oInvm = 23, oAm = 24, oQ = 48, oW = 52, oKu = 55;
    constexpr int oF1 = 67, oF2 = 70, oG1 = 94, oG2 = 97;
    const R* v = x + 4;
    const R* q = x + 7;
    const R* w = x + 11;
    auto PUT = [&](int i, R val) { if (live) rec[i * stride] = val; };
    auto HPUT = [&](int i, R val) { if (live) hand[i * hstride] = val; };
    R C[9];
    dcm(q, C);
#pragma unroll
    for (int i = 0; i < 9; i++) PUT(oC + i, C[i]);
    const R invm = R(1.0) / x[0];
    PUT(oInvm, invm);
    const R q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    const R u1 = u[0], u2 = u[1], u3 = u[2];
    R F[3] = {R(0.0), R(0.0), R(0.0)}, ff[3] = {R(0.0), R(0.0), R(0.0)};
    AeroPrep<R> A;
    FinPrep<R> Fp;
    int flags = 0;
    if (AERO) {
        aero_prep(p, v, C, A);
#pragma unroll
        for (int i = 0; i < 3; i++) F[i] = A.F[i];
        flags |= (A.on ? 1 : 0) | (A.clamped ? 2 : 0) | (A.has_lift ? 4 : 0);
        HPUT(17, A.ivn); HPUT(18, A.vn2); HPUT(19, A.c); HPUT(20, A.iln); HPUT(21, A.drag); HPUT(22, A.lift);
        HPUT(23, A.fs_td1); HPUT(24, A.fs_td2); HPUT(25, A.fs_tl1); HPUT(26, A.fs_tl2);
#pragma unroll
        for (int i = 0; i < 3; i++) { HPUT(27 + i, A.bv[i]); HPUT(30 + i, A.l[i]); }
    }
    if (FIN) {
        fin_prep(v, C, Fp);
        flags |= Fp.ok ? 8 : 0;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            ff[i] = u[3] * Fp.fd1[i] + u[4] * Fp.fd2[i];
            F[i] += ff[i];
            PUT(oF1 + i, Fp.fd1[i]);
            PUT(oF2 + i, Fp.fd2[i]);
            PUT(oG1 + i, p.JrF[3 * i] * Fp.fd1[0] + p.JrF[3 * i + 1] * Fp.fd1[1] + p.JrF[3 * i + 2] * Fp.fd1[2]);
            PUT(oG2 + i, p.JrF[3 * i] * Fp.fd2[0] + p.JrF[3 * i + 1] * Fp.fd2[1] + p.JrF[3 * i + 2] * Fp.fd2[2]);
            HPUT(33 + i, Fp.fd1[i]); HPUT(36 + i, Fp.b2[i]);
        }
        HPUT(39, Fp.inn);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) HPUT(i, q[i]);
#pragma unroll
    for (int i = 0; i < 3; i++) { HPUT(4 + i, v[i]); HPUT(7 + i, w[i]); }
#pragma unroll
    for (int i = 0; i < (FIN ? 5 : 3); i++) HPUT(10 + i, u[i]);
    HPUT(15, invm);
    HPUT(16, (R)flags);
    g[0] = -p.alpha * sqrt(u1 * u1 + u2 * u2 + u3 * u3);
    g[1] = v[0]; g[2] = v[1]; g[3] = v[2];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const R acc = (C[3 * i] * u1 + C[3 * i + 1] * u2 + C[3 * i + 2] * u3 + F[i]) * invm;
        g[4 + i] = acc;
        PUT(oAm + i, -acc * invm);
    }
    g[4] -= p.g0;
    g[7] = R(0.5) * (-w[0] * q1 - w[1] * q2 - w[2] * q3);
    g[8] = R(0.5) * (w[0] * q0 + w[2] * q2 - w[1] * q3);
    g[9] = R(0.5) * (w[1] * q0 - w[2] * q1 + w[0] * q3);
    g[10] = R(0.5) * (w[2] * q0 + w[1] * q1 - w[0] * q2);
    {
        R Jw[3], t[3];
#pragma unroll
        for (int i = 0; i < 3; i++) Jw[i] = p.J[3 * i] * w[0] + p.J[3 * i + 1] * w[1] + p.J[3 * i + 2] * w[2];
        t[0] = (p.rTB[1] * u3 - p.rTB[2] * u2) - (w[1] * Jw[2] - w[2] * Jw[1]);
        t[1] = (p.rTB[2] * u1 - p.rTB[0] * u3) - (w[2] * Jw[0] - w[0] * Jw[2]);
        t[2] = (p.rTB[0] * u2 - p.rTB[1] * u1) - (w[0] * Jw[1] - w[1] * Jw[0]);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            g[11 + i] = p.Jinv[3 * i] * t[0] + p.Jinv[3 * i + 1] * t[1] + p.Jinv[3 * i + 2] * t[2];
            if (FIN) g[11 + i] += p.JrF[3 * i] * ff[0] + p.JrF[3 * i + 1] * ff[1] + p.JrF[3 * i + 2] * ff[2];
        }
    }
#pragma unroll
    for (int i = 0; i < 14; i++) PUT(i, g[i]);
#pragma unroll
    for (int i = 0; i < 4; i++) PUT(oQ + i, q[i]);
#pragma unroll
    for (int i = 0; i < 3; i++) PUT(oW + i, w[i]);
    {
        const R un = sqrt(u1 * u1 + u2 * u2 + u3 * u3);
        const R k = un > R(0.0) ? -p.alpha / un : R(0.0);
        PUT(oKu, k * u1); PUT(oKu + 1, k * u2); PUT(oKu + 2, k * u3);
    }
}
template <bool AERO, bool FIN, typename R>
__device__ __forceinline__ void stage_cols_publish(const DynP<R>& p, const R* hand, int hstride, R* rec, int stride, bool live) {
    constexpr int oDq = 27, oMw = 39, oDv = 58, oWq = 73, oWv = 85;
    auto PUT = [&](int i, R val) { if (live) rec[i * stride] = val; };
    auto H = [&](int i) { return hand[i * hstride]; };
    R q[4], v[3], w[3], u[5] = {R(0.0), R(0.0), R(0.0), R(0.0), R(0.0)};
#pragma unroll
    for (int i = 0; i < 4; i++) q[i] = H(i);
#pragma unroll
    for (int i = 0; i < 3; i++) { v[i] = H(4 + i); w[i] = H(7 + i); }
#pragma unroll
    for (int i = 0; i < (FIN ? 5 : 3); i++) u[i] = H(10 + i);
    const R invm = H(15);
    const int flags = (int)H(16);
    const R q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    const R u1 = u[0], u2 = u[1], u3 = u[2];
    AeroPrep<R> A;
    FinPrep<R> Fp;
    if (AERO) {
        A.on = (flags & 1) != 0; A.clamped = (flags & 2) != 0; A.has_lift = (flags & 4) != 0;
        A.ivn = H(17); A.vn2 = H(18); A.c = H(19); A.iln = H(20); A.drag = H(21); A.lift = H(22);
        A.fs_td1 = H(23); A.fs_td2 = H(24); A.fs_tl1 = H(25); A.fs_tl2 = H(26);
#pragma unroll
        for (int i = 0; i < 3; i++) { A.bv[i] = H(27 + i); A.l[i] = H(30 + i); A.F[i] = R(0.0); }
        A.isos = R(1.0) / p.sos;
    }
    if (FIN) {
        Fp.ok = (flags & 8) != 0;
#pragma unroll
        for (int i = 0; i < 3; i++) { Fp.fd1[i] = H(33 + i); Fp.b2[i] = H(36 + i); Fp.fd2[i] = R(0.0); }
        Fp.inn = H(39);
    }
    auto column = [&](auto Jt) {
        constexpr int J = decltype(Jt)::value;
        R d[3] = {R(0.0), R(0.0), R(0.0)};
        if (AERO) aero_col<J>(q, v, A, d);
        if (FIN) {
            R df[3];
            fin_col<J>(q, v, Fp, u[3], u[4], df);
#pragma unroll
            for (int i = 0; i < 3; i++) {
                d[i] += df[i];
                PUT((J < 4 ? oWq + 4 * i + J : oWv + 3 * i + (J - 4)), p.JrF[3 * i] * df[0] + p.JrF[3 * i + 1] * df[1] + p.JrF[3 * i + 2] * df[2]);
            }
        }
        if (J < 4) {
            R D[3];   // column J of d(C u)/dq
            if (J == 0) { D[0] = R(2.0) * (-q3 * u2 + q2 * u3); D[1] = R(2.0) * (q3 * u1 - q1 * u3); D[2] = R(2.0) * (-q2 * u1 + q1 * u2); }
            if (J == 1) { D[0] = R(2.0) * (q2 * u2 + q3 * u3); D[1] = R(2.0) * (q2 * u1 - R(2.0) * q1 * u2 - q0 * u3); D[2] = R(2.0) * (q3 * u1 + q0 * u2 - R(2.0) * q1 * u3); }
            if (J == 2) { D[0] = R(2.0) * (-R(2.0) * q2 * u1 + q1 * u2 + q0 * u3); D[1] = R(2.0) * (q1 * u1 + q3 * u3); D[2] = R(2.0) * (-q0 * u1 + q3 * u2 - R(2.0) * q2 * u3); }
            if (J == 3) { D[0] = R(2.0) * (-R(2.0) * q3 * u1 - q0 * u2 + q1 * u3); D[1] = R(2.0) * (q0 * u1 - R(2.0) * q3 * u2 + q2 * u3); D[2] = R(2.0) * (q1 * u1 + q2 * u2); }
#pragma unroll
            for (int i = 0; i < 3; i++) PUT(oDq + 4 * i + J, (D[i] + d[i]) * invm);
        } else if (AERO || FIN) {
#pragma unroll
            for (int i = 0; i < 3; i++) PUT(oDv + 3 * i + (J - 4), d[i] * invm);
        }
    };
    column(std::integral_constant<int, 0>()); column(std::integral_constant<int, 1>());
    column(std::integral_constant<int, 2>()); column(std::integral_constant<int, 3>());
    column(std::integral_constant<int, 4>()); column(std::integral_constant<int, 5>()); column(std::integral_constant<int, 6>());
    {   // T = [w]x J - [Jw]x ; Mw = -Jinv T
        R Jw[3], T[9];
#pragma unroll
        for (int i = 0; i < 3; i++) Jw[i] = p.J[3 * i] * w[0] + p.J[3 * i + 1] * w[1] + p.J[3 * i + 2] * w[2];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            T[0 + j] = -w[2] * p.J[3 + j] + w[1] * p.J[6 + j];
            T[3 + j] = w[2] * p.J[0 + j] - w[0] * p.J[6 + j];
            T[6 + j] = -w[1] * p.J[0 + j] + w[0] * p.J[3 + j];
        }
        T[1] += Jw[2]; T[2] -= Jw[1];
        T[3] -= Jw[2]; T[5] += Jw[0];
        T[6] += Jw[1]; T[7] -= Jw[0];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++)
                PUT(oMw + 3 * i + j, -(p.Jinv[3 * i] * T[j] + p.Jinv[3 * i + 1] * T[3 + j] + p.Jinv[3 * i + 2] * T[6 + j]));
    }
}

// d/dt of one sensitivity column from a published stage record, read in THREE BATCHES (mass / position / velocity rows | attitude
// and rate rows | the RHS itself): at most ~45 of the record's 58 / 67 / 100 values are live at once next to the column's own 42
// (c, its RK accumulator and stage value).  Reading the whole record in one batch (round 2's column_deriv_rec) keeps one LDS round
// trip per stage but spilled (round 2: 97 / 240 VGPRs in the persistent kernel, exo / aero).
// wc[NU]: FOH weights of this column's control component (NU = 5 with FIN, else 3).
template <bool AERO, bool FIN, typename R>
__device__ __forceinline__ void column_deriv_rec_pieces(const DynP<R>& p, const R* rec, int stride, const R* c,
                                                        const R* wc, R gsel, R sigma, R* dc) {
    constexpr bool DV = AERO || FIN;
    constexpr int oC = 14, oInvm = 23, oAm = 24, oDq = 27, oMw = 39, oQ = 48, oW = 52, oKu = 55, oDv = 58;
    constexpr int oF1 = 67, oF2 = 70, oWq = 73, oWv = 85, oG1 = 94, oG2 = 97;
    auto RR = [&](int i) { return rec[i * stride]; };
    R a[14];
    {   // ---- mass, position and velocity rows ----
        R C[9], am[3], Dq[12], Dv[DV ? 9 : 1], f1[FIN ? 3 : 1], f2[FIN ? 3 : 1], ku[3];
#pragma unroll
        for (int i = 0; i < 9; i++) C[i] = RR(oC + i);
        if (DV) {
#pragma unroll
            for (int i = 0; i < 9; i++) Dv[i] = RR(oDv + i);
        }
#pragma unroll
        for (int i = 0; i < 12; i++) Dq[i] = RR(oDq + i);
#pragma unroll
        for (int i = 0; i < 3; i++) { am[i] = RR(oAm + i); ku[i] = RR(oKu + i); }
        if (FIN) {
#pragma unroll
            for (int i = 0; i < 3; i++) { f1[i] = RR(oF1 + i); f2[i] = RR(oF2 + i); }
        }
        const R invm = RR(oInvm);
        a[0] = ku[0] * wc[0] + ku[1] * wc[1] + ku[2] * wc[2];
        a[1] = c[4]; a[2] = c[5]; a[3] = c[6];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            R t = am[i] * c[0];
            t = fma(Dq[4 * i], c[7], t);
            t = fma(Dq[4 * i + 1], c[8], t);
            t = fma(Dq[4 * i + 2], c[9], t);
            t = fma(Dq[4 * i + 3], c[10], t);
            if (DV) {
                t = fma(Dv[3 * i], c[4], t);
                t = fma(Dv[3 * i + 1], c[5], t);
                t = fma(Dv[3 * i + 2], c[6], t);
            }
            R cw_ = C[3 * i] * wc[0] + C[3 * i + 1] * wc[1] + C[3 * i + 2] * wc[2];
            if (FIN) cw_ += f1[i] * wc[3] + f2[i] * wc[4];
            t = fma(cw_, invm, t);
            a[4 + i] = t;
        }
    }
    {   // ---- attitude and rate rows ----
        R q[4], w[3], Mw[9], Wq[FIN ? 12 : 1], Wv[FIN ? 9 : 1], g1[FIN ? 3 : 1], g2[FIN ? 3 : 1];
#pragma unroll
        for (int i = 0; i < 4; i++) q[i] = RR(oQ + i);
#pragma unroll
        for (int i = 0; i < 3; i++) w[i] = RR(oW + i);
#pragma unroll
        for (int i = 0; i < 9; i++) Mw[i] = RR(oMw + i);
        if (FIN) {
#pragma unroll
            for (int i = 0; i < 3; i++) { g1[i] = RR(oG1 + i); g2[i] = RR(oG2 + i); }
#pragma unroll
            for (int i = 0; i < 9; i++) Wv[i] = RR(oWv + i);
#pragma unroll
            for (int i = 0; i < 12; i++) Wq[i] = RR(oWq + i);
        }
        const R cq0 = c[7], cq1 = c[8], cq2 = c[9], cq3 = c[10];
        const R cw0 = c[11], cw1 = c[12], cw2 = c[13];
        a[7] = R(0.5) * (-w[0] * cq1 - w[1] * cq2 - w[2] * cq3 - q[1] * cw0 - q[2] * cw1 - q[3] * cw2);
        a[8] = R(0.5) * (w[0] * cq0 + w[2] * cq2 - w[1] * cq3 + q[0] * cw0 - q[3] * cw1 + q[2] * cw2);
        a[9] = R(0.5) * (w[1] * cq0 - w[2] * cq1 + w[0] * cq3 + q[3] * cw0 + q[0] * cw1 - q[1] * cw2);
        a[10] = R(0.5) * (w[2] * cq0 + w[1] * cq1 - w[0] * cq2 - q[2] * cw0 + q[1] * cw1 + q[0] * cw2);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            R t = Mw[3 * i] * cw0;
            t = fma(Mw[3 * i + 1], cw1, t);
            t = fma(Mw[3 * i + 2], cw2, t);
            if (FIN) {
                t = fma(Wq[4 * i], cq0, t);
                t = fma(Wq[4 * i + 1], cq1, t);
                t = fma(Wq[4 * i + 2], cq2, t);
                t = fma(Wq[4 * i + 3], cq3, t);
                t = fma(Wv[3 * i], c[4], t);
                t = fma(Wv[3 * i + 1], c[5], t);
                t = fma(Wv[3 * i + 2], c[6], t);
            }
            t = fma(p.JrT[3 * i], wc[0], t);
            t = fma(p.JrT[3 * i + 1], wc[1], t);
            t = fma(p.JrT[3 * i + 2], wc[2], t);
            if (FIN) {
                t = fma(g1[i], wc[3], t);
                t = fma(g2[i], wc[4], t);
            }
            a[11 + i] = t;
        }
    }
    R g[14];
#pragma unroll
    for (int i = 0; i < 14; i++) g[i] = RR(i);
#pragma unroll
    for (int i = 0; i < 14; i++) dc[i] = fma(sigma, a[i], gsel * g[i]);
}

}  // namespace scvx
