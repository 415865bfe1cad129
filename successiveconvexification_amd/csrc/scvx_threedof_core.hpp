// K0: the 3-DoF point-mass landing SOCP of FirstRound.solve_initial (initial_solve.jl:17-110, inside a #= =# block at
// HEAD; SURVEY.md 8f rank 2, BASELINE configs[0]) as a batched conic solve -- one wavefront per trajectory.
//
// The problem (all of it from initial_solve.jl, line numbers in tables() below): per node i = 0..N (N = K)
//     variables   r_i(3) v_i(3) ma_i | T_i(3) ga_i kaR_i ar_i(3)           and one global nkaR          (:49-58)
//     objective   min -ma_N + 100 nkaR                                                                    (:68, :39)
//     equalities  r_0 = rIi, v_0 = vIi, ma_0 = mwet, r_N = v_N = 0, T_N[2:3] = 0                          (:59-65)
//                 trapezoidal mass / position / velocity recursions with the fixed mass profile mu        (:24, :72-78)
//     cones       ma >= mdry, Tmin <= ga <= Tmax, ga cos(thetaMax) <= T1, glideslope SOC3, |T| <= ga SOC4,
//                 |ar| <= kaR SOC4 per node (:80-88); [nkaR; kaR_0..kaR_N] in SOC(N+2)                    (:69-70)
//
// Algorithm: the same infeasible-start Mehrotra predictor-corrector / Nesterov-Todd method as the oracle's generic
// solver (oracle/ipm.py, Vandenberghe's CVXOPT notation), with the linear algebra laid out for one wavefront:
//   * unknowns ordered BY NODE -- [initial rows | z_0 | dyn_0 | z_1 | dyn_1 | ... | z_N | final rows] -- so the condensed
//     KKT matrix [H + delta I, A'; A, -delta I] (H = E' W^-2 E, block diagonal per node) is BANDED, half bandwidth 19,
//     22 positions per node.  It is quasi-definite, so an LDL' without pivoting exists in this order; the factorisation
//     slides a 21-column window through LDS (one LDS round trip per column), the triangular solves keep the vector
//     in LDS and stream L from HBM.
//   * the one long cone couples every node through W^-2 = beta^-2 (2 w w' - J).  Its -J part is diagonal; the head nkaR
//     and the rank-one part go into a 2-unknown border: one extra banded solve per factorisation, a 2x2 system per solve.
//   * the condensed system squares the conditioning of W, so every Newton solve is followed by refinement passes on
//     the UNCONDENSED residual (rows [A' y - E' z; A x; -E x - W^2 z]), which restores the accuracy of a solver that
//     keeps dz as an unknown.  With one pass the iteration counts equal the oracle's (16 and 21 on its two test cases).
//
// The problem structure lives in tables built once on the host (Tables / build_tables) and shared by every trajectory
// of a batch: only the initial position and velocity differ.  Compiled for the device by
// scvx_threedof.hip and for the host (the CPU twin the tests use) by oracle/scvx_port.cpp.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <vector>
#include "scvx_ipm_core.hpp"

#if defined(SCVX_TD_PROF) && defined(__HIP_DEVICE_COMPILE__)
#define TD_TS(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define TD_TE(v, slot) prof[slot] += (double)(__builtin_amdgcn_s_memtime() - v)
#else
#define TD_TS(v)
#define TD_TE(v, slot)
#endif

// Refinement passes of the PREDICTOR solve.  Its direction only sets the centring parameter, so the unrefined solve
// is enough: measured on 2 x 128 dispersed instances, the iteration counts are unchanged (15.95 and 22.39 on average).
// The corrector solve is refined only once max(primal residual, relative gap) is below this: far from the optimum the
// condensed solve is accurate enough (W is well conditioned there): iteration counts unchanged for thresholds from 1e-2 down
// to 1e-5 on 3 x 128 dispersed instances (unflyable K = 30, flyable K = 30 and 50); about 60 % of the refinements go.
#ifndef TD_REFINE_FROM
#define TD_REFINE_FROM 1e-3
#endif
#ifndef TD_REFINE_MAX
#define TD_REFINE_MAX 4
#endif
#ifndef TD_REFINE_MORE
#define TD_REFINE_MORE 1e-2
#endif
#ifndef TD_PRED_REFINE
#define TD_PRED_REFINE 0
#endif

namespace scvx {
namespace td {

// Address spaces (device build): the solver's vectors, L and the tables are HBM (global_load/store, vmcnt only), the
// factorisation window and the solve vector are LDS (ds_read/write, lgkmcnt only).  A plain `double*` kept in the solver
// object loses its address space as soon as a routine is not inlined and every access becomes a FLAT one, which counts
// against both counters -- each LDS step of the factorisation then also waits for the L stores in flight.
#if defined(__HIP_DEVICE_COMPILE__)
#define TD_LOCAL __attribute__((address_space(3)))
#define TD_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)   // nothing is scheduled across this point
#define TD_UNROLL8 _Pragma("unroll 8")   // per-cone loops: the long cone (N + 2 rows) is walked by one lane, keep 8 loads in flight
#else
#define TD_LOCAL
#define TD_SCHED_FENCE()
#define TD_UNROLL8
#endif
typedef SCVX_GLOBAL double* gptr;
typedef const SCVX_GLOBAL double* cgptr;
typedef const SCVX_GLOBAL int* cgiptr;
typedef TD_LOCAL double* lptr;
typedef TD_LOCAL int* liptr;

constexpr int BW = 19;        // half bandwidth of the node-ordered KKT matrix
constexpr int BS = BW + 1;    // entries stored per band column (diagonal + BW below)
constexpr int NV = 15;        // variables per node: r(0..2) v(3..5) ma(6) T(7..9) ga(10) kaR(11) ar(12..14)
constexpr int NP = 22;        // band positions per node: NV variables + 7 multipliers of the recursion to node i+1
constexpr int NR = 15;        // cone rows per node: 4 linear, SOC3 (glideslope), SOC4 (thrust), SOC4 (virtual acceleration)
constexpr int AW = 8;         // ELL width of the symmetric equality part [0 A'; A 0]
constexpr int TW = 4;         // ELL width of E'
constexpr int ST = 44;       // columns staged ahead of the factorisation window (fetched a block early, in registers)
constexpr int LPAD = 1024;   // zero padding (doubles) before and after L and the constant band: the sweeps and the window fetch
                             // load unconditionally a little outside [0, nb) and mask, instead of branching around each load
constexpr int NSLOT = BS + ST; // LDS window of the factorisation: the BS live columns + the ST that become live during a block

// status codes of one solve (the K4 solver's, scvx.h)
enum { TD_OPTIMAL = 0, TD_ITER_CAP = 1, TD_STALLED = 2, TD_NONFINITE = 3, TD_ALMOST = 4, TD_INFEASIBLE = 5 };

struct Tables {
    int N, nb, m, ncone;      // nodes - 1, band size (nkaR sits at position nb), cone rows, cones
    int max_iter, refine;
    double tol, delta;
    double itg, cth;          // 1 / tan(gammaGs), cos(thetaMax): the two non-unit coefficients of the cone rows
    double nrm_c, nrm_h, b2_rest, mwet;   // max(1,|c|), max(1,|h|), sum of b^2 without the six initial r, v rows
    const int* a_col;     // [nb][AW]      symmetric equality part, -1 = empty
    const double* a_val;  // [nb][AW]
    const double* kc;     // [nb][BS]      its lower band, +delta / -delta on the diagonal
    const int* e_v0;      // [m]           cone row rho: e_rho(x) = c0 x[v0] + c1 x[v1] + h   (s = e(x), G = -E)
    const int* e_v1;      // [m]           -1 = none
    const double* e_c0;
    const double* e_c1;
    const double* e_h;
    const int* t_row;     // [nb+1][TW]    E': rows each band position appears in
    const double* t_coef; // [nb+1][TW]
    const double* q;      // [nb+1]        c at variable positions, b at equality positions (initial r, v rows per trajectory)
};

SCVX_HD int pos_z(int i, int l) { return 7 + NP * i + l; }
SCVX_HD int pos_dyn(int i, int j) { return 7 + NP * i + NV + j; }
SCVX_HD int pos_fin(int N, int j) { return 7 + NP * N + NV + j; }
SCVX_HD int band_size(int N) { return NP * N + 30; }
SCVX_HD int cone_rows(int N) { return NR * (N + 1) + N + 2; }
SCVX_HD int cone_count(int N) { return 7 * (N + 1) + 1; }
SCVX_HD bool is_var(int N, int pos) { return pos >= 7 && pos < 7 + NP * N + NV && (pos - 7) % NP < NV; }
// cone c -> first row and dimension (1 = a linear row); the long cone is the last
SCVX_HD void cone_of(int N, int c, int& off, int& dim) {
    if (c >= 7 * (N + 1)) { off = NR * (N + 1); dim = N + 2; return; }
    const int i = c / 7, k = c - 7 * i;
    if (k < 4) { off = NR * i + k; dim = 1; }
    else if (k == 4) { off = NR * i + 4; dim = 3; }
    else if (k == 5) { off = NR * i + 7; dim = 4; }
    else { off = NR * i + 11; dim = 4; }
}

// per-trajectory workspace (doubles)
struct Layout {
    int N, nb, m, ncone;
    size_t u, s, z, lam, wv, wb, ru, rz, du, dz, ds, bu, bz, t1, t2, t3, tu, tu2, hd, lb, ut, y, total;
    SCVX_HD void init(int N_) {
        N = N_; nb = band_size(N); m = cone_rows(N); ncone = cone_count(N);
        size_t o = 0;
        auto take = [&](size_t n) { const size_t r = o; o += (n + 7) & ~(size_t)7; return r; };
        const size_t nu = (size_t)nb + 1, mm = (size_t)m;
        u = take(nu); s = take(mm); z = take(mm); lam = take(mm); wv = take(mm); wb = take((size_t)ncone);
        ru = take(nu); rz = take(mm); du = take(nu); dz = take(mm); ds = take(mm); bu = take(nu); bz = take(mm);
        t1 = take(mm); t2 = take(mm); t3 = take(mm); tu = take(nu); tu2 = take(nu);
        hd = take((size_t)(N + 1) * NV * NV + 8);   // + a zero entry the window fetch reads for positions outside a node block
        lb = take((size_t)nb * BS + 2 * LPAD) + LPAD; ut = take((size_t)nb * BS + 2 * LPAD) + LPAD; y = take(nu);
        total = o;
    }
};
constexpr int NPAIR = BW * (BW + 1) / 2;   // entries of the trailing update of one column
SCVX_HD size_t fast_doubles(int N) {       // LDS per trajectory: window, solve vector, pair table (ints)
    return (size_t)NSLOT * BS + (((size_t)band_size(N) + 7) & ~(size_t)7) + (NPAIR + 1) / 2 + 8 + NSLOT;
}
SCVX_HD int out_doubles(int N) { return (N + 1) * NV + 1; }

struct Result {
    int status, iters;
    double pobj, gap, pres, dres;
};

// ------------------------------------------------------------------------------------------------------------------
// cone arithmetic on memory-resident rows (dim 1 = linear); formulas of oracle/ipm.py::Cone.
//
// Every routine is written once over a ROW RANGE policy:
//   Serial      one lane owns the whole cone (the 4 linear rows and the 3 small cones of a node);
//   Coop<Ex>    the long cone [nkaR; kaR_0..kaR_N]: row k belongs to lane k mod 64, lane 0 owns the head; lanes exchange
//               values only through wavefront reductions (sum1 / head), so a lane never reads a row another lane wrote in
//               the same pass and fused sequences (W^-1 W^-1 x, lam \ d followed by W ...) need no barrier.
// ------------------------------------------------------------------------------------------------------------------
// rows k0..q-1 in blocks of 8: all loads of a block (ld) before its stores (st), so that they are in flight together
// even where the output aliases an input (in-place W, Jordan product, division)
template <class LD, class ST_>
SCVX_HD void rows8(int k0, int q, LD&& ld, ST_&& st) {
    for (int k = k0; k < q; k += 8) {
        double t[8];
        SCVX_UNROLL
        for (int i = 0; i < 8; i++) t[i] = ld(k + i < q ? k + i : q - 1);
        SCVX_UNROLL
        for (int i = 0; i < 8; i++) if (k + i < q) st(k + i, t[i]);
    }
}
struct Serial {
    int q;
    SCVX_HD bool owns_head() const { return true; }
    SCVX_HD double head(cgptr x) const { return x[0]; }
    SCVX_HD double all(double x) const { return x; }
    template <class F> SCVX_HD double sum1(F&& f) const {   // over the tail rows 1..q-1
        double a = 0;
        TD_UNROLL8
        for (int k = 1; k < q; k++) a += f(k);
        return a;
    }
    template <class LD, class ST_> SCVX_HD void rows(int k0, LD&& ld, ST_&& st) const { rows8(k0, q, ld, st); }
};
template <class Ex>
struct Coop {
    Ex& ex;
    int q;
    SCVX_HD bool owns_head() const { return ex.lane() == 0; }
    SCVX_HD double head(cgptr x) const { return ex.sum(ex.lane() == 0 ? x[0] : 0.0); }
    SCVX_HD double all(double x) const { return ex.sum(ex.lane() == 0 ? x : 0.0); }   // a value lane 0 holds, to every lane
    template <class F> SCVX_HD double sum1(F&& f) const {
        double a = 0;
        for (int k = ex.lane(); k < q; k += ex.nlanes()) if (k >= 1) a += f(k);
        return ex.sum(a);
    }
    template <class LD, class ST_> SCVX_HD void rows(int k0, LD&& ld, ST_&& st) const {
        for (int k = ex.lane(); k < q; k += ex.nlanes()) if (k >= k0) st(k, ld(k));
    }
};

template <class R>
SCVX_HD void cone_nt(R rg, cgptr s, cgptr z, gptr v, double& beta, gptr lam) {
    if (rg.q == 1) { v[0] = sqrt(s[0] / z[0]); beta = 1.0; lam[0] = sqrt(s[0] * z[0]); return; }
    const double s0 = rg.head(s), z0 = rg.head(z);
    const double s1 = rg.sum1([&](int k) { return s[k] * s[k]; }), z1 = rg.sum1([&](int k) { return z[k] * z[k]; });
    const double sz = rg.sum1([&](int k) { return s[k] * z[k]; });
    const double ns = sqrt(s1), nz = sqrt(z1);
    // an iterate on the boundary to rounding (s0 - |s1| <= 0 in floating point) is kept a rounding error inside
    const double sj = sqrt(fmax(s0 - ns, 2.3e-16 * s0) * (s0 + ns)), zj = sqrt(fmax(z0 - nz, 2.3e-16 * z0) * (z0 + nz));
    const double isj = 1.0 / sj, izj = 1.0 / zj;
    const double gam = sqrt(0.5 * (1.0 + (s0 * z0 + sz) * isj * izj));
    const double ig = 0.5 / gam;
    const double wb0 = (s0 * isj + z0 * izj) * ig;
    const double den = 1.0 / sqrt(2.0 * (wb0 + 1.0));
    const double v0 = (wb0 + 1.0) * den;
    const double c1 = ig * den;
    beta = sqrt(sj * izj);
    // v tail, and lam = W z = beta (2 (v'z) v - J z); v'z from s and z alone so that no lane reads another lane's v
    const double vz = v0 * z0 + c1 * (sz * isj - z1 * izj);
    const double bt = beta;
    rg.rows(1, [&](int k) { return (s[k] * isj - z[k] * izj) * c1; }, [&](int k, double t) { v[k] = t; lam[k] = bt * (2.0 * vz * t + z[k]); });
    if (rg.owns_head()) { v[0] = v0; lam[0] = bt * (2.0 * vz * v0 - z0); }
}
// y = W x or W^-1 x (y may alias x)
template <class R>
SCVX_HD void cone_W(R rg, cgptr v, double beta, cgptr x, gptr y, bool inverse) {
    if (rg.q == 1) { y[0] = inverse ? x[0] / v[0] : x[0] * v[0]; return; }
    const double v0 = rg.head(v), x0 = rg.head(x);
    const double sg = inverse ? -1.0 : 1.0;
    const double vx = v0 * x0 + sg * rg.sum1([&](int k) { return v[k] * x[k]; });
    const double sc = inverse ? 1.0 / beta : beta;
    const double tw = 2.0 * sg * vx;
    rg.rows(1, [&](int k) { return (tw * v[k] + x[k]) * sc; }, [&](int k, double t) { y[k] = t; });
    if (rg.owns_head()) y[0] = (2.0 * vx * v0 - x0) * sc;
}
// o = a o b (Jordan product); o may alias a or b
template <class R>
SCVX_HD void cone_prod(R rg, cgptr a, cgptr b, gptr o) {
    if (rg.q == 1) { o[0] = a[0] * b[0]; return; }
    const double a0 = rg.head(a), b0 = rg.head(b);
    const double dot = a0 * b0 + rg.sum1([&](int k) { return a[k] * b[k]; });
    rg.rows(1, [&](int k) { return a0 * b[k] + b0 * a[k]; }, [&](int k, double t) { o[k] = t; });
    if (rg.owns_head()) o[0] = dot;
}
// o = lam \ d; o may alias d
template <class R>
SCVX_HD void cone_div(R rg, cgptr lam, cgptr d, gptr o) {
    if (rg.q == 1) { o[0] = d[0] / lam[0]; return; }
    const double l0 = rg.head(lam), d0 = rg.head(d);
    const double l1d1 = rg.sum1([&](int k) { return lam[k] * d[k]; }), l1l1 = rg.sum1([&](int k) { return lam[k] * lam[k]; });
    const double n1 = sqrt(l1l1);
    const double x0 = (l0 * d0 - l1d1) / (fmax(l0 - n1, 2.3e-16 * l0) * (l0 + n1));   // lam on the boundary to rounding: kept inside
    const double il0 = 1.0 / l0;
    rg.rows(1, [&](int k) { return (d[k] - x0 * lam[k]) * il0; }, [&](int k, double t) { o[k] = t; });
    if (rg.owns_head()) o[0] = x0;
}
// largest alpha with lam + alpha d in the cone (the same value in every lane of a cooperative range)
template <class R>
SCVX_HD double cone_maxstep(R rg, cgptr lam, cgptr d) {
    if (rg.q == 1) return d[0] < 0.0 ? -lam[0] / d[0] : INFINITY;
    const double l0 = rg.head(lam), d0 = rg.head(d);
    const double ll = l0 * l0 - rg.sum1([&](int k) { return lam[k] * lam[k]; });
    const double ld = l0 * d0 - rg.sum1([&](int k) { return lam[k] * d[k]; });
    const double dd = d0 * d0 - rg.sum1([&](int k) { return d[k] * d[k]; });
    return ipm::soc_maxstep_parts(l0, d0, ll, ld, dd);
}
// smallest t with x + t e in the cone, and the cone's part of |x|^2
template <class R>
SCVX_HD double cone_shift(R rg, cgptr x, double& n2) {
    if (rg.q == 1) { n2 = x[0] * x[0]; return -x[0]; }
    const double x0 = rg.head(x);
    const double n1 = rg.sum1([&](int k) { return x[k] * x[k]; });
    n2 = x0 * x0 + n1;
    return sqrt(n1) - x0;
}

// ------------------------------------------------------------------------------------------------------------------
// The 15 cone rows of ONE NODE in registers -- [4 linear | SOC3 rows 4..6 | SOC4 rows 7..10 | SOC4 rows 11..14] -- with
// every index a compile-time constant: a pass loads a node's rows in one round trip, does the whole fused sequence
// (E du, W^-1 W^-1, Jordan products ...) in registers and stores once.  (Walking the node's cones one after the other
// through memory costs a dependent load round per cone and per step of the sequence.)
// ------------------------------------------------------------------------------------------------------------------
struct NodeScal { double v[NR]; double b[3]; };   // NT scaling of a node: v rows (d for the linear rows), beta of the 3 cones

SCVX_HD void nd_load(cgptr p, double (&x)[NR]) { SCVX_UNROLL for (int r = 0; r < NR; r++) x[r] = p[r]; }
SCVX_HD void nd_store(gptr p, const double (&x)[NR]) { SCVX_UNROLL for (int r = 0; r < NR; r++) p[r] = x[r]; }
// rows of E for one node from its 15 variables d (r v ma T ga kaR ar): the row table of build_tables, hard-wired
// (build_tables checks that the two agree)
SCVX_HD void nd_E(const double (&d)[NV], double itg, double cth, double (&e)[NR]) {
    e[0] = d[6]; e[1] = d[10]; e[2] = -d[10]; e[3] = d[7] - cth * d[10];
    e[4] = d[0] * itg; e[5] = d[1]; e[6] = d[2];
    e[7] = d[10]; e[8] = d[7]; e[9] = d[8]; e[10] = d[9];
    e[11] = d[11]; e[12] = d[12]; e[13] = d[13]; e[14] = d[14];
}
template <int O, int Q>
SCVX_HD void rg_W(const double (&v)[NR], double beta, const double (&x)[NR], double (&y)[NR], bool inverse) {
    const double sg = inverse ? -1.0 : 1.0;
    double t = 0;
    SCVX_UNROLL
    for (int k = 1; k < Q; k++) t += v[O + k] * x[O + k];
    const double x0 = x[O];
    const double vx = v[O] * x0 + sg * t;
    const double sc = inverse ? 1.0 / beta : beta;
    const double tw = 2.0 * sg * vx;
    SCVX_UNROLL
    for (int k = 1; k < Q; k++) y[O + k] = (tw * v[O + k] + x[O + k]) * sc;
    y[O] = (2.0 * vx * v[O] - x0) * sc;
}
SCVX_HD void nd_W(const NodeScal& S, const double (&x)[NR], double (&y)[NR], bool inverse) {   // y may be x
    SCVX_UNROLL
    for (int r = 0; r < 4; r++) y[r] = inverse ? x[r] / S.v[r] : x[r] * S.v[r];
    rg_W<4, 3>(S.v, S.b[0], x, y, inverse);
    rg_W<7, 4>(S.v, S.b[1], x, y, inverse);
    rg_W<11, 4>(S.v, S.b[2], x, y, inverse);
}
template <int O, int Q>
SCVX_HD void rg_nt(const double (&s)[NR], const double (&z)[NR], double (&v)[NR], double& beta, double (&lam)[NR]) {
    double s1 = 0, z1 = 0, sz = 0;
    SCVX_UNROLL
    for (int k = 1; k < Q; k++) { s1 += s[O + k] * s[O + k]; z1 += z[O + k] * z[O + k]; sz += s[O + k] * z[O + k]; }
    const double s0 = s[O], z0 = z[O];
    const double ns = sqrt(s1), nz = sqrt(z1);
    const double sj = sqrt(fmax(s0 - ns, 2.3e-16 * s0) * (s0 + ns)), zj = sqrt(fmax(z0 - nz, 2.3e-16 * z0) * (z0 + nz));
    const double isj = 1.0 / sj, izj = 1.0 / zj;
    const double gam = sqrt(0.5 * (1.0 + (s0 * z0 + sz) * isj * izj));
    const double ig = 0.5 / gam;
    const double wb0 = (s0 * isj + z0 * izj) * ig;
    const double den = 1.0 / sqrt(2.0 * (wb0 + 1.0));
    const double v0 = (wb0 + 1.0) * den;
    const double c1 = ig * den;
    beta = sqrt(sj * izj);
    const double vz = v0 * z0 + c1 * (sz * isj - z1 * izj);
    SCVX_UNROLL
    for (int k = 1; k < Q; k++) {
        const double t = (s[O + k] * isj - z[O + k] * izj) * c1;
        v[O + k] = t;
        lam[O + k] = beta * (2.0 * vz * t + z[O + k]);
    }
    v[O] = v0;
    lam[O] = beta * (2.0 * vz * v0 - z0);
}
SCVX_HD void nd_nt(const double (&s)[NR], const double (&z)[NR], NodeScal& S, double (&lam)[NR]) {
    SCVX_UNROLL
    for (int r = 0; r < 4; r++) { S.v[r] = sqrt(s[r] / z[r]); lam[r] = sqrt(s[r] * z[r]); }
    rg_nt<4, 3>(s, z, S.v, S.b[0], lam);
    rg_nt<7, 4>(s, z, S.v, S.b[1], lam);
    rg_nt<11, 4>(s, z, S.v, S.b[2], lam);
}
template <int O, int Q>
SCVX_HD void rg_prod(const double (&a)[NR], const double (&b)[NR], double (&o)[NR]) {   // o may be a or b
    double dot = 0;
    SCVX_UNROLL
    for (int k = 0; k < Q; k++) dot += a[O + k] * b[O + k];
    const double a0 = a[O], b0 = b[O];
    SCVX_UNROLL
    for (int k = 1; k < Q; k++) o[O + k] = a0 * b[O + k] + b0 * a[O + k];
    o[O] = dot;
}
SCVX_HD void nd_prod(const double (&a)[NR], const double (&b)[NR], double (&o)[NR]) {
    SCVX_UNROLL
    for (int r = 0; r < 4; r++) o[r] = a[r] * b[r];
    rg_prod<4, 3>(a, b, o); rg_prod<7, 4>(a, b, o); rg_prod<11, 4>(a, b, o);
}
template <int O, int Q>
SCVX_HD void rg_div(const double (&lam)[NR], const double (&d)[NR], double (&o)[NR]) {   // o may be d
    double l1d1 = 0, l1l1 = 0;
    SCVX_UNROLL
    for (int k = 1; k < Q; k++) { l1d1 += lam[O + k] * d[O + k]; l1l1 += lam[O + k] * lam[O + k]; }
    const double l0 = lam[O];
    const double n1 = sqrt(l1l1);
    const double x0 = (l0 * d[O] - l1d1) / (fmax(l0 - n1, 2.3e-16 * l0) * (l0 + n1));   // lam on the boundary to rounding: kept inside
    const double il0 = 1.0 / l0;
    SCVX_UNROLL
    for (int k = 1; k < Q; k++) o[O + k] = (d[O + k] - x0 * lam[O + k]) * il0;
    o[O] = x0;
}
SCVX_HD void nd_div(const double (&lam)[NR], const double (&d)[NR], double (&o)[NR]) {
    SCVX_UNROLL
    for (int r = 0; r < 4; r++) o[r] = d[r] / lam[r];
    rg_div<4, 3>(lam, d, o); rg_div<7, 4>(lam, d, o); rg_div<11, 4>(lam, d, o);
}
template <int O, int Q>
SCVX_HD double rg_maxstep(const double (&lam)[NR], const double (&d)[NR]) {
    double ll = lam[O] * lam[O], ld = lam[O] * d[O], dd = d[O] * d[O];
    SCVX_UNROLL
    for (int k = 1; k < Q; k++) { ll -= lam[O + k] * lam[O + k]; ld -= lam[O + k] * d[O + k]; dd -= d[O + k] * d[O + k]; }
    return ipm::soc_maxstep_parts(lam[O], d[O], ll, ld, dd);
}
SCVX_HD double nd_maxstep(const double (&lam)[NR], const double (&d)[NR]) {
    double a = INFINITY;
    SCVX_UNROLL
    for (int r = 0; r < 4; r++) { const double ar = d[r] < 0.0 ? -lam[r] / d[r] : INFINITY; a = ar < a ? ar : a; }
    const double a1 = rg_maxstep<4, 3>(lam, d), a2 = rg_maxstep<7, 4>(lam, d), a3 = rg_maxstep<11, 4>(lam, d);
    a = a1 < a ? a1 : a; a = a2 < a ? a2 : a; a = a3 < a ? a3 : a;
    return a;
}
template <int O, int Q>
SCVX_HD double rg_shift(const double (&x)[NR]) {
    double n1 = 0;
    SCVX_UNROLL
    for (int k = 1; k < Q; k++) n1 += x[O + k] * x[O + k];
    return sqrt(n1) - x[O];
}
// smallest t with x + t e in the node's cones; n2 = |x|^2
SCVX_HD double nd_shift(const double (&x)[NR], double& n2) {
    double t = -INFINITY;
    n2 = 0;
    SCVX_UNROLL
    for (int r = 0; r < NR; r++) n2 += x[r] * x[r];
    SCVX_UNROLL
    for (int r = 0; r < 4; r++) t = -x[r] > t ? -x[r] : t;
    const double t1 = rg_shift<4, 3>(x), t2 = rg_shift<7, 4>(x), t3 = rg_shift<11, 4>(x);
    t = t1 > t ? t1 : t; t = t2 > t ? t2 : t; t = t3 > t ? t3 : t;
    return t;
}

// W^-2 entry (a, b) of the cone at rows O..O+Q-1:  beta^-2 (2 wh wh' - J),  wh = (2 v0^2 - 1, -2 v0 v_tail)
template <int O>
SCVX_HD double rg_w2(const double (&v)[NR], double b2, int a, int b) {
    const double v0 = v[O];
    const double wa = a == 0 ? 2.0 * v0 * v0 - 1.0 : -2.0 * v0 * v[O + a];
    const double wb = b == 0 ? 2.0 * v0 * v0 - 1.0 : -2.0 * v0 * v[O + b];
    return b2 * (2.0 * wa * wb - (a == b ? (a == 0 ? 1.0 : -1.0) : 0.0));
}
// The structural non-zeros (lower triangle) of a node's block of H = E' W^-2 E, straight from the scaling in registers:
// r (3x3, glideslope cone), ma, (T, ga) (thrust cone + the three linear rows on ga and T1), (kaR, ar).  The long cone's
// -J part on kaR is added once its beta is known.  The rest of the 15 x 15 block is zero and is never written.
SCVX_HD void nd_H(const NodeScal& S, double itg, double cth, gptr H) {
    const double c3[3] = {itg, 1.0, 1.0};
    const double b20 = 1.0 / (S.b[0] * S.b[0]), b21 = 1.0 / (S.b[1] * S.b[1]), b22 = 1.0 / (S.b[2] * S.b[2]);
    SCVX_UNROLL
    for (int a = 0; a < 3; a++)
        SCVX_UNROLL
        for (int b = 0; b <= a; b++) H[a * NV + b] = c3[a] * c3[b] * rg_w2<4>(S.v, b20, a, b);
    H[6 * NV + 6] = 1.0 / (S.v[0] * S.v[0]);
    const double w1 = 1.0 / (S.v[1] * S.v[1]), w2 = 1.0 / (S.v[2] * S.v[2]), w3 = 1.0 / (S.v[3] * S.v[3]);
    // thrust cone rows 7..10 <-> variables (ga = 10, T = 7, 8, 9)
    SCVX_UNROLL
    for (int a = 0; a < 4; a++)
        SCVX_UNROLL
        for (int b = 0; b <= a; b++) {
            const int va = a == 0 ? 10 : 6 + a, vb = b == 0 ? 10 : 6 + b;
            double val = rg_w2<7>(S.v, b21, a, b);
            if (a == 0 && b == 0) val += w1 + w2 + cth * cth * w3;   // Tmin <= ga, ga <= Tmax, ga cos(thetaMax) <= T1
            if (a == 1 && b == 0) val += -cth * w3;
            if (a == 1 && b == 1) val += w3;
            H[(va > vb ? va : vb) * NV + (va > vb ? vb : va)] = val;
        }
    SCVX_UNROLL
    for (int a = 0; a < 4; a++)
        SCVX_UNROLL
        for (int b = 0; b <= a; b++) H[(11 + a) * NV + 11 + b] = rg_w2<11>(S.v, b22, a, b);
}

// ------------------------------------------------------------------------------------------------------------------
// the solver: Ex provides lane(), nlanes(), sync(), sync_lds(), sum(), min(), fast() (scratch of fast_doubles(N))
// ------------------------------------------------------------------------------------------------------------------
template <class Ex>
struct Solver {
    Ex& ex;
    const Tables& T;
    Layout L;
    gptr w;           // this trajectory's workspace
    lptr win;         // fast scratch: factorisation window, then the solve vector
    lptr xs;
    lptr dorig;       // |diagonal| of each window column as assembled: the scale of its pivot floor
    liptr ptab;       // (a << 8 | b) of the trailing-update entries, 1 <= b <= a <= BW
    cgptr icv;           // initial position and velocity: the b entries of band rows 0..5
    double beta_big, w0_big, qY;
    double xreg = 0.0;   // extra static regularisation of the current factorisation (0 unless a factorisation had to be retried)
    struct { cgiptr a_col, e_v0, e_v1, t_row; cgptr a_val, kc, e_c0, e_c1, e_h, t_coef, q; } tt;   // T's tables, typed
#if defined(SCVX_TD_PROF)
    double prof[16] = {0};   // section cycles (diagnostic builds): tools/prof_threedof.py
#endif

    SCVX_HD Solver(Ex& e, const Tables& t, double* work) : ex(e), T(t), w((gptr)work) {
        L.init(T.N);
        tt.a_col = (cgiptr)T.a_col; tt.e_v0 = (cgiptr)T.e_v0; tt.e_v1 = (cgiptr)T.e_v1; tt.t_row = (cgiptr)T.t_row;
        tt.a_val = (cgptr)T.a_val; tt.kc = (cgptr)T.kc; tt.e_c0 = (cgptr)T.e_c0; tt.e_c1 = (cgptr)T.e_c1; tt.e_h = (cgptr)T.e_h;
        tt.t_coef = (cgptr)T.t_coef; tt.q = (cgptr)T.q;
        win = ex.fast();
        xs = win + (size_t)NSLOT * BS;
        dorig = xs + (((size_t)T.nb + 7) & ~(size_t)7);
        ptab = (liptr)(dorig + NSLOT + 1);
        for (int a = 1 + ex.lane(); a <= BW; a += ex.nlanes())
            for (int b = 1; b <= a; b++) ptab[a * (a - 1) / 2 + b - 1] = (a << 8) | b;
        ex.sync_lds();
    }
    SCVX_HD double bq(int pos) const { return pos < 6 ? icv[pos] : tt.q[pos]; }

    template <class F> SCVX_HD void each(int n, F&& f) { for (int i = ex.lane(); i < n; i += ex.nlanes()) f(i); }
    // rows i = lane, lane + nlanes, ... in blocks of U per lane: all loads of a block (ld, any value type) before its stores
    // (st).  A plain loop serialises the rows -- the store of row i needs its loads back before row i + 1 may issue its own,
    // and a row is a chain of two or three dependent loads (index, then value) -- so a pass took U times a memory latency.
    template <int U, class LD, class ST_> SCVX_HD void each2(int n, LD&& ld, ST_&& st) {
        const int nl = ex.nlanes();
        for (int i0 = ex.lane(); i0 < n; i0 += U * nl) {
            decltype(ld(0)) t[U];
            SCVX_UNROLL
            for (int k = 0; k < U; k++) { const int i = i0 + k * nl; t[k] = ld(i < n ? i : n - 1); }
            SCVX_UNROLL
            for (int k = 0; k < U; k++) { const int i = i0 + k * nl; if (i < n) st(i, t[k]); }
        }
    }
    template <class F> SCVX_HD double sum(int n, F&& f) {
        double a = 0;
        for (int i = ex.lane(); i < n; i += ex.nlanes()) a += f(i);
        return ex.sum(a);
    }
    // f(c, off, rg) for every cone: the small ones lane by lane (Serial), then the long cone by the whole wavefront (Coop)
    template <class F> SCVX_HD void all_cones(F&& f) {
        const int nsmall = T.ncone - 1;
        for (int c = ex.lane(); c < nsmall; c += ex.nlanes()) {
            int off, q;
            cone_of(T.N, c, off, q);
            f(c, off, Serial{q});
        }
        f(nsmall, NR * (T.N + 1), Coop<Ex>{ex, T.N + 2});
    }

    // node-level pieces of a cone pass: f(i) for the nodes (one lane each, rows in registers), g(c, off, rg) for the long cone
    template <int ID, class F, class G> SCVX_HD void cone_pass(F&& f, G&& g) {
#if defined(TD_GENERIC_MASK)
        if ((TD_GENERIC_MASK >> ID) & 1) { all_cones(g); return; }   // diagnostic: the memory-walking form of pass ID
#endif
        for (int i = ex.lane(); i <= T.N; i += ex.nlanes()) f(i);
        g(T.ncone - 1, NR * (T.N + 1), Coop<Ex>{ex, T.N + 2});
    }
    SCVX_HD void scal_load(int i, NodeScal& S) const {
        cgptr wv = w + L.wv + NR * i;
        cgptr wb = w + L.wb + 7 * i + 4;
        SCVX_UNROLL
        for (int r = 0; r < NR; r++) S.v[r] = wv[r];
        S.b[0] = wb[0]; S.b[1] = wb[1]; S.b[2] = wb[2];
    }
    SCVX_HD void vars_load(cgptr v, int i, double (&d)[NV]) const {
        cgptr pv = v + pos_z(i, 0);
        SCVX_UNROLL
        for (int l = 0; l < NV; l++) d[l] = pv[l];
    }
    // (E v)[r] and ([0 A'; A 0] v)[p]
    SCVX_HD double E_row(cgptr v, int r) const {
        double a = tt.e_c0[r] * v[tt.e_v0[r]];
        const int v1 = tt.e_v1[r];
        if (v1 >= 0) a += tt.e_c1[r] * v[v1];
        return a;
    }
    SCVX_HD double A_row(cgptr v, int p) const {
        double a = 0;
        if (p < T.nb) {
            SCVX_UNROLL
            for (int k = 0; k < AW; k++) {
                const int c = tt.a_col[p * AW + k];
                if (c >= 0) a += tt.a_val[p * AW + k] * v[c];
            }
        }
        return a;
    }
    // (E' zz)[p]
    SCVX_HD double Et_at(cgptr zz, int p) const {
        double a = 0;
        for (int k = 0; k < TW; k++) {
            const int r = tt.t_row[p * TW + k];
            if (r >= 0) a += tt.t_coef[p * TW + k] * zz[r];
        }
        return a;
    }
    // ---- NT scaling from (s, z); lam = W z; the per-node blocks of H = E' W^-2 E; border data of the long cone ----
    SCVX_HD_NI void scale(bool identity) {
        TD_TS(t0_);
        gptr wv = w + L.wv;
        gptr wb = w + L.wb;
        gptr hd = w + L.hd;
        if (identity) {
            each(T.m, [&](int r) { wv[r] = 0.0; });
            ex.sync();
            all_cones([&](int c, int off, auto rg) { if (rg.owns_head()) { wv[off] = 1.0; wb[c] = 1.0; } });
            each(T.N + 1, [&](int i) {
                NodeScal S;
                SCVX_UNROLL
                for (int r = 0; r < NR; r++) S.v[r] = (r < 5 || r == 7 || r == 11) ? 1.0 : 0.0;
                S.b[0] = 1.0; S.b[1] = 1.0; S.b[2] = 1.0;
                nd_H(S, T.itg, T.cth, hd + (size_t)i * NV * NV);
            });
        } else {
            cgptr s = w + L.s;
            cgptr z = w + L.z;
            gptr lam = w + L.lam;
            cone_pass<0>([&](int i) {
                double sr[NR], zr[NR], lr[NR];
                NodeScal S;
                nd_load(s + NR * i, sr); nd_load(z + NR * i, zr);
                nd_nt(sr, zr, S, lr);
                nd_store(wv + NR * i, S.v); nd_store(lam + NR * i, lr);
                gptr b7 = wb + 7 * i;
                b7[0] = 1.0; b7[1] = 1.0; b7[2] = 1.0; b7[3] = 1.0; b7[4] = S.b[0]; b7[5] = S.b[1]; b7[6] = S.b[2];
                nd_H(S, T.itg, T.cth, hd + (size_t)i * NV * NV);
            }, [&](int c, int off, auto rg) {
                double beta;
                cone_nt(rg, s + off, z + off, wv + off, beta, lam + off);
                if (rg.owns_head()) wb[c] = beta;
            });
        }
        ex.sync();
        TD_TE(t0_, 0);
        TD_TS(t1_);
        const int bo = NR * (T.N + 1);
        beta_big = wb[T.ncone - 1];
        const double v0 = wv[bo];
        w0_big = 2.0 * v0 * v0 - 1.0;
        const double ib2 = 1.0 / (beta_big * beta_big);
        each(T.N + 1, [&](int i) { hd[(size_t)i * NV * NV + 11 * NV + 11] += ib2; });   // the -J part of the long cone on kaR_i (same lane as nd_H)
        ex.sync();
        TD_TE(t1_, 1);
    }

    // ---- banded LDL' of [H + delta I, A'; A, -delta I] without the border; L and 1/d to HBM ----
    // entry (c + d, c) of the band matrix without the border: constant part + the node block of H.  Branch-free: kc is
    // zero-padded past nb, positions outside a node block read the zero entry behind the H blocks.
    static SCVX_HD double kcol(cgptr __restrict__ kc, cgptr __restrict__ hd, int N, int c, int d, double xreg) {
        const int pt = c - 7, i = pt / NP, l = pt - NP * i;
        const bool blk = pt >= 0 && i <= N && l + d < NV;   // l + d < NV implies l < NV: a variable column
        const int hz = (N + 1) * NV * NV;
        // xreg: extra regularisation of a retried factorisation, +xreg on variable pivots, -xreg on equality pivots
        const bool var = pt >= 0 && i <= N && l < NV;
        return kc[(size_t)c * BS + d] + hd[blk ? i * NV * NV + (l + d) * NV + l : hz] + (d == 0 ? (var ? xreg : -xreg) : 0.0);
    }
    // L by columns (lb[j][d] = L[j+d][j], lb[j][0] = 1/d_j) for the forward sweep and by rows (ut[r][d] = L[r][r-d]) for
    // the backward one, so both stream contiguous memory.  The window holds columns [j0, j0 + NSLOT) during the block of
    // ST steps starting at j0; the ST columns that enter afterwards are fetched into registers at the start of the
    // block, so no step waits for HBM.
    SCVX_HD_NI bool factor() {
        gptr __restrict__ lb = w + L.lb;
        gptr __restrict__ ut = w + L.ut;
        cgptr __restrict__ kc = tt.kc;
        cgptr __restrict__ hd = w + L.hd;
        lptr wn = win;
        lptr dor = dorig;
        const int N = T.N, nb = T.nb;
        const double xr = xreg;
        const int spare = (int)(dorig + NSLOT - win);   // dorig has NSLOT + 1 entries
        const int lane = ex.lane(), nl = ex.nlanes();
        constexpr int PER = (ST * BS + Ex::kLanes - 1) / Ex::kLanes;
        constexpr int PP = (NPAIR + Ex::kLanes - 1) / Ex::kLanes;
        for (int e = lane; e < NSLOT * BS; e += nl) {
            const int c = e / BS, d = e - c * BS;
            const double v = kcol(kc, hd, N, c, d, xr);
            wn[e] = v;
            if (d == 0) dor[c] = fabs(v);
        }
        int pab[PP];
        for (int q = 0; q < PP; q++) { const int p = lane + q * nl; pab[q] = p < NPAIR ? ptab[p] : 0; }
        ex.sync_lds();
        bool ok = true;
        double stage[PER];
        for (int j0 = 0; j0 < nb; j0 += ST) {
            const int cbase = j0 + NSLOT;
            SCVX_UNROLL
            for (int q = 0; q < PER; q++) {
                const int e = lane + q * nl;
                stage[q] = e < ST * BS ? kcol(kc, hd, N, cbase + e / BS, e % BS, xr) : 0.0;
            }
            const int jend = j0 + ST < nb ? j0 + ST : nb;
            for (int j = j0; j < jend; j++) {
                const TD_LOCAL double* col = wn + (j % NSLOT) * BS;
                // every LDS read of the step is requested before anything waits for one (the pivot, its scale, the operands of the
                // trailing update, this lane's entry of the column: one LDS round trip; in the compiler's own order the pivot's
                // division -- nine dependent instructions -- sat between the requests).  Worth 2.5 % of a solve at B = 1 and 1 % at
                // B = 8,192 (profiles/r04_k0_column_step.md): the step is bound by the ~135 instructions a single wavefront issues
                // for it, not by LDS latency.
                double dj = col[0];
                const double dsc = dor[j % NSLOT];
                double ca[PP], cb[PP], cw[PP];
                int ti[PP];
                SCVX_UNROLL
                for (int q = 0; q < PP; q++) {
                    const int ab = pab[q], a = ab >> 8, b = ab & 255;
                    ti[q] = ab ? ((j + b) % NSLOT) * BS + (a - b) : spare;
                    ca[q] = col[a]; cb[q] = col[b]; cw[q] = wn[ti[q]];
                }
                constexpr int PD = (BS + Ex::kLanes - 1) / Ex::kLanes;   // column entries per lane (1 on a wavefront)
                double cd[PD];
                SCVX_UNROLL
                for (int q = 0; q < PD; q++) { const int d = lane + q * nl; cd[q] = d < BS ? col[d] : 0.0; }
                TD_SCHED_FENCE();
                const bool var = is_var(N, j);
                // dynamic regularisation: the pivot keeps the sign quasi-definiteness gives it, and does not fall below
                // the rounding level of the entry it started from (an active cone's block is rank one to working
                // precision in the last iterations); the refinement passes absorb the perturbation
                const double fl = 1e-15 * dsc + 1e-13;
                if (!(fabs(dj) < 1e300)) ok = false;   // NaN or overflow: the caller retries with more regularisation
                if (var ? !(dj > fl) : !(dj < -fl)) dj = var ? fl : -fl;
                const double idj = 1.0 / dj;
                // all reads of the trailing update, then all writes (a lane without a q-th entry updates the spare double
                // behind the pivot scales)
                SCVX_UNROLL
                for (int q = 0; q < PP; q++) wn[ti[q]] = cw[q] - ca[q] * cb[q] * idj;
                SCVX_UNROLL
                for (int q = 0; q < PD; q++) {
                    const int d = lane + q * nl;
                    if (d < BS) {
                        const double lv = cd[q] * idj;
                        lb[(size_t)j * BS + d] = d == 0 ? idj : lv;
                        if (d >= 1 && j + d < nb) ut[(size_t)(j + d) * BS + d] = lv;
                    }
                }
                ex.sync_lds();
            }
            SCVX_UNROLL
            for (int q = 0; q < PER; q++) {
                const int e = lane + q * nl;
                if (e < ST * BS) {
                    const int c = cbase + e / BS, d = e % BS;
                    wn[(c % NSLOT) * BS + d] = stage[q];
                    if (d == 0) dor[c % NSLOT] = fabs(stage[q]);
                }
            }
            ex.sync_lds();
        }
        ex.sync();
        return ok;
    }
    // xs <- K_band^-1 xs
    SCVX_HD_NI void band_solve() {
        cgptr lb = w + L.lb;
        cgptr ut = w + L.ut;
        const int nb = T.nb;
        if constexpr (Ex::kRegisterSweep) {
            ex.band_sweeps(xs, lb, ut, nb);
        } else {
            const int lane = ex.lane(), nl = ex.nlanes();
            for (int j = 0; j < nb; j++) {
                const double xj = xs[j];
                for (int d = 1 + lane; d <= BW; d += nl)
                    if (j + d < nb) xs[j + d] -= lb[(size_t)j * BS + d] * xj;
                ex.sync_lds();
            }
            for (int j = lane; j < nb; j += nl) xs[j] *= lb[(size_t)j * BS];
            ex.sync_lds();
            for (int j = nb - 1; j > 0; j--) {
                const double xj = xs[j];
                for (int d = 1 + lane; d <= BW; d += nl)
                    if (j - d >= 0) xs[j - d] -= ut[(size_t)j * BS + d] * xj;
                ex.sync_lds();
            }
        }
    }
    SCVX_HD_NI bool factor_all() {
        TD_TS(t2_);
        const bool ok = factor();
        TD_TE(t2_, 2);
        TD_TS(t3_);
        // border: Y = K_band^-1 wh_b (wh_b = the long cone's wh tail at the kaR positions), qY = wh_b' Y
        cgptr wv = w + L.wv;
        const int bo = NR * (T.N + 1);
        const double v0 = wv[bo];
        each(T.nb, [&](int p) { xs[p] = 0.0; });
        ex.sync_lds();
        each(T.N + 1, [&](int i) { xs[pos_z(i, 11)] = -2.0 * v0 * wv[bo + 1 + i]; });
        ex.sync_lds();
        band_solve();
        gptr y = w + L.y;
        each(T.nb, [&](int p) { y[p] = xs[p]; });
        qY = sum(T.N + 1, [&](int i) { return -2.0 * v0 * wv[bo + 1 + i] * xs[pos_z(i, 11)]; });
        ex.sync();
        TD_TE(t3_, 3);
        return ok;
    }

    // ---- one condensed solve: [H A'; A 0][du] = [bu - E' W^-2 bz ; bu_eq],  dz = -W^-2 (E du + bz) ----
    SCVX_HD_NI void condensed(cgptr bu, cgptr bz, gptr du, gptr dz) {
        gptr t1 = w + L.t1;
        cgptr wv = w + L.wv;
        const int bo = NR * (T.N + 1);
        const double v0 = wv[bo];
        TD_TS(t4_);
        cgptr wb = w + L.wb;
        // t1 = W^-2 bz: both applications by the cone's own lane, no pass boundary in between
        cone_pass<1>([&](int i) {
            NodeScal S;
            double x[NR];
            scal_load(i, S); nd_load(bz + NR * i, x);
            nd_W(S, x, x, true); nd_W(S, x, x, true);
            nd_store(t1 + NR * i, x);
        }, [&](int c, int off, auto rg) {
            cone_W(rg, wv + off, wb[c], bz + off, t1 + off, true);
            cone_W(rg, wv + off, wb[c], t1 + off, t1 + off, true);
        });
        ex.sync();
        each2<4>(T.nb, [&](int p) { return bu[p] - Et_at(t1, p); }, [&](int p, double v) { xs[p] = v; });
        const double rnu = bu[T.nb] - Et_at(t1, T.nb);
        ex.sync_lds();
        TD_TE(t4_, 4);
        TD_TS(t5_);
        band_solve();
        TD_TE(t5_, 5);
        TD_TS(t6_);
        const double wk = sum(T.N + 1, [&](int i) { return -2.0 * v0 * wv[bo + 1 + i] * xs[pos_z(i, 11)]; });
        // [-beta^-2 + delta, w0; w0, -(beta^2/2 + qY)] [nu; t] = [rnu; -wk]
        const double b2 = beta_big * beta_big;
        const double m00 = -1.0 / b2 + T.delta, m01 = w0_big, m11 = -(0.5 * b2 + qY);
        const double det = m00 * m11 - m01 * m01;
        const double nu = (rnu * m11 + m01 * wk) / det;
        const double tt = (-m00 * wk - m01 * rnu) / det;
        cgptr y = w + L.y;
        each2<4>(T.nb, [&](int p) { return xs[p] - y[p] * tt; }, [&](int p, double v) { du[p] = v; });
        if (ex.lane() == 0) du[T.nb] = nu;
        ex.sync();
        // dz = -W^-2 (E du + bz), cone by cone
        cone_pass<2>([&](int i) {
            NodeScal S;
            double d[NV], e[NR], x[NR];
            scal_load(i, S); vars_load(du, i, d); nd_load(bz + NR * i, x);
            nd_E(d, T.itg, T.cth, e);
            SCVX_UNROLL
            for (int r = 0; r < NR; r++) x[r] = -(e[r] + x[r]);
            nd_W(S, x, x, true); nd_W(S, x, x, true);
            nd_store(dz + NR * i, x);
        }, [&](int c, int off, auto rg) {
            rg.rows(0, [&](int k) { return -(E_row(du, off + k) + bz[off + k]); }, [&](int k, double t) { dz[off + k] = t; });
            cone_W(rg, wv + off, wb[c], dz + off, dz + off, true);
            cone_W(rg, wv + off, wb[c], dz + off, dz + off, true);
        });
        ex.sync();
        TD_TE(t6_, 6);
    }
    // Newton solve with T.refine passes on the uncondensed residual; bu, bz are overwritten with the last residual
    SCVX_HD_NI void kkt_solve(gptr bu, gptr bz, gptr du, gptr dz, int passes) {
        condensed(bu, bz, du, dz);
        gptr r1 = w + L.tu;
        gptr r3 = w + L.t2;
        gptr ddu = w + L.tu2;
        gptr ddz = w + L.t3;
        // ADAPTIVE: `passes` are always done; while the last correction was large against the direction (|ddz| > 1e-2 |dz|: the
        // condensed solve was badly off -- on some instances W^-2 is so ill-conditioned in the last iterations that the first
        // correction exceeds the direction and the passes contract by only 0.3 each) further passes follow, up to TD_REFINE_MAX.
        // Typical iterations correct by 1e-6 or less and add nothing.
        for (int pass = 0, want = passes; pass < want; pass++) {
            // r1 = bu - ([0 A'; A 0] du - E' dz),  r3 = bz + E du + W^2 dz
            TD_TS(t7_);
            each2<4>(T.nb + 1, [&](int p) { return bu[p] - A_row(du, p) + Et_at(dz, p); }, [&](int p, double v) { r1[p] = v; });
            {
                cgptr wv = w + L.wv;
                cgptr wb = w + L.wb;
                cone_pass<3>([&](int i) {
                    NodeScal S;
                    double d[NV], e[NR], x[NR], b[NR];
                    scal_load(i, S); vars_load(du, i, d); nd_load(dz + NR * i, x); nd_load(bz + NR * i, b);
                    nd_E(d, T.itg, T.cth, e);
                    nd_W(S, x, x, false); nd_W(S, x, x, false);
                    SCVX_UNROLL
                    for (int r = 0; r < NR; r++) x[r] = b[r] + e[r] + x[r];
                    nd_store(r3 + NR * i, x);
                }, [&](int c, int off, auto rg) {
                    cone_W(rg, wv + off, wb[c], dz + off, r3 + off, false);
                    cone_W(rg, wv + off, wb[c], r3 + off, r3 + off, false);
                    rg.rows(0, [&](int k) { return bz[off + k] + E_row(du, off + k) + r3[off + k]; },
                            [&](int k, double t) { r3[off + k] = t; });
                });
            }
            ex.sync();
            TD_TE(t7_, 7);
            condensed(r1, r3, ddu, ddz);
            each2<4>(T.nb + 1, [&](int p) { return du[p] + ddu[p]; }, [&](int p, double v) { du[p] = v; });
            double cmax = 0, dmax = 0;
            each2<4>(T.m, [&](int r) { return dz[r] + ddz[r]; }, [&](int r, double v) {
                dz[r] = v;
                const double c = fabs(ddz[r]), d = fabs(v);
                cmax = c > cmax ? c : cmax; dmax = d > dmax ? d : dmax;
            });
            cmax = -ex.min(-cmax); dmax = -ex.min(-dmax);
            ex.sync();
            if (pass + 1 == want && want < TD_REFINE_MAX && passes > 0 && cmax > TD_REFINE_MORE * dmax) want++;
        }
    }

    // shift x into the interior of the cone if it is not (CVXOPT initialisation)
    SCVX_HD_NI void shift_in(gptr x) {
        double t = -INFINITY, n2 = 0;
        cone_pass<4>([&](int i) {
            double xr[NR], c2;
            nd_load(x + NR * i, xr);
            const double tc = nd_shift(xr, c2);
            t = tc > t ? tc : t; n2 += c2;
        }, [&](int, int off, auto rg) {
            double c2;
            const double tc = cone_shift(rg, x + off, c2);
            if (rg.owns_head()) { t = tc > t ? tc : t; n2 += c2; }
        });
        t = -ex.min(-t);
        n2 = ex.sum(n2);
        const double nrm = sqrt(n2);
        if (t >= -1e-8 * (nrm > 1.0 ? nrm : 1.0)) {
            all_cones([&](int, int off, auto rg) { if (rg.owns_head()) x[off] += 1.0 + t; });
        }
        ex.sync();
    }
    // ds = -rz + E du; t1 = W^-1 ds, t2 = W dz; the largest step keeping lam + alpha t1 and lam + alpha t2 in the cone --
    // one pass, every cone by its own lane
    SCVX_HD_NI double step_pass(cgptr du, cgptr dz) {
        cgptr wv = w + L.wv;
        cgptr wb = w + L.wb;
        cgptr lam = w + L.lam;
        cgptr rz = w + L.rz;
        gptr ds = w + L.ds, t1 = w + L.t1, t2 = w + L.t2;
        double a = INFINITY;
        cone_pass<5>([&](int i) {
            NodeScal S;
            double d[NV], e[NR], x[NR], y[NR], lr[NR];
            scal_load(i, S); vars_load(du, i, d); nd_load(rz + NR * i, x); nd_load(dz + NR * i, y); nd_load(lam + NR * i, lr);
            nd_E(d, T.itg, T.cth, e);
            SCVX_UNROLL
            for (int r = 0; r < NR; r++) x[r] = -x[r] + e[r];
            nd_store(ds + NR * i, x);
            nd_W(S, x, x, true); nd_W(S, y, y, false);
            nd_store(t1 + NR * i, x); nd_store(t2 + NR * i, y);
            const double a1 = nd_maxstep(lr, x), a2 = nd_maxstep(lr, y);
            const double am = a1 < a2 ? a1 : a2;
            a = am < a ? am : a;
            if (!(am == am)) a = -1.0;   // non-finite: a comparison would drop it
        }, [&](int c, int off, auto rg) {
            rg.rows(0, [&](int k) { return -rz[off + k] + E_row(du, off + k); }, [&](int k, double t) { ds[off + k] = t; });
            cone_W(rg, wv + off, wb[c], ds + off, t1 + off, true);
            cone_W(rg, wv + off, wb[c], dz + off, t2 + off, false);
            const double a1 = cone_maxstep(rg, lam + off, t1 + off), a2 = cone_maxstep(rg, lam + off, t2 + off);
            const double am = a1 < a2 ? a1 : a2;
            a = am < a ? am : a;
            if (!(am == am)) a = -1.0;   // non-finite: a comparison would drop it
        });
        a = ex.min(a);
        ex.sync();
        return a >= 0.0 ? a : 0.0;   // a non-finite direction takes no step: the caller stops
    }

    // (kept out of solve(), like every pass that holds a node's rows in registers: inlined into the kernel body next to the
    // other passes, hipcc 7.2 produced a kernel that returned NaN after the first iteration -- either pass alone was fine)
    // combined direction's right-hand side: ds_rhs = -lam o lam - (W^-1 ds_a) o (W dz_a) + sigma mu e;  bz = -rz - W (lam \ ds_rhs)
    SCVX_HD_NI void corr_rhs(double sm) {
        cgptr wv_ = w + L.wv;
        cgptr wb_ = w + L.wb;
        cgptr lam = w + L.lam;
        cgptr rz = w + L.rz;
        gptr t1 = w + L.t1, t2 = w + L.t2, bz = w + L.bz;
        cone_pass<6>([&](int i) {
            NodeScal S;
            double x[NR], y[NR], lr[NR], rr[NR];
            scal_load(i, S); nd_load(t1 + NR * i, x); nd_load(t2 + NR * i, y); nd_load(lam + NR * i, lr); nd_load(rz + NR * i, rr);
            nd_prod(x, y, x);
            nd_prod(lr, lr, y);
            SCVX_UNROLL
            for (int r = 0; r < NR; r++) x[r] = -y[r] - x[r] + ((r < 4 || r == 4 || r == 7 || r == 11) ? sm : 0.0);
            nd_div(lr, x, x);
            nd_W(S, x, x, false);
            SCVX_UNROLL
            for (int r = 0; r < NR; r++) x[r] = -rr[r] - x[r];
            nd_store(bz + NR * i, x);
        }, [&](int c, int off, auto rg) {
            cone_prod(rg, t1 + off, t2 + off, t1 + off);
            cone_prod(rg, lam + off, lam + off, t2 + off);
            rg.rows(0, [&](int k) { return -t2[off + k] - t1[off + k] + (k == 0 ? sm : 0.0); }, [&](int k, double t) { t1[off + k] = t; });
            cone_div(rg, lam + off, t1 + off, t1 + off);
            cone_W(rg, wv_ + off, wb_[c], t1 + off, t1 + off, false);
            rg.rows(0, [&](int k) { return -rz[off + k] - t1[off + k]; }, [&](int k, double t) { bz[off + k] = t; });
        });
    }

    // (the passes of solve()'s loop are routines of their own, like every pass: inlined into the kernel body their loop-invariant
    // addresses were hoisted out of the iteration loop and spilled)
    SCVX_HD_NI void residuals(double& pobj, double& dobj, double& nx, double& ny, double& gap, double& nz) {
        gptr u = w + L.u, s = w + L.s, z = w + L.z, ru = w + L.ru, rz = w + L.rz;
        const int nb = T.nb, m = T.m, N = T.N;
        pobj = 0; dobj = 0; nx = 0; ny = 0;
        struct Two { double r, o, w; };
        each2<4>(nb + 1, [&](int p) {
            const bool var = p == nb || is_var(N, p);
            const double qp = var ? tt.q[p] : bq(p);
            const double a = A_row(u, p);
            Two t;
            t.r = var ? qp + a - Et_at(z, p) : a - qp;
            t.o = qp * u[p];
            t.w = 0.0;
            return t;
        }, [&](int p, Two t) {
            ru[p] = t.r;
            if (p == nb || is_var(N, p)) { nx += t.r * t.r; pobj += t.o; }
            else { ny += t.r * t.r; dobj -= t.o; }
        });
        gap = 0; nz = 0;
        each2<4>(m, [&](int r) {
            Two t;
            t.w = s[r];
            t.r = t.w - (E_row(u, r) + tt.e_h[r]);
            t.o = z[r];
            return t;
        }, [&](int r, Two t) {
            rz[r] = t.r; nz += t.r * t.r; gap += t.w * t.o; dobj -= tt.e_h[r] * t.o;
        });
        pobj = ex.sum(pobj); dobj = ex.sum(dobj); nx = ex.sum(nx); ny = ex.sum(ny); nz = ex.sum(nz); gap = ex.sum(gap);
        ex.sync();
    }
    SCVX_HD_NI void pred_rhs() {
        cgptr ru = w + L.ru, rz = w + L.rz, s = w + L.s;
        gptr bu = w + L.bu, bz = w + L.bz;
        const int nb = T.nb, m = T.m;
        each2<4>(nb + 1, [&](int p) { return -ru[p]; }, [&](int p, double v) { bu[p] = v; });
        each2<4>(m, [&](int r) { return -rz[r] + s[r]; }, [&](int r, double v) { bz[r] = v; });
        ex.sync();
    }
    SCVX_HD_NI void corr_bu() {
        cgptr ru = w + L.ru;
        gptr bu = w + L.bu;
        each2<4>(T.nb + 1, [&](int p) { return -ru[p]; }, [&](int p, double v) { bu[p] = v; });
        ex.sync();
    }
    SCVX_HD_NI void take_step(double alpha) {
        gptr u = w + L.u, s = w + L.s, z = w + L.z;
        cgptr du = w + L.du, dz = w + L.dz, ds = w + L.ds;
        const int nb = T.nb, m = T.m;
        each2<4>(nb + 1, [&](int p) { return u[p] + alpha * du[p]; }, [&](int p, double v) { u[p] = v; });
        each2<4>(m, [&](int r) { return z[r] + alpha * dz[r]; }, [&](int r, double v) { z[r] = v; });
        each2<4>(m, [&](int r) { return s[r] + alpha * ds[r]; }, [&](int r, double v) { s[r] = v; });
        ex.sync();
    }

    SCVX_HD Result solve(const double* ic_, double* out_) {
        cgptr ic = (cgptr)ic_;
        gptr out = (gptr)out_;
        icv = ic;
        gptr u = w + L.u, s = w + L.s, z = w + L.z;
        gptr du = w + L.du, dz = w + L.dz;
        gptr bu = w + L.bu, bz = w + L.bz;
        const int nb = T.nb, m = T.m, N = T.N;
        TD_TS(tt_);
        Result R;
        R.status = TD_ITER_CAP; R.iters = 0; R.pobj = 0; R.gap = INFINITY; R.pres = INFINITY; R.dres = INFINITY;
        double b2 = T.b2_rest;
        for (int i = 0; i < 6; i++) b2 += icv[i] * icv[i];
        const double nrm_b = sqrt(b2) > 1.0 ? sqrt(b2) : 1.0;

        // the zero entry behind the H blocks and the padding around L (read, masked, by the window fetch and the sweeps)
        each((N + 1) * NV * NV + 8, [&](int k) { (w + L.hd)[k] = 0.0; });   // the H blocks' structural zeros, and the zero entry behind them
        each(LPAD, [&](int k) {
            (w + L.lb)[-1 - k] = 0.0; (w + L.lb)[(size_t)nb * BS + k] = 0.0;
            (w + L.ut)[-1 - k] = 0.0; (w + L.ut)[(size_t)nb * BS + k] = 0.0;
        });
        // the corner of the row copy that no factorisation writes (row r, d > r: L[r][r - d] does not exist): the register sweeps
        // read it unmasked
        each(BS * BS, [&](int k) { const int r = k / BS, d = k - BS * r; if (d > r && r < nb) (w + L.ut)[(size_t)r * BS + d] = 0.0; });
        ex.sync();
        // ---- initial point: W = I,  [0 A' G'; A 0 0; G 0 -I][x; y; z] = [-c; b; h],  s = -z, shifted into the cone ----
        scale(true);
        bool ok = factor_all();
        each(nb + 1, [&](int p) { bu[p] = (p == nb || is_var(N, p)) ? -tt.q[p] : bq(p); });
        each(m, [&](int r) { bz[r] = tt.e_h[r]; });
        ex.sync();
        kkt_solve(bu, bz, u, z, T.refine);
        each(m, [&](int r) { s[r] = -z[r]; });
        ex.sync();
        shift_in(s);
        shift_in(z);

        const int degree = T.ncone;
        double best_pres = INFINITY;
        int flat = 0;
        bool near = false, almost = false;
        int near_run = 0;
        for (int it = 1; it <= T.max_iter; it++) {
            R.iters = it;
            TD_TS(t8_);
            // residuals: ru = [c + A'y - E'z at variables; A x - b at equalities], rz = s - e(x)
            double pobj, dobj, nx, ny, gap, nz;
            residuals(pobj, dobj, nx, ny, gap, nz);
            const double pres = fmax(sqrt(ny) / nrm_b, sqrt(nz) / T.nrm_h), dres = sqrt(nx) / T.nrm_c;
            const double relgap = gap / fmax(1.0, fmax(fabs(pobj), fabs(dobj)));
            R.pobj = pobj; R.gap = gap; R.pres = pres; R.dres = dres;
            TD_TE(t8_, 8);
            SCVX_DBG("td %3d pobj %+.10e dobj %+.10e gap %.2e pres %.2e dres %.2e\n", it, pobj, dobj, gap, pres, dres);
            if (!(pres == pres) || !(dres == dres) || !(gap == gap) || !ok) { R.status = TD_NONFINITE; break; }
            near = pres < 10.0 * T.tol && dres < 10.0 * T.tol && relgap < 100.0 * T.tol;
            almost = fmax(pres, fmax(dres, relgap)) < 1e-6;   // a breakdown here is reported as ALMOST optimal (K4's status 4), not as a failure
            if (pres < T.tol && dres < T.tol && (gap < T.tol || relgap < T.tol)) { R.status = TD_OPTIMAL; break; }
            // the numerical floor without a breakdown: on some instances the dual residual sits at a few 1e-9 (the z update
            // loses digits to W^-2) while the gap keeps closing; three consecutive iterates inside the near-optimal band
            // (residuals < 10 tol, relative gap < 100 tol -- what the oracle's solver accepts at a breakdown) end the solve
            near_run = near ? near_run + 1 : 0;
            if (near_run >= 3) { R.status = TD_OPTIMAL; break; }
            // primal infeasibility shows as a primal residual that stops falling while complementarity and the dual
            // residual converge (the multipliers run off along a Farkas ray, dobj grows without bound)
            if (pres < 0.9 * best_pres) { best_pres = pres; flat = 0; } else flat++;
            if (flat >= 5 && dres < 1e-6 && relgap < 1e-6) { R.status = TD_INFEASIBLE; break; }
            if (it == T.max_iter) break;

            scale(false);
            ok = factor_all();
            // a non-finite pivot (the matrix lost quasi-definiteness to rounding): retry with 1e-7, 1e-5 on the diagonal --
            // the refinement passes work against the unregularised system, so the direction stays a Newton direction
            for (int retry = 0; !ok && retry < 2; retry++) {
                xreg = retry == 0 ? 1e-7 : 1e-5;
                ok = factor_all();
            }
            const bool retried = xreg != 0.0;
            xreg = 0.0;
            if (!ok) { R.status = near ? TD_OPTIMAL : (almost ? TD_ALMOST : TD_NONFINITE); break; }
            const double mu = gap / degree;

            // predictor: ds_rhs = -lam o lam, so W (lam \ ds_rhs) = -W lam = -s:  bz = -rz + s
            pred_rhs();
            kkt_solve(bu, bz, du, dz, TD_PRED_REFINE);
            TD_TS(t9_);
            double alpha = fmin(1.0, step_pass(du, dz));
            const double sigma = (1.0 - alpha) * (1.0 - alpha) * (1.0 - alpha);
            // combined: ds_rhs = -lam o lam - (W^-1 ds_a) o (W dz_a) + sigma mu e;  bz = -rz - W (lam \ ds_rhs)
            corr_rhs(sigma * mu);
            corr_bu();
            TD_TE(t9_, 9);
            kkt_solve(bu, bz, du, dz, (retried || fmax(pres, relgap) < TD_REFINE_FROM) ? (T.refine > 1 ? T.refine : 1) + (retried ? 1 : 0) : 0);
            TD_TS(t10_);
            alpha = fmin(1.0, 0.99 * step_pass(du, dz));
            // the numerical floor: an iterate that is a certified near-optimum (the band the oracle's solver, oracle/ipm.py,
            // and Mosek / ECOS at their default tolerances report as OPTIMAL) is accepted when the KKT system breaks down
            if (!(alpha >= 1e-8)) { R.status = near ? TD_OPTIMAL : (almost ? TD_ALMOST : (alpha == alpha ? TD_STALLED : TD_NONFINITE)); break; }
            take_step(alpha);
            TD_TE(t10_, 10);
        }
        // the variables, node by node, then nkaR
        each((N + 1) * NV, [&](int k) { out[k] = u[pos_z(k / NV, k % NV)]; });
        if (ex.lane() == 0) out[(N + 1) * NV] = u[nb];
        ex.sync();
        TD_TE(tt_, 15);
        return R;
    }
};

// ------------------------------------------------------------------------------------------------------------------
// host: the constant tables of one DescentProblem
// ------------------------------------------------------------------------------------------------------------------
struct Problem3 {   // the DescentProblem fields solve_initial reads (initial_solve.jl:19-45)
    int K;
    double alpha, tf_guess, mwet, mdry, g, Tmin, Tmax, thetaMax, gammaGs;
};
struct HostTables {
    std::vector<int> a_col, e_v0, e_v1, t_row;
    std::vector<double> a_val, kc, e_c0, e_c1, e_h, t_coef, q;
    Tables t;
};
// returns an empty string, or why the tables cannot be built
inline const char* build_tables(const Problem3& P, double tol, int max_iter, int refine, double delta, HostTables& H) {
    const int N = P.K;
    if (N < 1) return "K >= 1 required";
    const int nb = band_size(N), m = cone_rows(N);
    const double dt = P.tf_guess / N;                                    // initial_solve.jl:23
    const double d2r = M_PI / 180.0;
    const double tggs = std::tan(P.gammaGs * d2r), cth = std::cos(P.thetaMax * d2r);   // :41-42
    const double wkar = 100.0;                                           // :39
    std::vector<double> mu(N + 1);
    for (int k = 0; k <= N; k++) mu[k] = ((double)(N - k) / N) * P.mwet + ((double)k / N) * P.mdry;   // :24
    H.a_col.assign((size_t)nb * AW, -1); H.a_val.assign((size_t)nb * AW, 0.0);
    H.kc.assign((size_t)nb * BS + 2 * LPAD, 0.0);   // zero past nb: the window fetch runs NSLOT + ST columns ahead
    H.q.assign((size_t)nb + 1, 0.0);
    const char* err = nullptr;
    auto put = [&](int row, int col, double v) {   // A[row, col] = v, both triangles
        for (int pass = 0; pass < 2; pass++) {
            const int r = pass ? col : row, c = pass ? row : col;
            int k = 0;
            while (k < AW && H.a_col[(size_t)r * AW + k] >= 0) k++;
            if (k == AW) { err = "equality row wider than the ELL width"; return; }
            H.a_col[(size_t)r * AW + k] = c; H.a_val[(size_t)r * AW + k] = v;
        }
        const int lo = row < col ? row : col, hi = row < col ? col : row;
        if (hi - lo > BW) { err = "entry outside the band"; return; }
        H.kc[(size_t)lo * BS + (hi - lo)] = v;
    };
    // boundary rows (:59-65); the six initial r, v values are per trajectory
    for (int j = 0; j < 7; j++) put(j, pos_z(0, j), 1.0);
    H.q[6] = P.mwet;
    for (int j = 0; j < 6; j++) put(pos_fin(N, j), pos_z(N, j), 1.0);
    put(pos_fin(N, 6), pos_z(N, 8), 1.0);
    put(pos_fin(N, 7), pos_z(N, 9), 1.0);
    const double gv[3] = {-P.g, 0.0, 0.0};
    for (int i = 0; i < N; i++) {
        for (int j = 0; j < 3; j++) {
            // r_{i+1} - r_i - v_i dt - dt^2/3 (T_i/mu_i + ar_i) - dt^2/6 (T_{i+1}/mu_{i+1} + ar_{i+1}) = dt^2/2 g   (:74-77)
            const int rr = pos_dyn(i, j);
            put(rr, pos_z(i + 1, j), 1.0); put(rr, pos_z(i, j), -1.0); put(rr, pos_z(i, 3 + j), -dt);
            put(rr, pos_z(i, 7 + j), -dt * dt / 3 / mu[i]); put(rr, pos_z(i, 12 + j), -dt * dt / 3);
            put(rr, pos_z(i + 1, 7 + j), -dt * dt / 6 / mu[i + 1]); put(rr, pos_z(i + 1, 12 + j), -dt * dt / 6);
            H.q[rr] = dt * dt / 2 * gv[j];
            // v_{i+1} - v_i - dt/2 (T_i/mu_i + ar_i + T_{i+1}/mu_{i+1} + ar_{i+1}) = dt g                              (:78)
            const int rv = pos_dyn(i, 3 + j);
            put(rv, pos_z(i + 1, 3 + j), 1.0); put(rv, pos_z(i, 3 + j), -1.0);
            put(rv, pos_z(i, 7 + j), -dt / 2 / mu[i]); put(rv, pos_z(i, 12 + j), -dt / 2);
            put(rv, pos_z(i + 1, 7 + j), -dt / 2 / mu[i + 1]); put(rv, pos_z(i + 1, 12 + j), -dt / 2);
            H.q[rv] = dt * gv[j];
        }
        // ma_{i+1} = ma_i - alpha (ga_i + ga_{i+1}) dt/2                                                                (:73)
        const int rm = pos_dyn(i, 6);
        put(rm, pos_z(i + 1, 6), 1.0); put(rm, pos_z(i, 6), -1.0);
        put(rm, pos_z(i, 10), P.alpha * dt / 2); put(rm, pos_z(i + 1, 10), P.alpha * dt / 2);
    }
    if (err) return err;
    for (int p = 0; p < nb; p++) H.kc[(size_t)p * BS] = is_var(N, p) ? delta : -delta;
    // objective (:68)
    H.q[pos_z(N, 6)] = -1.0;
    H.q[nb] = wkar;
    // cone rows (:69-70, :80-88)
    H.e_v0.assign(m, -1); H.e_v1.assign(m, -1); H.e_c0.assign(m, 0.0); H.e_c1.assign(m, 0.0); H.e_h.assign(m, 0.0);
    auto row = [&](int r, int v0, double c0, int v1, double c1, double h) {
        H.e_v0[r] = v0; H.e_c0[r] = c0; H.e_v1[r] = v1; H.e_c1[r] = c1; H.e_h[r] = h;
    };
    for (int i = 0; i <= N; i++) {
        const int r0 = NR * i;
        row(r0 + 0, pos_z(i, 6), 1.0, -1, 0.0, -P.mdry);            // mdry <= ma
        row(r0 + 1, pos_z(i, 10), 1.0, -1, 0.0, -P.Tmin);           // Tmin <= ga
        row(r0 + 2, pos_z(i, 10), -1.0, -1, 0.0, P.Tmax);           // ga <= Tmax
        row(r0 + 3, pos_z(i, 7), 1.0, pos_z(i, 10), -cth, 0.0);     // ga cos(thetaMax) <= T1
        row(r0 + 4, pos_z(i, 0), 1.0 / tggs, -1, 0.0, 0.0);         // [r1/tan(gs); r2; r3] in SOC3
        row(r0 + 5, pos_z(i, 1), 1.0, -1, 0.0, 0.0);
        row(r0 + 6, pos_z(i, 2), 1.0, -1, 0.0, 0.0);
        row(r0 + 7, pos_z(i, 10), 1.0, -1, 0.0, 0.0);               // [ga; T] in SOC4
        for (int j = 0; j < 3; j++) row(r0 + 8 + j, pos_z(i, 7 + j), 1.0, -1, 0.0, 0.0);
        row(r0 + 11, pos_z(i, 11), 1.0, -1, 0.0, 0.0);              // [kaR; ar] in SOC4
        for (int j = 0; j < 3; j++) row(r0 + 12 + j, pos_z(i, 12 + j), 1.0, -1, 0.0, 0.0);
    }
    const int bo = NR * (N + 1);
    row(bo, nb, 1.0, -1, 0.0, 0.0);                                 // [nkaR; kaR] in SOC(N+2)
    for (int i = 0; i <= N; i++) row(bo + 1 + i, pos_z(i, 11), 1.0, -1, 0.0, 0.0);
    H.t_row.assign((size_t)(nb + 1) * TW, -1); H.t_coef.assign((size_t)(nb + 1) * TW, 0.0);
    for (int r = 0; r < m; r++)
        for (int e = 0; e < 2; e++) {
            const int v = e == 0 ? H.e_v0[r] : H.e_v1[r];
            if (v < 0) continue;
            int k = 0;
            while (k < TW && H.t_row[(size_t)v * TW + k] >= 0) k++;
            if (k == TW) return "variable in more cone rows than the ELL width";
            H.t_row[(size_t)v * TW + k] = r; H.t_coef[(size_t)v * TW + k] = e == 0 ? H.e_c0[r] : H.e_c1[r];
        }
    Tables& t = H.t;
    t.N = N; t.nb = nb; t.m = m; t.ncone = cone_count(N);
    t.max_iter = max_iter; t.refine = refine; t.tol = tol; t.delta = delta;
    double c2 = 0, h2 = 0, bb = 0;
    for (int p = 0; p <= nb; p++) {
        if (p == nb || is_var(N, p)) c2 += H.q[p] * H.q[p];
        else if (p >= 6) bb += H.q[p] * H.q[p];
    }
    for (int r = 0; r < m; r++) h2 += H.e_h[r] * H.e_h[r];
    t.nrm_c = std::sqrt(c2) > 1.0 ? std::sqrt(c2) : 1.0;
    t.nrm_h = std::sqrt(h2) > 1.0 ? std::sqrt(h2) : 1.0;
    t.b2_rest = bb;
    t.mwet = P.mwet;
    t.itg = 1.0 / tggs; t.cth = cth;
    // the hard-wired node rows of nd_E must be the row table
    for (int i = 0; i <= N; i++) {
        double d[NV], e[NR];
        for (int probe = 0; probe < NV; probe++) {
            for (int l = 0; l < NV; l++) d[l] = l == probe ? 1.0 : 0.0;
            nd_E(d, t.itg, t.cth, e);
            for (int r = 0; r < NR; r++) {
                const int row = NR * i + r;
                double ref = 0;
                if (H.e_v0[row] == pos_z(i, probe)) ref += H.e_c0[row];
                if (H.e_v1[row] == pos_z(i, probe)) ref += H.e_c1[row];
                if (ref != e[r]) return "node row map out of step with the row table";
            }
        }
    }
    t.a_col = H.a_col.data(); t.a_val = H.a_val.data(); t.kc = H.kc.data();
    t.e_v0 = H.e_v0.data(); t.e_v1 = H.e_v1.data(); t.e_c0 = H.e_c0.data(); t.e_c1 = H.e_c1.data(); t.e_h = H.e_h.data();
    t.t_row = H.t_row.data(); t.t_coef = H.t_coef.data(); t.q = H.q.data();
    return nullptr;
}

}  // namespace td
}  // namespace scvx
