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

namespace scvx {
namespace td {

// Address spaces (device build): the solver's vectors, L and the tables are HBM (global_load/store, vmcnt only), the
// factorisation window and the solve vector are LDS (ds_read/write, lgkmcnt only).  A plain `double*` kept in the solver
// object loses its address space as soon as a routine is not inlined and every access becomes a FLAT one, which counts
// against both counters -- each LDS step of the factorisation then also waits for the L stores in flight.
#if defined(__HIP_DEVICE_COMPILE__)
#define TD_LOCAL __attribute__((address_space(3)))
#else
#define TD_LOCAL
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
constexpr int ST = 32;       // columns staged ahead of the factorisation window (fetched a block early, in registers)
constexpr int LPAD = 1024;   // zero padding (doubles) before and after L and the constant band: the sweeps and the window fetch
                             // load unconditionally a little outside [0, nb) and mask, instead of branching around each load
constexpr int NSLOT = BS + ST; // LDS window of the factorisation: the BS live columns + the ST that become live during a block

// status codes of one solve (the K4 solver's, scvx.h)
enum { TD_OPTIMAL = 0, TD_ITER_CAP = 1, TD_STALLED = 2, TD_NONFINITE = 3, TD_INFEASIBLE = 5 };

struct Tables {
    int N, nb, m, ncone;      // nodes - 1, band size (nkaR sits at position nb), cone rows, cones
    int max_iter, refine;
    double tol, delta;
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
// small-cone arithmetic on memory-resident rows (dim 1 = linear); formulas of oracle/ipm.py::Cone
// ------------------------------------------------------------------------------------------------------------------
SCVX_HD double jdot(cgptr a, cgptr b, int q) {
    double t = a[0] * b[0];
    for (int k = 1; k < q; k++) t -= a[k] * b[k];
    return t;
}
SCVX_HD void cone_nt(cgptr s, cgptr z, int q, gptr v, double& beta, gptr lam) {
    if (q == 1) { v[0] = sqrt(s[0] / z[0]); beta = 1.0; lam[0] = sqrt(s[0] * z[0]); return; }
    double s1 = 0, z1 = 0, sz = 0;
    for (int k = 1; k < q; k++) { s1 += s[k] * s[k]; z1 += z[k] * z[k]; sz += s[k] * z[k]; }
    const double ns = sqrt(s1), nz = sqrt(z1);
    const double sj = sqrt((s[0] - ns) * (s[0] + ns)), zj = sqrt((z[0] - nz) * (z[0] + nz));
    const double isj = 1.0 / sj, izj = 1.0 / zj;
    const double gam = sqrt(0.5 * (1.0 + (s[0] * z[0] + sz) * isj * izj));
    const double ig = 0.5 / gam;
    const double wb0 = (s[0] * isj + z[0] * izj) * ig;
    const double den = 1.0 / sqrt(2.0 * (wb0 + 1.0));
    v[0] = (wb0 + 1.0) * den;
    for (int k = 1; k < q; k++) v[k] = (s[k] * isj - z[k] * izj) * ig * den;
    beta = sqrt(sj * izj);
    // lam = W z
    double vz = 0;
    for (int k = 0; k < q; k++) vz += v[k] * z[k];
    lam[0] = beta * (2.0 * vz * v[0] - z[0]);
    for (int k = 1; k < q; k++) lam[k] = beta * (2.0 * vz * v[k] + z[k]);
}
// y = W x or W^-1 x (y may alias x)
SCVX_HD void cone_W(cgptr v, double beta, int q, cgptr x, gptr y, bool inverse) {
    if (q == 1) { y[0] = inverse ? x[0] / v[0] : x[0] * v[0]; return; }
    double vx = v[0] * x[0];
    if (!inverse) { for (int k = 1; k < q; k++) vx += v[k] * x[k]; }
    else { for (int k = 1; k < q; k++) vx -= v[k] * x[k]; }
    const double sc = inverse ? 1.0 / beta : beta;
    const double y0 = (2.0 * vx * v[0] - x[0]) * sc;
    for (int k = 1; k < q; k++) y[k] = ((inverse ? -2.0 : 2.0) * vx * v[k] + x[k]) * sc;
    y[0] = y0;
}
// o = a o b (Jordan product); o may alias a or b
SCVX_HD void cone_prod(cgptr a, cgptr b, int q, gptr o) {
    if (q == 1) { o[0] = a[0] * b[0]; return; }
    double dot = 0;
    for (int k = 0; k < q; k++) dot += a[k] * b[k];
    const double a0 = a[0], b0 = b[0];
    for (int k = 1; k < q; k++) o[k] = a0 * b[k] + b0 * a[k];
    o[0] = dot;
}
// o = lam \ d; o may alias d
SCVX_HD void cone_div(cgptr lam, cgptr d, int q, gptr o) {
    if (q == 1) { o[0] = d[0] / lam[0]; return; }
    double l1d1 = 0, l1l1 = 0;
    for (int k = 1; k < q; k++) { l1d1 += lam[k] * d[k]; l1l1 += lam[k] * lam[k]; }
    const double l0 = lam[0];
    const double det = l0 * l0 - l1l1;
    const double x0 = (l0 * d[0] - l1d1) / det;
    for (int k = 1; k < q; k++) o[k] = (d[k] - x0 * lam[k]) / l0;
    o[0] = x0;
}
// largest alpha with lam + alpha d in the cone
SCVX_HD double cone_maxstep(cgptr lam, cgptr d, int q) {
    if (q == 1) return d[0] < 0.0 ? -lam[0] / d[0] : INFINITY;
    return ipm::soc_maxstep_parts(lam[0], d[0], jdot(lam, lam, q), jdot(lam, d, q), jdot(d, d, q));
}
// smallest t with x + t e in the cone
SCVX_HD double cone_shift(cgptr x, int q) {
    if (q == 1) return -x[0];
    double n1 = 0;
    for (int k = 1; k < q; k++) n1 += x[k] * x[k];
    return sqrt(n1) - x[0];
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
    template <class F> SCVX_HD double sum(int n, F&& f) {
        double a = 0;
        for (int i = ex.lane(); i < n; i += ex.nlanes()) a += f(i);
        return ex.sum(a);
    }
    template <class F> SCVX_HD void each_cone(F&& f) {
        for (int c = ex.lane(); c < T.ncone; c += ex.nlanes()) {
            int off, q;
            cone_of(T.N, c, off, q);
            f(c, off, q);
        }
    }

    // o = [0 A'; A 0] v over the band positions (position nb, nkaR, has no equality entry)
    SCVX_HD void A_apply(cgptr v, gptr o) {
        each(T.nb, [&](int p) {
            double a = 0;
            for (int k = 0; k < AW; k++) {
                const int c = tt.a_col[p * AW + k];
                if (c >= 0) a += tt.a_val[p * AW + k] * v[c];
            }
            o[p] = a;
        });
        if (ex.lane() == 0) o[T.nb] = 0.0;
    }
    // o = E x (+ h)
    SCVX_HD void E_apply(cgptr v, gptr o, bool with_h) {
        each(T.m, [&](int r) {
            double a = tt.e_c0[r] * v[tt.e_v0[r]];
            const int v1 = tt.e_v1[r];
            if (v1 >= 0) a += tt.e_c1[r] * v[v1];
            o[r] = a + (with_h ? tt.e_h[r] : 0.0);
        });
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
    // o = W in or W^-1 in over all cones
    SCVX_HD void W_all(cgptr in, gptr o, bool inverse) {
        cgptr wv = w + L.wv;
        cgptr wb = w + L.wb;
        each_cone([&](int c, int off, int q) { cone_W(wv + off, wb[c], q, in + off, o + off, inverse); });
    }

    // ---- NT scaling from (s, z); lam = W z; the per-node blocks of H = E' W^-2 E; border data of the long cone ----
    SCVX_HD_NI void scale(bool identity) {
        TD_TS(t0_);
        gptr wv = w + L.wv;
        gptr wb = w + L.wb;
        if (identity) {
            each(T.m, [&](int r) { wv[r] = 0.0; });
            ex.sync();
            each_cone([&](int c, int off, int) { wv[off] = 1.0; wb[c] = 1.0; });
        } else {
            cgptr s = w + L.s;
            cgptr z = w + L.z;
            gptr lam = w + L.lam;
            each_cone([&](int c, int off, int q) { cone_nt(s + off, z + off, q, wv + off, wb[c], lam + off); });
        }
        ex.sync();
        TD_TE(t0_, 0);
        TD_TS(t1_);
        const int bo = NR * (T.N + 1);
        beta_big = wb[T.ncone - 1];
        const double v0 = wv[bo];
        w0_big = 2.0 * v0 * v0 - 1.0;
        const double ib2 = 1.0 / (beta_big * beta_big);
        gptr hd = w + L.hd;
        // one lane per node: its 7 cones in turn (they share entries)
        each(T.N + 1, [&](int i) {
            gptr H = hd + (size_t)i * NV * NV;
            for (int k = 0; k < NV * NV; k++) H[k] = 0.0;
            for (int cc = 0; cc < 7; cc++) {
                int off, q;
                cone_of(T.N, 7 * i + cc, off, q);
                const double b2 = 1.0 / (wb[7 * i + cc] * wb[7 * i + cc]);
                const double vh = wv[off];
                for (int a = 0; a < q; a++) {
                    for (int b = 0; b <= a; b++) {
                        double w2;
                        if (q == 1) w2 = 1.0 / (vh * vh);
                        else {
                            // W^-2 = beta^-2 (2 wh wh' - J), wh = (2 v0^2 - 1, -2 v0 v_tail)
                            const double wa = a == 0 ? 2.0 * vh * vh - 1.0 : -2.0 * vh * wv[off + a];
                            const double wbb = b == 0 ? 2.0 * vh * vh - 1.0 : -2.0 * vh * wv[off + b];
                            w2 = b2 * (2.0 * wa * wbb - (a == b ? (a == 0 ? 1.0 : -1.0) : 0.0));
                        }
                        // entries of rows a and b
                        const int ra = off + a, rb = off + b;
                        for (int ea = 0; ea < 2; ea++) {
                            const int va = ea == 0 ? tt.e_v0[ra] : tt.e_v1[ra];
                            if (va < 0) continue;
                            const double ca = ea == 0 ? tt.e_c0[ra] : tt.e_c1[ra];
                            const int la = (va - 7) % NP;
                            for (int eb = 0; eb < 2; eb++) {
                                const int vb = eb == 0 ? tt.e_v0[rb] : tt.e_v1[rb];
                                if (vb < 0) continue;
                                const double cb = eb == 0 ? tt.e_c0[rb] : tt.e_c1[rb];
                                const int lb_ = (vb - 7) % NP;
                                const double val = ca * w2 * cb;
                                H[la * NV + lb_] += val;
                                if (a != b) H[lb_ * NV + la] += val;
                            }
                        }
                    }
                }
            }
            H[11 * NV + 11] += ib2;   // the -J part of the long cone on kaR_i
        });
        ex.sync();
        TD_TE(t1_, 1);
    }

    // ---- banded LDL' of [H + delta I, A'; A, -delta I] without the border; L and 1/d to HBM ----
    // entry (c + d, c) of the band matrix without the border: constant part + the node block of H.  Branch-free: kc is
    // zero-padded past nb, positions outside a node block read the zero entry behind the H blocks.
    static SCVX_HD double kcol(cgptr __restrict__ kc, cgptr __restrict__ hd, int N, int c, int d) {
        const int pt = c - 7, i = pt / NP, l = pt - NP * i;
        const bool blk = pt >= 0 && i <= N && l + d < NV;   // l + d < NV implies l < NV: a variable column
        const int hz = (N + 1) * NV * NV;
        return kc[(size_t)c * BS + d] + hd[blk ? i * NV * NV + (l + d) * NV + l : hz];
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
        const int spare = (int)(dorig + NSLOT - win);   // dorig has NSLOT + 1 entries
        const int lane = ex.lane(), nl = ex.nlanes();
        constexpr int PER = (ST * BS + Ex::kLanes - 1) / Ex::kLanes;
        constexpr int PP = (NPAIR + Ex::kLanes - 1) / Ex::kLanes;
        for (int e = lane; e < NSLOT * BS; e += nl) {
            const int c = e / BS, d = e - c * BS;
            const double v = kcol(kc, hd, N, c, d);
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
                stage[q] = e < ST * BS ? kcol(kc, hd, N, cbase + e / BS, e % BS) : 0.0;
            }
            const int jend = j0 + ST < nb ? j0 + ST : nb;
            for (int j = j0; j < jend; j++) {
                const TD_LOCAL double* col = wn + (j % NSLOT) * BS;
                double dj = col[0];
                const bool var = is_var(N, j);
                // dynamic regularisation: the pivot keeps the sign quasi-definiteness gives it, and does not fall below
                // the rounding level of the entry it started from (an active cone's block is rank one to working
                // precision in the last iterations); the refinement passes absorb the perturbation
                const double fl = 1e-15 * dor[j % NSLOT] + 1e-13;
                if (var ? !(dj > fl) : !(dj < -fl)) { if (!(dj == dj)) ok = false; dj = var ? fl : -fl; }
                const double idj = 1.0 / dj;
                // all reads of the trailing update, then all writes: one LDS round trip per step (a lane without a
                // q-th entry updates the spare double behind the pivot scales)
                double tv[PP];
                int ti[PP];
                SCVX_UNROLL
                for (int q = 0; q < PP; q++) {
                    const int ab = pab[q], a = ab >> 8, b = ab & 255;
                    ti[q] = ab ? ((j + b) % NSLOT) * BS + (a - b) : spare;
                    tv[q] = wn[ti[q]] - col[a] * col[b] * idj;
                }
                SCVX_UNROLL
                for (int q = 0; q < PP; q++) wn[ti[q]] = tv[q];
                for (int d = lane; d < BS; d += nl) {
                    const double lv = col[d] * idj;
                    lb[(size_t)j * BS + d] = d == 0 ? idj : lv;
                    if (d >= 1 && j + d < nb) ut[(size_t)(j + d) * BS + d] = lv;
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
        W_all(bz, t1, true);
        ex.sync();
        W_all(t1, t1, true);
        ex.sync();
        each(T.nb, [&](int p) { xs[p] = bu[p] - Et_at(t1, p); });
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
        each(T.nb, [&](int p) { du[p] = xs[p] - y[p] * tt; });
        if (ex.lane() == 0) du[T.nb] = nu;
        ex.sync();
        E_apply(du, dz, false);
        ex.sync();
        each(T.m, [&](int r) { dz[r] = -(dz[r] + bz[r]); });
        ex.sync();
        W_all(dz, dz, true);
        ex.sync();
        W_all(dz, dz, true);
        ex.sync();
        TD_TE(t6_, 6);
    }
    // Newton solve with T.refine passes on the uncondensed residual; bu, bz are overwritten with the last residual
    SCVX_HD_NI void kkt_solve(gptr bu, gptr bz, gptr du, gptr dz) {
        condensed(bu, bz, du, dz);
        gptr r1 = w + L.tu;
        gptr r3 = w + L.t2;
        gptr ddu = w + L.tu2;
        gptr ddz = w + L.t3;
        for (int pass = 0; pass < T.refine; pass++) {
            // r1 = bu - ([0 A'; A 0] du - E' dz),  r3 = bz + E du + W^2 dz
            TD_TS(t7_);
            A_apply(du, r1);
            W_all(dz, r3, false);
            ex.sync();
            W_all(r3, r3, false);
            ex.sync();
            each(T.nb + 1, [&](int p) { r1[p] = bu[p] - r1[p] + Et_at(dz, p); });
            each(T.m, [&](int r) {
                double a = tt.e_c0[r] * du[tt.e_v0[r]];
                const int v1 = tt.e_v1[r];
                if (v1 >= 0) a += tt.e_c1[r] * du[v1];
                r3[r] = bz[r] + a + r3[r];
            });
            ex.sync();
            TD_TE(t7_, 7);
            condensed(r1, r3, ddu, ddz);
            each(T.nb + 1, [&](int p) { du[p] += ddu[p]; });
            each(T.m, [&](int r) { dz[r] += ddz[r]; });
            ex.sync();
        }
    }

    // shift x into the interior of the cone if it is not (CVXOPT initialisation)
    SCVX_HD_NI void shift_in(gptr x) {
        double t = -INFINITY, n2 = 0;
        for (int c = ex.lane(); c < T.ncone; c += ex.nlanes()) {
            int off, q;
            cone_of(T.N, c, off, q);
            const double tc = cone_shift(x + off, q);
            t = tc > t ? tc : t;
            for (int k = 0; k < q; k++) n2 += x[off + k] * x[off + k];
        }
        t = -ex.min(-t);
        n2 = ex.sum(n2);
        const double nrm = sqrt(n2);
        if (t >= -1e-8 * (nrm > 1.0 ? nrm : 1.0)) {
            each_cone([&](int, int off, int) { x[off] += 1.0 + t; });
        }
        ex.sync();
    }
    // min over cones of the largest step keeping lam + alpha d inside
    SCVX_HD_NI double max_step(cgptr d) {
        cgptr lam = w + L.lam;
        double a = INFINITY;
        for (int c = ex.lane(); c < T.ncone; c += ex.nlanes()) {
            int off, q;
            cone_of(T.N, c, off, q);
            const double ac = cone_maxstep(lam + off, d + off, q);
            a = ac < a ? ac : a;
        }
        return ex.min(a);
    }

    SCVX_HD Result solve(const double* ic_, double* out_) {
        cgptr ic = (cgptr)ic_;
        gptr out = (gptr)out_;
        icv = ic;
        gptr u = w + L.u, s = w + L.s, z = w + L.z, lam = w + L.lam;
        gptr ru = w + L.ru, rz = w + L.rz, du = w + L.du, dz = w + L.dz, ds = w + L.ds;
        gptr bu = w + L.bu, bz = w + L.bz, t1 = w + L.t1, t2 = w + L.t2, t3 = w + L.t3;
        const int nb = T.nb, m = T.m, N = T.N;
        TD_TS(tt_);
        Result R;
        R.status = TD_ITER_CAP; R.iters = 0; R.pobj = 0; R.gap = INFINITY; R.pres = INFINITY; R.dres = INFINITY;
        double b2 = T.b2_rest;
        for (int i = 0; i < 6; i++) b2 += icv[i] * icv[i];
        const double nrm_b = sqrt(b2) > 1.0 ? sqrt(b2) : 1.0;

        // the zero entry behind the H blocks and the padding around L (read, masked, by the window fetch and the sweeps)
        each(8, [&](int k) { (w + L.hd)[(size_t)(N + 1) * NV * NV + k] = 0.0; });
        each(LPAD, [&](int k) {
            (w + L.lb)[-1 - k] = 0.0; (w + L.lb)[(size_t)nb * BS + k] = 0.0;
            (w + L.ut)[-1 - k] = 0.0; (w + L.ut)[(size_t)nb * BS + k] = 0.0;
        });
        ex.sync();
        // ---- initial point: W = I,  [0 A' G'; A 0 0; G 0 -I][x; y; z] = [-c; b; h],  s = -z, shifted into the cone ----
        scale(true);
        bool ok = factor_all();
        each(nb + 1, [&](int p) { bu[p] = (p == nb || is_var(N, p)) ? -tt.q[p] : bq(p); });
        each(m, [&](int r) { bz[r] = tt.e_h[r]; });
        ex.sync();
        kkt_solve(bu, bz, u, z);
        each(m, [&](int r) { s[r] = -z[r]; });
        ex.sync();
        shift_in(s);
        shift_in(z);

        const int degree = T.ncone;
        double best_pres = INFINITY;
        int flat = 0;
        for (int it = 1; it <= T.max_iter; it++) {
            R.iters = it;
            TD_TS(t8_);
            // residuals: ru = [c + A'y - E'z at variables; A x - b at equalities], rz = s - e(x)
            A_apply(u, ru);
            E_apply(u, rz, true);
            ex.sync();
            double pobj = 0, dobj = 0, nx = 0, ny = 0;
            for (int p = ex.lane(); p <= nb; p += ex.nlanes()) {
                const bool var = p == nb || is_var(N, p);
                const double qp = var ? tt.q[p] : bq(p);
                if (var) {
                    const double r = qp + ru[p] - Et_at(z, p);
                    ru[p] = r; nx += r * r; pobj += qp * u[p];
                } else {
                    const double r = ru[p] - qp;
                    ru[p] = r; ny += r * r; dobj -= qp * u[p];
                }
            }
            double gap = 0, nz = 0;
            for (int r = ex.lane(); r < m; r += ex.nlanes()) {
                const double rr = s[r] - rz[r];
                rz[r] = rr; nz += rr * rr; gap += s[r] * z[r]; dobj -= tt.e_h[r] * z[r];
            }
            pobj = ex.sum(pobj); dobj = ex.sum(dobj); nx = ex.sum(nx); ny = ex.sum(ny); nz = ex.sum(nz); gap = ex.sum(gap);
            ex.sync();
            const double pres = fmax(sqrt(ny) / nrm_b, sqrt(nz) / T.nrm_h), dres = sqrt(nx) / T.nrm_c;
            const double relgap = gap / fmax(1.0, fmax(fabs(pobj), fabs(dobj)));
            R.pobj = pobj; R.gap = gap; R.pres = pres; R.dres = dres;
            TD_TE(t8_, 8);
            SCVX_DBG("td %3d pobj %+.10e dobj %+.10e gap %.2e pres %.2e dres %.2e\n", it, pobj, dobj, gap, pres, dres);
            if (!(pres == pres) || !(dres == dres) || !(gap == gap) || !ok) { R.status = TD_NONFINITE; break; }
            if (pres < T.tol && dres < T.tol && (gap < T.tol || relgap < T.tol)) { R.status = TD_OPTIMAL; break; }
            // primal infeasibility shows as a primal residual that stops falling while complementarity and the dual
            // residual converge (the multipliers run off along a Farkas ray, dobj grows without bound)
            if (pres < 0.9 * best_pres) { best_pres = pres; flat = 0; } else flat++;
            if (flat >= 5 && dres < 1e-6 && relgap < 1e-6) { R.status = TD_INFEASIBLE; break; }
            if (it == T.max_iter) break;

            scale(false);
            ok = factor_all();
            if (!ok) { R.status = TD_NONFINITE; break; }
            const double mu = gap / degree;

            // predictor: ds_rhs = -lam o lam, so W (lam \ ds_rhs) = -W lam = -s:  bz = -rz + s
            each(nb + 1, [&](int p) { bu[p] = -ru[p]; });
            each(m, [&](int r) { bz[r] = -rz[r] + s[r]; });
            ex.sync();
            kkt_solve(bu, bz, du, dz);
            TD_TS(t9_);
            E_apply(du, ds, false);
            ex.sync();
            each(m, [&](int r) { ds[r] = -rz[r] + ds[r]; });
            ex.sync();
            W_all(ds, t1, true);    // W^-1 ds
            W_all(dz, t2, false);   // W dz
            ex.sync();
            double alpha = fmin(1.0, fmin(max_step(t1), max_step(t2)));
            const double sigma = (1.0 - alpha) * (1.0 - alpha) * (1.0 - alpha);
            // combined: ds_rhs = -lam o lam - (W^-1 ds_a) o (W dz_a) + sigma mu e;  bz = -rz - W (lam \ ds_rhs)
            each_cone([&](int, int off, int q) {
                cone_prod(t1 + off, t2 + off, q, t1 + off);
                cone_prod(lam + off, lam + off, q, t2 + off);
                for (int k = 0; k < q; k++) t1[off + k] = -t2[off + k] - t1[off + k];
                t1[off] += sigma * mu;
                cone_div(lam + off, t1 + off, q, t1 + off);
            });
            ex.sync();
            W_all(t1, t1, false);
            ex.sync();
            each(nb + 1, [&](int p) { bu[p] = -ru[p]; });
            each(m, [&](int r) { bz[r] = -rz[r] - t1[r]; });
            ex.sync();
            TD_TE(t9_, 9);
            kkt_solve(bu, bz, du, dz);
            TD_TS(t10_);
            E_apply(du, ds, false);
            ex.sync();
            each(m, [&](int r) { ds[r] = -rz[r] + ds[r]; });
            ex.sync();
            W_all(ds, t1, true);
            W_all(dz, t2, false);
            ex.sync();
            alpha = fmin(1.0, 0.99 * fmin(max_step(t1), max_step(t2)));
            if (!(alpha >= 1e-8)) { R.status = alpha == alpha ? TD_STALLED : TD_NONFINITE; break; }
            each(nb + 1, [&](int p) { u[p] += alpha * du[p]; });
            each(m, [&](int r) { z[r] += alpha * dz[r]; s[r] += alpha * ds[r]; });
            ex.sync();
            TD_TE(t10_, 10);
        }
        // the variables, node by node, then nkaR
        each((N + 1) * NV, [&](int k) { out[k] = u[pos_z(k / NV, k % NV)]; });
        if (ex.lane() == 0) out[(N + 1) * NV] = u[nb];
        ex.sync();
        TD_TE(tt_, 15);
        (void)t3;
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
    t.a_col = H.a_col.data(); t.a_val = H.a_val.data(); t.kc = H.kc.data();
    t.e_v0 = H.e_v0.data(); t.e_v1 = H.e_v1.data(); t.e_c0 = H.e_c0.data(); t.e_c1 = H.e_c1.data(); t.e_h = H.e_h.data();
    t.t_row = H.t_row.data(); t.t_coef = H.t_coef.data(); t.q = H.q.data();
    return nullptr;
}

}  // namespace td
}  // namespace scvx
