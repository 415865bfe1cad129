// Structure-exploiting interior-point solver for the SCvx trust-region SOCP — portable core.
//
// Replaces MOI.optimize!(model) (rocketland.jl:271; Mosek / ECOS interior-point solvers) for the model
// Rocketland.build_model assembles (rocketland.jl:53-219).  One *executor* solves one trajectory's
// subproblem; the executor abstraction `Ex` supplies lane(), nlanes(), sync(), sum(), min(), scratch():
//     device  : one 64-lane wavefront per trajectory (WaveEx, scvx_batch.hip) or four per trajectory for small
//               batches (BlockEx), reductions by cross-lane shuffles, scratch tiles in LDS
//     host    : one thread per trajectory (the CPU twin timed as bench.py's cpu_baseline)
// The algorithm (Mehrotra predictor-corrector, Nesterov-Todd scaling, CVXOPT-style initial point) and
// every formula follow the validated numpy twin oracle/ipm_struct.py; see DESIGN.md §SOCP for the maths.
//
// Reduced variables  w = (dx[K+1][14], du[K+1][NU], nu[K][14], s, tnu, ttr, ts)   ("var vector", NV); NU = 3, or 5 with the
// fin extension (build-defined, SURVEY N2: u[4:5] = fin force coordinates, cone |u[4:5]| <= finmxf at every node,
// rocketland.jl:203-209 as commented there)
// Cone vector layout ("cone vector", NC):
//     gs[K][3] tilt[K][3] rate[K][4] mass[K] tb[K+1][4] tc[K+1][4] lb[K+1] fin[nfin][3] dp[ndp][4] nu[14K+1] tr[(14+NU)(K+1)+1] sg[2] rk[1]
//     (fin: nfin = K+1 when NU = 5, else 0)
//     (dp: the optional dynamic-pressure cones, ndp = K when Consts::vmax > 0, else 0)
#pragma once
#include <math.h>
#include <stdint.h>
#include <type_traits>

#if defined(__HIPCC__)
#define SCVX_HD __host__ __device__ __forceinline__
#define SCVX_HD_NI __host__ __device__ __attribute__((noinline))
#else
#define SCVX_HD inline
#define SCVX_HD_NI inline
#endif
// optional in-kernel section timers (diagnostic builds only: -DSCVX_IPM_PROF)
#if defined(SCVX_IPM_PROF) && defined(__HIP_DEVICE_COMPILE__)
#define SCVX_TS(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define SCVX_TE(v, slot) prof[slot] += (double)(__builtin_amdgcn_s_memtime() - v)
#else
#define SCVX_TS(v)
#define SCVX_TE(v, slot)
#endif
// Solver status (Result::status, surfaced by scvx_batch_get_solver_stats):
//     0 optimal          pres, dres and relgap all below tol
//     1 iteration cap    best iterate returned, merit >= accept
//     2 stalled          numerical floor / KKT breakdown with merit >= accept
//     3 non-finite
//     5 infeasible       a boundary value fixed by rocketland.jl:109-115 violates a path cone of node 1 (no interior
//                        point exists; an interior-point iteration would only diverge)
//     4 almost optimal   stopped on the numerical floor with tol <= merit < accept (MOI's ALMOST_OPTIMAL band; the
//                        reference treats anything but OPTIMAL as an error, rocketland.jl:273-276 -- set accept = tol
//                        to get exactly that)
// iterations without a new best merit (below 1e-5) after which the iterate is taken to sit on its numerical floor
// The r block and the thrust block of a node: assembled and Cholesky-factorised (0, default) or inverted from their square-root
// factors by a Householder QR (1).  Measured on the twin, first failures over 60 random classes / 40 fin classes
// (tools/k4_fuzz.py, profiles/r03_k4_fuzz.md): 1.77 % / 1.30 % assembled, 2.09 % / 1.26 % square-root -- no gain: what limits
// those solves is the elimination of dz (W^-2 has condition ~ v0^4), not the factorisation of the 3x3 blocks.  Kept as a switch.
#ifndef SCVX_NODE_QR
#define SCVX_NODE_QR 0
#endif
#ifndef SCVX_STALL_ITERS
#define SCVX_STALL_ITERS 3
#endif
// factor of Mehrotra's balancing shift of the starting point (0 disables)
#ifndef SCVX_INIT_BALANCE
#define SCVX_INIT_BALANCE 1.0
#endif
// how far inside its cone (in multiples of e) every block of the least-squares starting point is put (shift_into_cone)
#ifndef SCVX_INIT_SHIFT
#define SCVX_INIT_SHIFT 0.25
#endif
// a solve after a rejected step restarts from the kept iterate while that iterate's trust-region norm is below this fraction of the new radius
// (twin, bench mix, iterations per solve: 0.7 13.23, 0.8 13.16, 0.9 13.03, 1.0 12.74, 1.1 12.70, 1.2 12.77, 1.5 12.87, 2.0 14.36 with
// failures, no condition 16.84 with failures: the kept iterate must still lie inside the new radius)
// (those figures with the iterate kept at merit 1e-4).  Just below 1: the recomputed radius slack then stays above its floor (1e-3 rk).
#ifndef SCVX_WARM_RADIUS
#define SCVX_WARM_RADIUS 0.999
#endif
// weight of the kept optimum in the starting point of a solve whose new radius cuts that optimum off (0 = plain cold start, the
// default), and the fraction of the new radius it is pulled to.  OFF: 0.8 saves 4-7 % of the iterations (comment at its use), but a
// run with it and a run with cold starts then end 1.6e-4 apart in x after 14 solve_steps (16 dispersed trajectories, device) where
// they end 1e-6 apart without it -- the two take different paths to each flat optimum and the outer iteration carries the difference
// along.  Closeness of repeated runs (and of the run to the oracle's) is worth more than those iterations.
#ifndef SCVX_BLEND_WARM
#define SCVX_BLEND_WARM 0.0
#endif
#ifndef SCVX_BLEND_RADIUS
#define SCVX_BLEND_RADIUS 0.9
#endif
// fraction of the distance to the cone boundary taken by the combined step
#ifndef SCVX_STEP_FRAC
#define SCVX_STEP_FRAC 0.98
#endif
// iterative refinement of the predictor's solve as well (1) or of the corrector's only (0): the predictor only sets the
// centering parameter and the second-order term; refining it measured no change in iterations or final merit at
// B = 8192 and costs 6 % of the throughput
#ifndef SCVX_REFINE_PRED
#define SCVX_REFINE_PRED 0
#endif
// merit growth factor over the best iterate (once that is inside the acceptance band) that ends the solve at once:
// past the numerical floor the dual residual jumps by orders of magnitude from one iterate to the next
#ifndef SCVX_BLOWUP_STOP
#define SCVX_BLOWUP_STOP 10.0
#endif
// Warm start of the solve that follows a REJECTED step (same about / dynam, radius halved: rocketland.jl:299-301): the
// iterate of the previous solve at the first merit below this value is kept in the workspace and the next solve of the
// same subproblem data starts from it (only the radius row differs, and it is used only while that row stays inactive:
// the old central path is then the new one).  Measured on 256 trajectories x 14 steps (tools/twin_stats.py --warm), IPM
// iterations of a warm-started solve / mean over all solves: kept at 1e-2: 7.3 / 15.2, 1e-3: 6.0 / 14.7, 1e-4: 5.0 / 14.3,
// 1e-5: 3.9 / 13.9 (cold: 19.0 / 19.7); every solve status 0 in all four.  Rounds 2-3 shipped 1e-4 ("a well-centred iterate").
// Late round 3, with the kept iterate required to lie inside the new radius (SCVX_WARM_RADIUS), later is better all the way to the
// optimum itself -- mean iterations per solve on the bench mix / first failures on 100 random classes (tools/k4_fuzz.py):
// 1e-4 12.74 / 1.19 %, 1e-5 12.26, 1e-6 11.83, 1e-7 11.44, 3e-8 11.25 / 1.03 %, the optimal iterate (merit < tol) 11.10 / 0.96 %.
// Kept at the optimum, the next solve's first residual evaluation -- of the NEW problem, with the new radius row -- usually finds
// the optimality conditions met and returns after that evaluation (status 0, 1 iteration, no factorisation); otherwise the iteration
// carries on from there.  The value is a multiple of the solver tolerance.
#ifndef SCVX_WARM_SAVE
#define SCVX_WARM_SAVE 1.0
#endif
// Storage type of the block-tridiagonal FACTOR (Solver's FStor: packed L_k^-1 and the coupling tiles N_k, 301 numbers per segment,
// read four times per interior-point iteration).  float (measured in round 4, profiles/r04_factor_f32.md): the
// tiles are computed in double and rounded once when stored, every load widens, and the banded solves they serve sit under the
// double-precision operator refinement of newton_solve.  NUMERICALLY FREE -- twin, 64 dispersed trajectories x 14 solve_steps: 11.17
// iterations per solve either way, 1,842 vs 1,861 refinement corrections, every solve optimal; first failures over 60 random classes
// 1.39 % vs 1.41 % -- and it moves 9.6 % fewer bytes per launch (PMC: 0.369 vs 0.408 TB), yet the kernel is 8 % SLOWER on the device
// (83.5 vs 77.1 ms per launch of the bench mix): the conic solve's time is not set by its byte count alone.  Default: double.
#ifndef SCVX_FACTOR_T
#define SCVX_FACTOR_T double
#endif
#ifndef SCVX_STREAM_U
#define SCVX_STREAM_U 4   // elements in flight per lane in the streaming loops (Solver::stream)
#endif
#ifndef SCVX_REFINE_FROM
#define SCVX_REFINE_FROM 1e-4
#endif
// the corrector's refinement check folded into the solve and the direction pass (Solver::newton_corr, round 5); 0 = the separate
// H_apply / E'dy passes of rounds 2-4 (newton_solve)
#ifndef SCVX_REFINE_FUSED
#define SCVX_REFINE_FUSED 1
#endif
// E'y and E V formed inside the factorisation loop instead of by two passes over D of their own (build_kkt(res), round 5); 0 = the separate
// passes.  MEASURED AND OFF (profiles/r05_k4_byte_budget.md section 4): it removes 5.3 % of the kernel's HBM traffic (PMC: 3.35 -> 3.17 MB per
// interior-point iteration) and makes the kernel 9.5 % SLOWER (64.75 -> 70.9 ms per launch of the bench mix): the two small matrix-vector
// products, their node slices and stores add ~8 k cycles to each of the 50 dependent steps of the loop (in-kernel timers: loop 19.2 M ->
// 25.8 M cycles per solve), and a wavefront's time in that loop is NOT hidden behind the other wavefronts' streaming -- the loop's latency
// and the streamed bytes add up.  Kept as a switch because it is the measurement that says so.
#ifndef SCVX_FUSED_RES
#define SCVX_FUSED_RES 0
#endif
// newton_corr inlined into attempt_solve (SCVX_HD) or a routine of its own (SCVX_HD_NI: scratch 832 -> 752 B per lane; B = 8192 +0.9 %,
// B = 1024 -1.3 % -- measured, the headline's choice stays)
#ifndef SCVX_NEWTON_CORR_ATTR
#define SCVX_NEWTON_CORR_ATTR SCVX_HD
#endif
// the big cones' reduction sums carried from update_pass to the next scale_pass (1) or re-taken by two sweeps per cone (0).  MEASURED
// AND OFF (round 5, profiles/r05_k4_byte_budget.md section 5): -0.9 % at B = 8192 and B = 1024 (four sweeps less per iteration), same iteration
// counts on the sample problems, but first failures over 40 random classes 1.42 % -> 1.56 % single-attempt with 7 non-finite exits (the
// expanded sums cancel where a cone's s and z are not yet near-complementary): not worth the robustness.
#ifndef SCVX_CARRY_BIGSUMS
#define SCVX_CARRY_BIGSUMS 0
#endif
// CARRIED RESIDUALS (round 6).  The dual and the equality residual are LINEAR in the iterate (rx = c + E'y - J'Z, ry = E V + dk) and the
// Newton direction cancels them exactly, so after a step of length alpha  rx+ = (1 - alpha) rx - alpha r1,  ry+ = (1 - alpha) ry - alpha r2
// with r1, r2 the residuals of the reduced KKT solve -- 1e-6 ... 1e-10 of the right-hand side, irrelevant while the residuals themselves
// are large.  While the merit of an iterate is above SCVX_RESID_FRESH_FROM the next iteration therefore does not re-evaluate them (one
// cone-map pass, E'y over A' and D, E V over D, a norm: 0.34 of the 3.2 MB an iteration moves) but SCALES the vectors and their norms; from
// that merit on -- the endgame, where the solve errors are what limits the solver (section 2.2) and where every stopping test is taken --
// and at least every eighth iteration they are evaluated as before.  A carried merit below the threshold is re-evaluated in the same
// iteration, so no test ever passes on carried values.  0 = evaluate in every iteration (rounds 1-5).
#ifndef SCVX_RESID_UPDATE
#define SCVX_RESID_UPDATE 1
#endif
#ifndef SCVX_RESID_FRESH_FROM
#define SCVX_RESID_FRESH_FROM 1e-4
#endif
// 1: the two-ended factorisation (four wavefronts per trajectory) carries the border in t-space like the other forms (round 6); 0: the
// round-5 border (four back-substituted systems after the loop), kept for A/B runs
#ifndef SCVX_TWISTED_TSPACE
#define SCVX_TWISTED_TSPACE 1
#endif
// refinement of a Newton solve stops once its first-row residual is below this fraction of the dual tolerance
#ifndef SCVX_REFINE_STOP
#define SCVX_REFINE_STOP 0.1
#endif
#if defined(__HIPCC__)
#define SCVX_UNROLL _Pragma("unroll")
#else
#define SCVX_UNROLL
#endif
// The device executors keep the solver object and the constants in LDS (socp_body's frame): told so, the compiler addresses the members
// with ds_read / ds_write instead of flat instructions (which also count against vmcnt: a member read inside a streaming loop then
// waits for every global load in flight).  First statement of every non-inlined routine.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SCVX_NO_LDS_FRAME)
#define SCVX_THIS_LDS() do { __builtin_assume(__builtin_amdgcn_is_shared((const void*)this)); __builtin_assume(__builtin_amdgcn_is_shared((const void*)&C)); } while (0)
#else
#define SCVX_THIS_LDS() do { } while (0)
#endif
#define SCVX_T0() SCVX_TS(t0_)
#define SCVX_T1(slot) SCVX_TE(t0_, slot)
#if defined(SCVX_IPM_DEBUG) && !defined(__HIPCC__)
#include <stdio.h>
#ifndef SCVX_DBG_NODE
#define SCVX_DBG_NODE -1
#endif
#define SCVX_DBG(...) fprintf(stderr, __VA_ARGS__)
#else
#define SCVX_DBG(...)
#endif

// optional host-side event counters (diagnostic twin builds only: -DSCVX_COUNTERS; tools/twin_stats.py --counters)
#if defined(SCVX_COUNTERS) && !defined(__HIPCC__)
extern "C" long long scvx_counters[8];   // 0 newton solves, 1 refinement checks (operator-form residual), 2 refinement correction solves, 3 factorisations
#define SCVX_COUNT(i) do { _Pragma("omp atomic") scvx_counters[i]++; } while (0)
#else
#define SCVX_COUNT(i)
#endif

namespace scvx {
namespace ipm {

// Pointers into the solver's HBM-resident state are typed with the global address space on the device, so every
// access compiles to global_load/global_store (vmcnt only).  A plain `double*` kept in the solver object is
// reloaded from memory in each non-inlined routine, loses its address space and becomes a flat access, which counts
// against lgkmcnt as well -- and every wait for an LDS tile then also waits for the HBM traffic in flight.
//
// STORAGE TYPE.  The solver is a template on the element type T of what it keeps in HBM: the linearisation D it reads
// and its whole workspace.  T = double is the reference precision; T = float ("f32 storage", BASELINE configs[3-4])
// halves the bytes of a kernel that is bound by them.  All arithmetic is double either way: every load widens, every
// store rounds, and norms / dot products / the Schur pivots accumulate in double registers and LDS.  The
// SCvx iterate itself (xbar, ubar, endpoint: 1,600 values per trajectory) stays double in both modes.
#if defined(__HIP_DEVICE_COMPILE__)
#define SCVX_GLOBAL __attribute__((address_space(1)))
#else
#define SCVX_GLOBAL
#endif
template <class T> struct gp {
    typedef SCVX_GLOBAL T elem;
    typedef elem* ptr;
    typedef const elem* cptr;
};
typedef gp<double>::elem gdouble;
typedef gp<double>::ptr gptr;    // double storage: what the device executors are written against
typedef gp<double>::cptr cgptr;

struct Consts {
    int K, max_iter, refine, pad;
    int warm, retries;   // warm: warm-start the solve that follows a rejected step (scvx_solver_opts.warm_start); retries: see Solver::solve
    double tol, accept;   // accept: acceptance band of a floor-limited iterate (status 4), >= tol
    double itan, sqcm, icos, Tmax, Tmin, omMax, mdry, wNu, mwet;
    double finmxf; // fin extension: |u[4:5]| <= finmxf (rocketland.jl:205; read only by the NU = 5 instantiation)
    double vmax;   // dynamic-pressure limit |v_k| <= sqrt(2 dpMax / rho) (master.jl:27,30; 0 = not enforced, as in the reference)
    double rIf[3], vIf[3], qBIf[4], wBi[3], wBf[3];
};

// Packed storage of the lower-triangular 14x14 inverse factor L_k^-1 (105 doubles instead of 196: the factor is read
// four times per solve from HBM).  Row i (i + 1 entries) shares a 15-slot line with row 13 - i (14 - i entries):
//     entry (i, j), j <= i, lives at  linv_row(i) + j.
enum { LINV_SZ = 105 };
SCVX_HD int linv_row(int i) { return i < 7 ? 15 * i : 15 * (13 - i) + (14 - i); }

// compact per-node inverse of the x-block of Hb: [hm | Hr 3x3 | Hv 3x3 | hq | Hq34 2x2 | Hw 3x3] = 33 doubles
// (Hv is a multiple of the identity unless the dynamic-pressure cone is enforced).  The u-block of a node follows:
// 3x3 thrust block (9 doubles), and with the fin extension the 2x2 fin block (4 more): hu_size(NU).
enum { HX_M = 0, HX_R = 1, HX_V = 10, HX_Q = 19, HX_Q34 = 20, HX_W = 24, HX_SZ = 33 };
SCVX_HD constexpr int hu_size(int nu) { return nu == 5 ? 13 : 9; }

struct Layout {
    int K, nx, nu_, nloc, nv, iS, iTNU, iTTR, iTS;
    int o_gs, o_tilt, o_rate, o_mass, o_tb, o_tc, o_lb, o_fin, o_dp, o_nu, o_tr, o_sg, o_rk, nc;
    int c_gs, c_tilt, c_rate, c_mass, c_tb, c_tc, c_lb, c_fin, c_dp, c_nu, c_tr, c_sg, c_rk, ncones, nsmall;
    int ndp, nfin, NU;
    int ny;
    SCVX_HD void init(int K_, bool with_dp = false, int nu = 3) {
        K = K_;
        NU = nu == 5 ? 5 : 3;
        ndp = with_dp ? K_ : 0;
        nfin = NU == 5 ? K_ + 1 : 0;
        nx = 14 * (K + 1);
        nu_ = NU * (K + 1);
        nloc = nx + nu_ + 14 * K;
        iS = nloc; iTNU = nloc + 1; iTTR = nloc + 2; iTS = nloc + 3;
        nv = nloc + 4;
        ny = 14 * K;
        o_gs = 0; o_tilt = 3 * K; o_rate = 6 * K; o_mass = 10 * K; o_tb = 11 * K;
        o_tc = o_tb + 4 * (K + 1); o_lb = o_tc + 4 * (K + 1); o_fin = o_lb + (K + 1); o_dp = o_fin + 3 * nfin; o_nu = o_dp + 4 * ndp;
        o_tr = o_nu + 14 * K + 1; o_sg = o_tr + (14 + NU) * (K + 1) + 1; o_rk = o_sg + 2; nc = o_rk + 1;
        c_gs = 0; c_tilt = K; c_rate = 2 * K; c_mass = 3 * K; c_tb = 4 * K; c_tc = c_tb + K + 1; c_lb = c_tc + K + 1;
        c_fin = c_lb + K + 1;
        c_dp = c_fin + nfin;
        nsmall = c_dp + ndp;
        c_nu = nsmall; c_tr = nsmall + 1; c_sg = nsmall + 2; c_rk = nsmall + 3; ncones = nsmall + 4;
    }
    // doubles of per-trajectory workspace
    SCVX_HD size_t work_doubles() const {
        size_t n = 0;
        n += (size_t)ny;                 // dk
        n += (size_t)nv * 8;             // V, rx, gx, dw, r1, cw, Vbest, tmpv
        n += (size_t)ny * 10;            // y, ry, dy, r2, cy, tmpy, tmpy2, rp, tq0, tq1
        n += (size_t)nc * 8;             // S, Z, lam, Wv, Wibz, tmpc, Wirz, sd
        n += (size_t)ncones;             // Wbeta
        n += (size_t)(K + 1) * HX_SZ + (size_t)(K + 1) * hu_size(NU);  // hx, hu
        n += (size_t)K * (LINV_SZ + 196);  // Linv (packed lower triangle), Nf
        n += (size_t)K * 196;            // At: the state blocks A_k of D transposed (coalesced E' products)
        n += (size_t)ny;                 // tchain
        n += (size_t)ny * 4 + nloc;      // ys, ytr, ynu, rtr, ptl
        n += (size_t)nloc * 2;           // tmpl, tmpl2
        n += (size_t)3 * (K + 1);        // uhat (thrust part of the control)
        n += (size_t)(K + 1);            // lb0
        n += 64;                         // scalars
        n += (size_t)nv + ny + 2 * (size_t)nc + 8;   // warm-start iterate (Vw, yw, Sw, Zw) + its header
        return n;
    }
};

// ------------------------------------------------------------------------------------------------
// scalar helpers for small second-order cones (dim <= 4), W = beta (2 v v' - J)
// ------------------------------------------------------------------------------------------------
template <class PS, class PZ, class PV>
SCVX_HD void soc_nt_small(PS s, PZ z, int d, PV v, double& beta) {
    double s1 = 0, z1 = 0, sz = 0;
    for (int i = 1; i < d; i++) { s1 += s[i] * s[i]; z1 += z[i] * z[i]; sz += s[i] * z[i]; }
    const double sj = sqrt(s[0] * s[0] - s1), zj = sqrt(z[0] * z[0] - z1);
    const double isj = 1.0 / sj, izj = 1.0 / zj;
    const double gam = sqrt(0.5 * (1.0 + (s[0] * z[0] + sz) * isj * izj));
    const double ig = 0.5 / gam;
    const double wb0 = (s[0] * isj + z[0] * izj) * ig;
    const double den = 1.0 / sqrt(2.0 * (wb0 + 1.0));
    v[0] = (wb0 + 1.0) * den;
    for (int i = 1; i < d; i++) v[i] = (s[i] * isj - z[i] * izj) * ig * den;
    beta = sqrt(sj * izj);
}
// y = W x (inverse=false) or W^-1 x
template <class PV, class PX, class PY>
SCVX_HD void soc_W_small(PV v, double beta, int d, PX x, PY y, bool inverse) {
    double vx = v[0] * x[0];
    if (!inverse) { for (int i = 1; i < d; i++) vx += v[i] * x[i]; }
    else { for (int i = 1; i < d; i++) vx -= v[i] * x[i]; }
    const double sc = inverse ? 1.0 / beta : beta;
    const double y0 = (2.0 * vx * v[0] - x[0]) * sc;
    for (int i = 1; i < d; i++) y[i] = ((inverse ? -2.0 : 2.0) * vx * v[i] + x[i]) * sc;
    y[0] = y0;
}
// W^-2 = [[h00, h01 v1'],[h01 v1, b2 I + h11 v1 v1']]
SCVX_HD void soc_w2(const double v0, const double n1, const double beta, double& h00, double& h01, double& h11, double& b2) {
    b2 = 1.0 / (beta * beta);
    const double a = 2.0 * v0 * v0 - 1.0;
    h00 = b2 * (a * a + 4.0 * v0 * v0 * n1);
    h01 = b2 * (-4.0 * v0 * (v0 * v0 + n1));
    h11 = b2 * 8.0 * v0 * v0;
}
SCVX_HD double soc_maxstep_parts(double l0, double d0, double ll, double ld, double dd) {
    // ll = lam'J lam, ld = lam'J d, dd = d'J d
    double amax = INFINITY;
    if (d0 < 0.0) amax = -l0 / d0;
    const double a = dd, b = 2.0 * ld, c = ll;
    const double disc = b * b - 4.0 * a * c;
    if (disc >= 0.0) {
        const double sq = sqrt(disc);
        const double qq = -0.5 * (b + (b >= 0.0 ? sq : -sq));
        const double r1 = (qq != 0.0) ? c / qq : INFINITY;
        const double r2 = (a != 0.0) ? qq / a : INFINITY;
        if (r1 > 0.0 && r1 < amax) amax = r1;
        if (r2 > 0.0 && r2 < amax) amax = r2;
    }
    return amax;
}
// inverse of a symmetric positive definite 3x3 (row-major in/out) through its Cholesky factor: backward
// stable for the nearly rank-one blocks d I + kappa v v' an active cone produces (the cofactor formula is not).
// In the last iterations (mu ~ 1e-9) such a block is rank-one to working precision and a Schur pivot can round to
// zero or below: pivots are floored at 1e-15 of their diagonal entry (the dynamic regularisation of the 14x14 tiles).
SCVX_HD double piv_floor(double p, double diag) { const double f = 1e-15 * diag; return p > f ? p : f; }
template <class PO>
SCVX_HD void inv3(const double* M, PO Mi) {
    const double l00 = sqrt(M[0]);
    const double l10 = M[3] / l00, l20 = M[6] / l00;
    const double l11 = sqrt(piv_floor(M[4] - l10 * l10, M[4]));
    const double l21 = (M[7] - l20 * l10) / l11;
    const double l22 = sqrt(piv_floor(M[8] - l20 * l20 - l21 * l21, M[8]));
    // Linv (lower)
    const double i00 = 1.0 / l00, i11 = 1.0 / l11, i22 = 1.0 / l22;
    const double i10 = -l10 * i00 * i11;
    const double i21 = -l21 * i11 * i22;
    const double i20 = -(l20 * i00 + l21 * i10) * i22;
    // Mi = Linv' Linv
    Mi[0] = i00 * i00 + i10 * i10 + i20 * i20;
    Mi[1] = Mi[3] = i10 * i11 + i20 * i21;
    Mi[2] = Mi[6] = i20 * i22;
    Mi[4] = i11 * i11 + i21 * i21;
    Mi[5] = Mi[7] = i21 * i22;
    Mi[8] = i22 * i22;
}
template <class PO>
SCVX_HD void inv2(double a, double b, double d, PO Mi) {  // [[a b],[b d]] SPD
    const double l00 = sqrt(a), l10 = b / l00, l11 = sqrt(piv_floor(d - l10 * l10, d));
    const double i00 = 1.0 / l00, i11 = 1.0 / l11, i10 = -l10 * i00 * i11;
    Mi[0] = i00 * i00 + i10 * i10; Mi[1] = Mi[2] = i10 * i11; Mi[3] = i11 * i11;
}

// ---- node blocks in SQUARE-ROOT form -------------------------------------------------------------------------------
// A node block of Hb is  M = a I + sum_c Jc' W_c^-2 Jc  over the cones that touch the node's variables.  Near the end of a
// solve an active cone contributes entries of size v0^4 / beta^2 ~ 1e15 while the block's small eigenvalues stay O(1): assembled
// and factorised as a matrix (cond ~ 1e16) the small pivots of a Cholesky factorisation are differences of those huge entries
// and come out as noise -- for the two cones whose head row is a VARIABLE (glideslope: r1 / tan(gamma); gimbal: u1 / cos(delta))
// the assembled form even contains an indefinite part (b2 (I - icos^2 e1 e1')) that only the rank-one term makes definite.
// That inconsistency between Hb^-1 and the operator form J' W^-1 W^-1 J was the first-row residual 1e5 that no refinement
// pass could contract (a trajectory frozen with SCVX_ST_SOLVER; profiles/r02_k4_fuzz.md rows 13/16/22/23).
//   * blocks of the form  a I + kappa v v'  (rate, tilt, dynamic pressure, fins, Tmax) are inverted in closed form, every entry
//     a sum of positive terms;
//   * the r block and the thrust block are never assembled: M = A'A with A = [sqrt(a) I; W_c^-1 Jc; ...] (6 or 9 rows), and
//     M^-1 = R^-1 R^-T from the Householder QR of A -- the conditioning of A is the square root of M's.
// inverse of a I + kappa v v' (3x3, row-major out)
template <class PO>
SCVX_HD void inv_iso_rank1_3(double a, double kappa, double v0, double v1, double v2, PO Mi) {
    const double n2 = v0 * v0 + v1 * v1 + v2 * v2;
    const double den = 1.0 / (a + kappa * n2), ia = 1.0 / a;
    const double off = -kappa * ia * den;
    Mi[0] = (a + kappa * (v1 * v1 + v2 * v2)) * ia * den;
    Mi[4] = (a + kappa * (v0 * v0 + v2 * v2)) * ia * den;
    Mi[8] = (a + kappa * (v0 * v0 + v1 * v1)) * ia * den;
    Mi[1] = Mi[3] = off * v0 * v1;
    Mi[2] = Mi[6] = off * v0 * v2;
    Mi[5] = Mi[7] = off * v1 * v2;
}
template <class PO>
SCVX_HD void inv_iso_rank1_2(double a, double kappa, double v0, double v1, PO Mi) {
    const double den = 1.0 / (a + kappa * (v0 * v0 + v1 * v1)), ia = 1.0 / a;
    Mi[0] = (a + kappa * v1 * v1) * ia * den;
    Mi[3] = (a + kappa * v0 * v0) * ia * den;
    Mi[1] = Mi[2] = -kappa * v0 * v1 * ia * den;
}
// row r of W^-1 Jc for a cone of dimension D whose scaling is (v, beta): x = column of Jc (D entries) -> W^-1 x
template <int D>
SCVX_HD void soc_Winv_col(const double* v, double ibeta, const double* x, double* y) {
    double vx = v[0] * x[0];
    for (int i = 1; i < D; i++) vx -= v[i] * x[i];
    y[0] = (2.0 * vx * v[0] - x[0]) * ibeta;
    for (int i = 1; i < D; i++) y[i] = (-2.0 * vx * v[i] + x[i]) * ibeta;
}
// (A'A)^-1 for A = NR x 3 (row-major, destroyed): Householder QR, then R^-1 R^-T.  Mi row-major 3x3.
template <int NR, class PO>
SCVX_HD void qr_inv3(double (&A)[NR][3], PO Mi) {
    double R[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int j = 0; j < 3; j++) {
        double nrm2 = 0;
        for (int r = j; r < NR; r++) nrm2 += A[r][j] * A[r][j];
        const double nrm = sqrt(nrm2);
        const double x0 = A[j][j];
        const double alpha = x0 > 0.0 ? -nrm : nrm;
        R[j][j] = alpha;
        const double v0 = x0 - alpha;                       // v = x - alpha e1 (no cancellation: alpha has the opposite sign)
        const double vtv = nrm2 - x0 * x0 + v0 * v0;
        if (vtv > 0.0) {
            const double tv = 2.0 / vtv;
            for (int c = j + 1; c < 3; c++) {
                double d = v0 * A[j][c];
                for (int r = j + 1; r < NR; r++) d += A[r][j] * A[r][c];
                d *= tv;
                R[j][c] = A[j][c] - d * v0;
                for (int r = j + 1; r < NR; r++) A[r][c] -= d * A[r][j];
            }
        } else {
            for (int c = j + 1; c < 3; c++) R[j][c] = A[j][c];
        }
    }
    // guard: a zero column cannot occur (sqrt(a) I is part of A), but keep the pivots finite
    for (int j = 0; j < 3; j++) if (!(fabs(R[j][j]) > 1e-300)) R[j][j] = 1e-300;
    const double i00 = 1.0 / R[0][0], i11 = 1.0 / R[1][1], i22 = 1.0 / R[2][2];
    const double i01 = -R[0][1] * i00 * i11;
    const double i12 = -R[1][2] * i11 * i22;
    const double i02 = -(R[0][1] * i12 + R[0][2] * i22) * i00;
    // Mi = Ri Ri'  (Ri upper triangular)
    Mi[0] = i00 * i00 + i01 * i01 + i02 * i02;
    Mi[1] = Mi[3] = i01 * i11 + i02 * i12;
    Mi[2] = Mi[6] = i02 * i22;
    Mi[4] = i11 * i11 + i12 * i12;
    Mi[5] = Mi[7] = i12 * i22;
    Mi[8] = i22 * i22;
}

template <class PH>
SCVX_HD double hxi_entry(PH h, int a, int b) {
    if (a == 0) return b == 0 ? h[HX_M] : 0.0;
    if (a < 4) return (b >= 1 && b < 4) ? h[HX_R + 3 * (a - 1) + (b - 1)] : 0.0;
    if (a < 7) return (b >= 4 && b < 7) ? h[HX_V + 3 * (a - 4) + (b - 4)] : 0.0;
    if (a < 9) return a == b ? h[HX_Q] : 0.0;
    if (a < 11) return (b >= 9 && b < 11) ? h[HX_Q34 + 2 * (a - 9) + (b - 9)] : 0.0;
    return (b >= 11) ? h[HX_W + 3 * (a - 11) + (b - 11)] : 0.0;
}
// y = Hxi * x for one node (14-vectors)
template <class PH>
SCVX_HD void hxi_apply(PH h, const double* x, double* y) {
    y[0] = h[HX_M] * x[0];
    for (int i = 0; i < 3; i++) y[1 + i] = h[HX_R + 3 * i] * x[1] + h[HX_R + 3 * i + 1] * x[2] + h[HX_R + 3 * i + 2] * x[3];
    for (int i = 0; i < 3; i++) y[4 + i] = h[HX_V + 3 * i] * x[4] + h[HX_V + 3 * i + 1] * x[5] + h[HX_V + 3 * i + 2] * x[6];
    y[7] = h[HX_Q] * x[7]; y[8] = h[HX_Q] * x[8];
    y[9] = h[HX_Q34] * x[9] + h[HX_Q34 + 1] * x[10];
    y[10] = h[HX_Q34 + 2] * x[9] + h[HX_Q34 + 3] * x[10];
    for (int i = 0; i < 3; i++) y[11 + i] = h[HX_W + 3 * i] * x[11] + h[HX_W + 3 * i + 1] * x[12] + h[HX_W + 3 * i + 2] * x[13];
}
// Dense position (14 i + j) of entry l of the compact node inverse in the 14x14 tile Hxi; l = HX_SZ is the second copy of the
// (q0, q1) diagonal value HX_Q at (8, 8).  The other 162 entries of the tile are structural zeros: the factorisation loops zero the
// tile once and scatter these 34 per segment (round 5; hxi_entry per element -- six data-dependent branches for each of 196
// elements -- was 2.9 k of the loop's 24 k cycles per segment).
SCVX_HD int hx_dense_pos(int l) {
    if (l >= HX_SZ) return 8 * 15;
    if (l >= HX_W) { const int q = l - HX_W; return 14 * (11 + q / 3) + 11 + q % 3; }
    if (l >= HX_Q34) { const int q = l - HX_Q34; return 14 * (9 + q / 2) + 9 + q % 2; }
    if (l == HX_Q) return 7 * 15;
    if (l >= HX_V) { const int q = l - HX_V; return 14 * (4 + q / 3) + 4 + q % 3; }
    if (l >= HX_R) { const int q = l - HX_R; return 14 * (1 + q / 3) + 1 + q % 3; }
    return 0;
}
// ------------------------------------------------------------------------------------------------
// the solver
// ------------------------------------------------------------------------------------------------
struct Result {
    int status;  // see the status table at the top of this file
    int iters;
    double merit, pobj;
    int warmed;   // the solve started from the kept iterate of the previous solve (SCVX_WARM_SAVE)
    int attempts; // step rules tried (1 unless the first ended on the numerical floor: Solver::solve)
};

// FStor: element type of the block-tridiagonal FACTOR (packed L_k^-1 and the coupling tiles N_k: 301 numbers per segment, read
// four times per interior-point iteration).  float there makes every banded solve a preconditioner of accuracy ~6e-8 cond(S)
// under the double-precision operator refinement of newton_solve (measured in profiles/r04_factor_f32.md).
template <class Ex, class Stor = double, class DStor = Stor, int NU = 3, class FStor = SCVX_FACTOR_T>
struct Solver {
    static_assert(NU == 3 || NU == 5, "control_dim 3 (the reference's live model) or 5 (fin extension)");
    static constexpr int NP = 14 + 2 * NU + 1;      // columns of a derivative tile: [A | B- | B+ | Sigma]
    static constexpr int DSZ = 14 * NP;             // doubles per tile (column-major 14 x NP)
    static constexpr int CS = 14 + 2 * NU;          // the sigma column
    static constexpr int NXU = 14 + NU;             // trust-region rows per node
    static constexpr int HU_SZ = hu_size(NU);       // node u-block inverse: 3x3 thrust (+ 2x2 fin)
    static constexpr int NODE_SZ = HX_SZ + HU_SZ;   // compact node inverse
    static constexpr int TW = 14 + 2 * NU;          // columns of the [TA | TBm | TBp] tile
    static constexpr int TS = NU == 5 ? 26 : 22;    // its row stride in LDS (conflict-free fragment reads)
    static constexpr int BPN = 14 * NU;             // a B+ / B- block (column-major 14 x NU)
    static constexpr int NPW = (DSZ + 63) / 64;     // registers per lane that hold one tile in a 64-lane wavefront
    typedef typename gp<Stor>::ptr gptr;     // workspace (storage type)
    typedef typename gp<Stor>::cptr cgptr;
    typedef typename gp<DStor>::ptr dptr;
    typedef typename gp<DStor>::cptr dcptr;   // the linearisation D as the discretisation kernel wrote it (double, or float
                                              // behind scvx_batch_set_linearization_f32: every load widens, arithmetic stays double)
    typedef typename gp<double>::cptr cdptr;  // the SCvx iterate: always double
    typedef typename gp<FStor>::ptr fptr;     // the factor (Linv, Nf)
    typedef typename gp<FStor>::cptr cfptr;
    Ex& ex;
    const Consts& C;
    Layout L;
    // inputs
    cdptr xbar, ubar, endpoint;
    dcptr D;
    double rk;
    // workspace
    gptr dk, V, rx, gx, dw, r1, cw, Vbest, tmpv;
    gptr y, ry, dy, r2, cy, tmpy, tmpy2, rp, tq0, tq1;
    gptr S, Z, lam, Wv, Wibz, tmpc, Wirz, sd;
    gptr Wbeta;
    gptr hx, hu;
    fptr Linv, Nf;
    gptr tchain;
    dptr At;   // A_k' copies, in the element type of D (a float D uses half of the slot)
    gptr ys, ytr, ynu;   // the three border multipliers  S y = Sg, E Hb^-1 Ptr, hnui Pnu
    gptr ptl, rtr;       // ptl = Hb^-1 Ptr (local, zero on nu),  rtr = E ptl (formed only for the y-space border of the two-ended factorisation)
    gptr tmpl, tmpl2;
    gptr uhat, lb0;
    gptr wh, Vw, yw, Sw, Zw;   // warm-start iterate of the last solve (wh[0] = 1: valid), see SCVX_WARM_SAVE
    // per-factorisation scalars
    double h_tr[4], h_nu[4], hrk, hnui;
    double q_tr[2], q_nu[2], q_sg[4], h00s;   // (v0, |v1|^2) of the two big cones; (v0, v1, b2, h01 v1 / h00) of the 2-cone (ts; s)
    double css, cst, csn, cts, ctt, ctn, cns, cnt_, pny;
    // step-rule settings of the current attempt (Solver::solve's ladder; attempt 0 = the SCVX_* defaults above)
    double p_step_frac = SCVX_STEP_FRAC, p_init_shift = SCVX_INIT_SHIFT, p_init_balance = SCVX_INIT_BALANCE;
    bool p_sigma_cube = false;
    double p_mu_floor = 0.25;   // SCVX_MU_FLOOR (defined where it is used, in attempt_solve)
    double bigvz[2];  // <v, W dz>_1 of the two big cones (corr_dir_pass -> update_pass)
    // Body sums of the two big cones' NEW iterate, taken by update_pass while it writes it: <s,s>, <z,z>, <s,z>, <s,r>, <z,r>, <r,r> with
    // r = s - a(V).  The next scale_pass forms every scalar of the scaling from them (its two reduction sweeps over s, z, V are not
    // run); bigs_ok says the sums describe S / Z / V as they stand (cleared when a solve starts).
    double bigs[2][6];
    bool bigs_ok;
    double bigvv[2];  // |v1|^2 of the two big cones' scaling vectors (scale_pass / identity_scaling -> build_kkt: no sweep of its own)
    double bigq[2][6]; // body sums of the two big cones from the predictor's direction pass: <l,l>, <l,a>, <a,a>, <v,l>, <v,wr>, <v,a>
                       // (a = W^-1 ds_aff): every reduction corr_rhs_pass needs is a combination of these (no reduction sweep of its own)
    double res_nrx2, res_nry2, res_sgy;   // build_kkt(res): |rx|^2 over the local rows, |ry|^2, Sg . y
    // BORDER IN t-SPACE (round 5).  With S = L L' every border coefficient <r_a, S^-1 r_b> is the inner product <t_a, t_b> of the
    // FORWARD-substituted right-hand sides t = L^-1 r, which the factorisation loop has at hand: it accumulates their 4 x 4 Gram matrix
    // (columns: Sg | rtr | hnui Pnu | the predictor's right-hand side) on one MFMA accumulator.  The three border systems are never
    // back-substituted: ys / ytr / ynu keep t_s, t_tr, t_nu, a solve forward-substitutes its own right-hand side, takes three inner
    // products with them, solves the 3 x 3 border, combines IN t-SPACE and back-substitutes once.  Per iteration: one backward
    // sweep over the factor with 4 right-hand sides, the border-coefficient pass and two y-space passes per solve are gone.
    bool tsp;           // this factorisation's border is kept in t-space (every form but the two-ended factorisation)
    double gram[16];    // <t_a, t_b>, a, b in {s, tr, nu, pred}
    double gn_pred;     // <Pnu, gx_nu> of the predictor's right-hand side
    double bigvx2[2]; // v0 wij_0 - <v, wij>_1 of the two big cones, wij = W^-1 J dw (dir_pass<false, true> -> cone_map_t(hbig))
#if defined(SCVX_IPM_PROF)
    double prof[32];   // in-kernel section timers (diagnostic builds)
#endif
    double cur_gate;   // max(pres, relgap) of the current iterate: refinement only pays in the endgame (dres is left out:
                       // an inaccurate solve RAISES it, and must not switch the refinement off)

    SCVX_HD Solver(Ex& e, const Consts& c) : ex(e), C(c) {
        L.init(c.K, c.vmax > 0.0, NU);
#if defined(SCVX_IPM_PROF)
        for (int i = 0; i < 32; i++) prof[i] = 0.0;
#endif
    }

    SCVX_HD void carve(gptr w) {
        const int nv = L.nv, ny = L.ny, nc = L.nc, nloc = L.nloc, K = L.K;
        dk = w; w += ny;
        V = w; w += nv; rx = w; w += nv; gx = w; w += nv; dw = w; w += nv; r1 = w; w += nv; cw = w; w += nv;
        Vbest = w; w += nv; tmpv = w; w += nv;
        y = w; w += ny; ry = w; w += ny; dy = w; w += ny; r2 = w; w += ny; cy = w; w += ny; tmpy = w; w += ny; tmpy2 = w; w += ny; rp = w; w += ny; tq0 = w; w += ny; tq1 = w; w += ny;
        S = w; w += nc; Z = w; w += nc; lam = w; w += nc; Wv = w; w += nc;
        Wibz = w; w += nc; tmpc = w; w += nc; Wirz = w; w += nc; sd = w; w += nc;
        Wbeta = w; w += L.ncones;
        hx = w; w += (size_t)(K + 1) * HX_SZ; hu = w; w += (size_t)(K + 1) * HU_SZ;
        Linv = (fptr)w; w += (size_t)K * LINV_SZ; Nf = (fptr)w; w += (size_t)K * 196; At = (dptr)w; w += (size_t)K * 196;   // a float factor uses half of its slots
        tchain = w; w += ny;
        ys = w; w += ny; ytr = w; w += ny; ynu = w; w += ny; rtr = w; w += ny; ptl = w; w += nloc;
        tmpl = w; w += nloc; tmpl2 = w; w += nloc;
        uhat = w; w += 3 * (K + 1); lb0 = w; w += (K + 1);
        w += 64;   // scalars (unused slots kept for layout stability)
        wh = w; w += 8; Vw = w; w += nv; yw = w; w += ny; Sw = w; w += nc; Zw = w; w += nc;
    }

    // ---- fixed-component masks (rocketland.jl:109-115) ----
    SCVX_HD bool fixed_x(int k, int j) const {
        if (k == 0) return !(j >= 7 && j < 11);
        if (k == L.K) return j != 0;
        return false;
    }
    SCVX_HD bool fixed_u(int k, int c) const { return k == L.K && (c == 1 || c == 2); }   // u[2:3, K+1] = 0 (rocketland.jl:115)

    // ---- node u-block inverse Hui (compact: 3x3 thrust block row-major at h[0..8], 2x2 fin block at h[9..12]) ----
    // element (i, c) of  B Hui  for the 14 x NU block of the tile Dt that starts at column c0
    template <class PH>
    static SCVX_HD double bhu(const double* Dt, int c0, int i, int c, PH h) {
        if (NU == 3 || c < 3) return Dt[14 * c0 + i] * h[c] + Dt[14 * (c0 + 1) + i] * h[3 + c] + Dt[14 * (c0 + 2) + i] * h[6 + c];
        return Dt[14 * (c0 + 3) + i] * h[9 + (c - 3)] + Dt[14 * (c0 + 4) + i] * h[11 + (c - 3)];
    }
    // So(i, j) = -TA(i, j) + sum_c TBm(i, c) Bp(j, c)   (T row stride TS, Bp column-major 14 x NU)
    static SCVX_HD double so_elem(const double* T, const double* Bp, int i, int j) {
        double a = -T[TS * i + j];
        SCVX_UNROLL
        for (int c = 0; c < NU; c++) a += T[TS * i + 14 + c] * Bp[14 * c + j];
        return a;
    }

    // ---- parallel vector helpers ----
    // Streaming loop over [i0, n) with U independent elements in flight per lane: ld(i) gathers the inputs of element
    // i (returned by value), st(i, v) computes and stores.  All loads of a batch are issued before its first store, so
    // the memory latency is paid once per U elements instead of once per element (the solver state streams from HBM;
    // a plain `for` makes the compiler wait on every element because the stores may alias the next loads).
    // Element order per lane is ascending, so reductions accumulated in st() keep their summation order.
    template <int U = SCVX_STREAM_U, class LD, class ST>
    SCVX_HD void stream(int i0, int n, LD&& ld, ST&& st) {
        const int nl = ex.nlanes();
        int i = i0 + ex.lane();
        for (; i + (U - 1) * nl < n; i += U * nl) {
            decltype(ld(0)) v[U];
            SCVX_UNROLL
            for (int q = 0; q < U; q++) v[q] = ld(i + q * nl);
            SCVX_UNROLL
            for (int q = 0; q < U; q++) st(i + q * nl, v[q]);
        }
        for (; i < n; i += nl) st(i, ld(i));
    }
    struct D2 { double a, b; };
    struct D3 { double a, b, c; };
    struct D4 { double a, b, c, d; };
    struct D5 { double a, b, c, d, e; };

    SCVX_HD double dot(cgptr a, cgptr b, int n) {
        double s = 0;
        stream(0, n, [&](int i) { return D2{a[i], b[i]}; }, [&](int, const D2& v) { s += v.a * v.b; });
        return ex.sum(s);
    }
    SCVX_HD double sumsq(cgptr a, int n) {
        double s = 0;
        stream<8>(0, n, [&](int i) { return a[i]; }, [&](int, double v) { s += v * v; });
        return ex.sum(s);
    }
    SCVX_HD void zero(gptr o, int n) {
        for (int i = ex.lane(); i < n; i += ex.nlanes()) o[i] = 0.0;
        ex.sync();
    }
    SCVX_HD void copy(gptr o, cgptr a, int n) {
        stream<8>(0, n, [&](int i) { return a[i]; }, [&](int i, double v) { o[i] = v; });
        ex.sync();
    }

    // ---- E (linearised dynamics rows, rocketland.jl:117-133) ----
    // out[k][i] = sum_j D_k[i][j] [dx_k; du_k; du_{k+1}; s]_j + nu_k[i] - dx_{k+1}[i]   (with_s: include the s column)
    //             + sa * add[k][i]  (add may be null);  returns |out|^2
    SCVX_HD_NI double E_apply(cgptr v, gptr out, bool with_s, cgptr add = nullptr, double sa = 1.0) {
        SCVX_THIS_LDS();
        SCVX_T0();
        const int K = L.K;
        const double s = with_s ? v[L.iS] : 0.0;
        double n2 = 0;
        for (int r = ex.lane(); r < 14 * K; r += ex.nlanes()) {
            const int k = r / 14, i = r - 14 * k;
            dcptr Dk = D + (size_t)k * DSZ + i;
            cgptr dx = v + 14 * k;
            cgptr du = v + L.nx + NU * k;
            const double ad = add ? add[r] : 0.0;   // issued with the batch of loads below, not after it
            double a = 0;
            for (int j = 0; j < 14; j++) a += Dk[14 * j] * dx[j];
            for (int j = 0; j < 2 * NU; j++) a += Dk[14 * (14 + j)] * du[j];  // du_k then du_{k+1} are adjacent
            a += Dk[14 * CS] * s;
            a += v[L.nx + L.nu_ + r] - v[14 * (k + 1) + i];
            a += sa * ad;
            out[r] = a;
            n2 += a * a;
        }
        ex.sync();
        SCVX_T1(1);
        return ex.sum(n2);
    }
    // g = E_loc' y on the local part (dx, du, nu); returns Sg . y (the s entry) to every lane.
    // mode 1: g = base - E_loc' y;  mode 2: g = E_loc' y - base  (base may be g itself: each entry is read and written by
    // the same lane)
    // pc / pn (mode 1 only): base - pc Ptr - pn Pnu - E_loc' y, the right-hand side of a solve's final Hb^-1 (kkt_solve)
    // eout (optional): the raw products E_loc' y on the (dx, du) rows as well (the refinement's operator-form check reuses them instead
    // of a second pass over D: newton_corr)
    // n2out (mode 2, the dual residual rx = c + E'y - J'Z): the cost entry (-1 on the final mass) is added, the fixed rows
    // (rocketland.jl:109-115) are written as zeros and |g|^2 over the local rows is returned through it -- no mask / norm passes after
    SCVX_HD_NI double Et_apply(cgptr yy, gptr g, cgptr base = nullptr, int mode = 0, double pc = 0.0, double pn = 0.0, gptr eout = nullptr,
                               double* n2out = nullptr) {
        SCVX_THIS_LDS();
        SCVX_T0();
        const int K = L.K;
        const bool corr = pc != 0.0 || pn != 0.0;
        const bool resn = n2out != nullptr;
        double n2 = 0.0;
        cgptr Pt = Wv + L.o_tr + 1; cgptr Pn = Wv + L.o_nu + 1;
        for (int t = ex.lane(); t < L.nx; t += ex.nlanes()) {
            const int k = t / 14, j = t - 14 * k;
            const double b0 = mode ? base[t] - (corr ? pc * Pt[t] : 0.0) : 0.0;   // loaded with the batch below
            double a = 0;
            if (k < K) {
                dcptr col = At + (size_t)k * 196 + j;   // A_k' row-major: lanes j read consecutive doubles
                cgptr yk = yy + 14 * k;
                for (int i = 0; i < 14; i++) a += col[14 * i] * yk[i];
            }
            if (k > 0) a -= yy[14 * (k - 1) + j];
            if (eout) eout[t] = a;
            if (mode) a = mode == 1 ? b0 - a : a - b0;
            if (resn) {
                if (t == 14 * K) a += -1.0;
                if ((k == 0 && fixed_x(0, j)) || (k == K && fixed_x(K, j))) a = 0.0;
                n2 += a * a;
            }
            g[t] = a;
        }
        for (int t = ex.lane(); t < L.nu_; t += ex.nlanes()) {
            const int k = t / NU, c = t - NU * k;
            const double b0 = mode ? base[L.nx + t] - (corr ? pc * Pt[L.nx + t] : 0.0) : 0.0;
            double a = 0;
            if (k < K) {
                dcptr col = D + (size_t)k * DSZ + 14 * (14 + c);
                cgptr yk = yy + 14 * k;
                for (int i = 0; i < 14; i++) a += col[i] * yk[i];
            }
            if (k > 0) {
                dcptr col = D + (size_t)(k - 1) * DSZ + 14 * (14 + NU + c);
                cgptr yk = yy + 14 * (k - 1);
                for (int i = 0; i < 14; i++) a += col[i] * yk[i];
            }
            if (eout) eout[L.nx + t] = a;
            if (mode) a = mode == 1 ? b0 - a : a - b0;
            if (resn) {
                if (k == K && fixed_u(K, c)) a = 0.0;
                n2 += a * a;
            }
            g[L.nx + t] = a;
        }
        double sg = 0;
        {
            const dcptr D_ = D; cgptr bn = mode ? base + L.nx + L.nu_ : yy; const gptr gn = g + L.nx + L.nu_;
            if (corr)
                stream(0, 14 * K, [&](int r) { const int k = r / 14, i = r - 14 * k; return D4{yy[r], bn[r], D_[(size_t)k * DSZ + 14 * CS + i], Pn[r]}; },
                       [&](int r, const D4& w) { gn[r] = (w.b - pn * w.d) - w.a; sg += w.c * w.a; });
            else
            stream(0, 14 * K, [&](int r) { const int k = r / 14, i = r - 14 * k; return D3{yy[r], bn[r], D_[(size_t)k * DSZ + 14 * CS + i]}; },
                   [&](int r, const D3& w) { const double o = mode == 0 ? w.a : (mode == 1 ? w.b - w.a : w.a - w.b); gn[r] = o; n2 += o * o; sg += w.c * w.a; });
        }
        ex.sync();
        SCVX_T1(20);
        if (resn) *n2out = ex.sum(n2);
        return ex.sum(sg);
    }

    // ---- cone maps: a(w), J dw, J' z ----
    // out = a(v) if affine else J v
    SCVX_HD_NI void cone_map(cgptr v, gptr out, bool affine) {
        SCVX_THIS_LDS();
        SCVX_T0();
        const int K = L.K;
        const double af = affine ? 1.0 : 0.0;
        for (int k = ex.lane(); k <= K; k += ex.nlanes()) {
            cgptr dx = v + 14 * k;
            cgptr du = v + L.nx + NU * k;
            double x[14], u[NU];
            for (int j = 0; j < 14; j++) x[j] = af * xbar[14 * k + j] + dx[j];
            for (int c = 0; c < NU; c++) u[c] = af * ubar[NU * k + c] + du[c];
            if (k < K) {
                gptr g = out + L.o_gs + 3 * k;
                g[0] = x[1] * C.itan; g[1] = x[2]; g[2] = x[3];
                gptr t = out + L.o_tilt + 3 * k;
                t[0] = af * C.sqcm; t[1] = x[9]; t[2] = x[10];
                gptr r = out + L.o_rate + 4 * k;
                r[0] = af * C.omMax; r[1] = x[11]; r[2] = x[12]; r[3] = x[13];
                if (k < L.ndp) {
                    gptr d = out + L.o_dp + 4 * k;
                    d[0] = af * C.vmax; d[1] = x[4]; d[2] = x[5]; d[3] = x[6];
                }
            }
            if (k >= 1) out[L.o_mass + (k - 1)] = x[0] - af * C.mdry;
            gptr tb = out + L.o_tb + 4 * k;
            tb[0] = af * C.Tmax; tb[1] = u[0]; tb[2] = u[1]; tb[3] = u[2];
            gptr tc = out + L.o_tc + 4 * k;
            tc[0] = u[0] * C.icos; tc[1] = u[0]; tc[2] = u[1]; tc[3] = u[2];
            out[L.o_lb + k] = uhat[3 * k] * du[0] + uhat[3 * k + 1] * du[1] + uhat[3 * k + 2] * du[2] - af * lb0[k];
            if constexpr (NU == 5) {
                gptr f = out + L.o_fin + 3 * k;
                f[0] = af * C.finmxf; f[1] = u[3]; f[2] = u[4];
            }
        }
        {
            cgptr vn = v + L.nx + L.nu_; gptr on = out + L.o_nu + 1; gptr ot = out + L.o_tr + 1;
            stream<8>(0, 14 * K, [&](int i) { return vn[i]; }, [&](int i, double x) { on[i] = x; });
            stream<8>(0, L.nx + L.nu_, [&](int i) { return v[i]; }, [&](int i, double x) { ot[i] = x; });
        }
        if (ex.lane() == 0) {
            out[L.o_nu] = v[L.iTNU];
            out[L.o_tr] = v[L.iTTR];
            out[L.o_sg] = v[L.iTS];
            out[L.o_sg + 1] = v[L.iS];
            out[L.o_rk] = af * rk - v[L.iTTR];
        }
        ex.sync();
        SCVX_T1(21);
    }
    // g = J' z (var-shaped, all nv entries written), or g = -sub - J' z when sub is given (the Newton right-hand side)
    // hbig: the bodies of the two big cones in z hold wij = W^-1 (J dw) (dir_pass<false, true>); their second scaling
    // W^-1 wij = (-2 vx2 v_i + wij_i) / beta is applied here, on the fly (the heads in z are final)
    // neg (without sub): g = -J' z
    // mask: the fixed rows (rocketland.jl:109-115) are written as zeros (what a mask_fixed pass after it would do)
    SCVX_HD_NI void cone_map_t(cgptr z, gptr g, cgptr sub = nullptr, bool hbig = false, bool neg = false, bool mask = false) {
        SCVX_THIS_LDS();
        SCVX_T0();
        const int K = L.K;
        const double ht_c = hbig ? -2.0 * bigvx2[1] : 0.0, ht_s = hbig ? 1.0 / Wbeta[L.c_tr] : 1.0;
        const double hn_c = hbig ? -2.0 * bigvx2[0] : 0.0, hn_s = hbig ? 1.0 / Wbeta[L.c_nu] : 1.0;
        cgptr vtr = Wv + L.o_tr + 1;
        for (int k = ex.lane(); k <= K; k += ex.nlanes()) {
            double gl[NXU];   // the node's rows are assembled in registers and stored once
            cgptr trx = z + L.o_tr + 1 + 14 * k;
            if (hbig) { for (int j = 0; j < 14; j++) gl[j] = (ht_c * vtr[14 * k + j] + trx[j]) * ht_s; }
            else
            for (int j = 0; j < 14; j++) gl[j] = trx[j];
            if (k >= 1) gl[0] += z[L.o_mass + (k - 1)];
            if (k < K) {
                cgptr gs = z + L.o_gs + 3 * k;
                gl[1] += gs[0] * C.itan; gl[2] += gs[1]; gl[3] += gs[2];
                cgptr t = z + L.o_tilt + 3 * k;
                gl[9] += t[1]; gl[10] += t[2];
                cgptr r = z + L.o_rate + 4 * k;
                gl[11] += r[1]; gl[12] += r[2]; gl[13] += r[3];
                if (k < L.ndp) {
                    cgptr d = z + L.o_dp + 4 * k;
                    gl[4] += d[1]; gl[5] += d[2]; gl[6] += d[3];
                }
            }
            cgptr tru = z + L.o_tr + 1 + L.nx + NU * k;
            cgptr tb = z + L.o_tb + 4 * k;
            cgptr tc = z + L.o_tc + 4 * k;
            const double zl = z[L.o_lb + k];
            double tu[NU];
            for (int c = 0; c < NU; c++) tu[c] = hbig ? (ht_c * vtr[L.nx + NU * k + c] + tru[c]) * ht_s : (double)tru[c];
            for (int c = 0; c < 3; c++) gl[14 + c] = tu[c] + tb[1 + c] + tc[1 + c] + zl * uhat[3 * k + c];
            gl[14] += tc[0] * C.icos;
            if constexpr (NU == 5) {
                cgptr f = z + L.o_fin + 3 * k;
                gl[17] = tu[3] + f[1]; gl[18] = tu[4] + f[2];
            }
            gptr gx_ = g + 14 * k; gptr gu = g + L.nx + NU * k;
            if (sub) {
                double sb[NXU];
                for (int j = 0; j < 14; j++) sb[j] = sub[14 * k + j];
                for (int c = 0; c < NU; c++) sb[14 + c] = sub[L.nx + NU * k + c];
                const bool m0 = mask && k == 0, mK = mask && k == K;
                for (int j = 0; j < 14; j++) gx_[j] = ((m0 && fixed_x(0, j)) || (mK && fixed_x(K, j))) ? 0.0 : -sb[j] - gl[j];
                for (int c = 0; c < NU; c++) gu[c] = (mK && fixed_u(K, c)) ? 0.0 : -sb[14 + c] - gl[14 + c];
            } else {
                const double sg = neg ? -1.0 : 1.0;
                for (int j = 0; j < 14; j++) gx_[j] = sg * gl[j];
                for (int c = 0; c < NU; c++) gu[c] = sg * gl[14 + c];
            }
        }
        {
            cgptr zn = z + L.o_nu + 1; gptr gn = g + L.nx + L.nu_;
            if (sub) {
                cgptr sn = sub + L.nx + L.nu_;
                stream(0, 14 * K, [&](int i) { return D2{zn[i], sn[i]}; }, [&](int i, const D2& x) { gn[i] = -x.b - x.a; });
            } else if (hbig) {
                cgptr vn = Wv + L.o_nu + 1;
                stream(0, 14 * K, [&](int i) { return D2{zn[i], vn[i]}; }, [&](int i, const D2& x) { gn[i] = (hn_c * x.b + x.a) * hn_s; });
            } else {
                const double sg = neg ? -1.0 : 1.0;
                stream<8>(0, 14 * K, [&](int i) { return zn[i]; }, [&](int i, double x) { gn[i] = sg * x; });
            }
        }
        if (ex.lane() == 0) {
            const double sg = neg ? -1.0 : 1.0;
            const double g0 = z[L.o_nu], g1 = z[L.o_tr] - z[L.o_rk], g2 = z[L.o_sg], g3 = z[L.o_sg + 1];
            g[L.iTNU] = sub ? -sub[L.iTNU] - g0 : sg * g0;
            g[L.iTTR] = sub ? -sub[L.iTTR] - g1 : sg * g1;
            g[L.iTS] = sub ? -sub[L.iTS] - g2 : sg * g2;
            g[L.iS] = sub ? -sub[L.iS] - g3 : sg * g3;
        }
        ex.sync();
        SCVX_T1(21);
    }
    SCVX_HD void mask_fixed(gptr g) {
        for (int j = ex.lane(); j < 14; j += ex.nlanes()) {
            if (fixed_x(0, j)) g[j] = 0.0;
            if (fixed_x(L.K, j)) g[14 * L.K + j] = 0.0;
        }
        if (ex.lane() == 0) { g[L.nx + NU * L.K + 1] = 0.0; g[L.nx + NU * L.K + 2] = 0.0; }
        ex.sync();
    }

    // ---- big-cone helpers (all lanes cooperate) ----
    SCVX_HD void big_W(int off, int dim, int cidx, cgptr x, gptr yv, bool inverse) {
        cgptr v = Wv + off;
        cgptr xo = x + off; gptr yo = yv + off;
        double vx = 0;
        stream(1, dim, [&](int i) { return D2{v[i], xo[i]}; }, [&](int, const D2& q) { vx += q.a * q.b; });
        vx = ex.sum(vx);
        vx = v[0] * x[off] + (inverse ? -vx : vx);
        const double beta = Wbeta[cidx];
        const double sc = inverse ? 1.0 / beta : beta;
        const double x0 = x[off];
        const double tw = (inverse ? -2.0 : 2.0) * vx;
        ex.sync();  // x may alias yv: every lane has read x[off] before lane 0 overwrites it
        stream(1, dim, [&](int i) { return D2{v[i], xo[i]}; }, [&](int i, const D2& q) { yo[i] = (tw * q.a + q.b) * sc; });
        if (ex.lane() == 0) yv[off] = (2.0 * vx * v[0] - x0) * sc;
    }
    // ---- uniform iteration over the small cones: one strided loop per (dimension, block), dimension known at compile
    // time so the per-cone code unrolls and its loads batch; f(std::integral_constant<int,D>, offset, cone index) ----
    template <int D, class F>
    SCVX_HD void each_small(int off0, int c0, int n, F&& f) {
        for (int q = ex.lane(); q < n; q += ex.nlanes()) f(std::integral_constant<int, D>(), off0 + D * q, c0 + q);
    }
    template <class F>
    SCVX_HD void all_small(F&& f, bool with_sg) {
        const int K = L.K;
        each_small<3>(L.o_gs, L.c_gs, 2 * K, f);            // gs, tilt are adjacent
        each_small<4>(L.o_rate, L.c_rate, K, f);
        each_small<1>(L.o_mass, L.c_mass, K, f);
        each_small<4>(L.o_tb, L.c_tb, 2 * (K + 1), f);      // tb, tc are adjacent
        each_small<1>(L.o_lb, L.c_lb, K + 1, f);
        each_small<3>(L.o_fin, L.c_fin, L.nfin, f);
        each_small<4>(L.o_dp, L.c_dp, L.ndp, f);
        if (with_sg) { each_small<2>(L.o_sg, L.c_sg, 1, f); each_small<1>(L.o_rk, L.c_rk, 1, f); }
    }

    SCVX_HD void identity_scaling() {
        for (int i = ex.lane(); i < L.nc; i += ex.nlanes()) Wv[i] = 0.0;
        ex.sync();
        all_small([&](auto, int off, int c) { Wv[off] = 1.0; Wbeta[c] = 1.0; }, false);
        if (ex.lane() == 0) {
            Wv[L.o_nu] = 1.0; Wv[L.o_tr] = 1.0; Wv[L.o_sg] = 1.0; Wv[L.o_rk] = 1.0;
            Wbeta[L.c_nu] = Wbeta[L.c_tr] = Wbeta[L.c_sg] = Wbeta[L.c_rk] = 1.0;
        }
        bigvv[0] = bigvv[1] = 0.0;
        ex.sync();
    }
    // out = W in  /  W^-1 in   (cone vectors; in may alias out)
    SCVX_HD_NI void W_all(cgptr in, gptr out, bool inverse) {
        SCVX_THIS_LDS();
        SCVX_T0();
        all_small([&](auto Dt_, int off, int c) {
            constexpr int d = decltype(Dt_)::value;
            const double beta = Wbeta[c];
            if (d == 1) out[off] = inverse ? in[off] / beta : in[off] * beta;
            else {
                double xv[d], vv[d], yv[d];
                for (int i = 0; i < d; i++) { xv[i] = in[off + i]; vv[i] = Wv[off + i]; }
                soc_W_small(vv, beta, d, xv, yv, inverse);
                for (int i = 0; i < d; i++) out[off + i] = yv[i];
            }
        }, true);
        big_W(L.o_nu, 14 * L.K + 1, L.c_nu, in, out, inverse);
        big_W(L.o_tr, NXU * (L.K + 1) + 1, L.c_tr, in, out, inverse);
        ex.sync();
        SCVX_T1(8);
    }
    // ------------------------------------------------------------------------------------------------
    // Fused cone passes.  The solver state streams from HBM, so one interior-point iteration is organised as five
    // sweeps over the cones, each reading what it needs once and writing only what a later sweep consumes:
    //     scale_pass      S, Z, V        -> Wv, Wbeta, lam, Wirz = W^-1 (S - a(V)), tmpc = W^-1 (lam - Wirz), S'Z, |rz|^2
    //     pred_dir_pass   dw             -> sd = W^-1 ds_aff (W dz_aff = -lam - sd is implied), affine step length
    //     corr_rhs_pass   lam, sd, sigmu -> Wibz, tmpc = W^-1 Wibz
    //     corr_dir_pass   dw             -> sd = W dz, step length, <v, W dz> of the big cones
    //     update_pass     alpha          -> Z += alpha W^-1 sd,  S += alpha (J dw - (S - a(V)))
    // A small cone (dimension <= 4) lives in the registers of one lane for the whole sweep; the two big trust-region
    // cones are swept cooperatively: reductions first (their scalars enter every element), then one apply sweep.
    // a(v) / J v of a small cone are gathered straight from the variable vector (no cone-shaped temporary).
    // ------------------------------------------------------------------------------------------------
    enum { G_GS3 = 0, G_RATE = 1, G_MASS = 2, G_T4 = 3, G_LB = 4, G_SG = 5, G_RK = 6, G_DP = 7, G_FIN = 8 };
    template <int GRP, int D, class F>
    SCVX_HD void each_small_g(int off0, int c0, int n, F&& f) {
        for (int q = ex.lane(); q < n; q += ex.nlanes())
            f(std::integral_constant<int, GRP>(), std::integral_constant<int, D>(), off0 + D * q, c0 + q, q);
    }
    // f(group tag, dimension tag, offset in the cone vector, cone index, index inside the group)
    template <class F>
    SCVX_HD void for_small(F&& f) {
        const int K = L.K;
        each_small_g<G_GS3, 3>(L.o_gs, L.c_gs, 2 * K, f);            // gs, tilt are adjacent
        each_small_g<G_RATE, 4>(L.o_rate, L.c_rate, K, f);
        each_small_g<G_MASS, 1>(L.o_mass, L.c_mass, K, f);
        each_small_g<G_T4, 4>(L.o_tb, L.c_tb, 2 * (K + 1), f);       // tb, tc are adjacent
        each_small_g<G_LB, 1>(L.o_lb, L.c_lb, K + 1, f);
        if constexpr (NU == 5) each_small_g<G_FIN, 3>(L.o_fin, L.c_fin, L.nfin, f);
        each_small_g<G_DP, 4>(L.o_dp, L.c_dp, L.ndp, f);
        each_small_g<G_SG, 2>(L.o_sg, L.c_sg, 1, f);
        each_small_g<G_RK, 1>(L.o_rk, L.c_rk, 1, f);
    }
    // one small cone of a(v) (af = 1) or J v (af = 0): the rows cone_map writes for it
    template <int GRP, int D>
    SCVX_HD void small_gather(int q, cgptr v, double af, double (&o)[D]) const {
        const int K = L.K;
        if constexpr (GRP == G_GS3) {
            const bool gs = q < K;
            const int k = gs ? q : q - K;
            cgptr dx = v + 14 * k; cdptr xb = xbar + 14 * k;
            const int i1 = gs ? 2 : 9, i2 = gs ? 3 : 10;
            const double x1 = af * xb[1] + dx[1];
            o[0] = gs ? x1 * C.itan : af * C.sqcm;
            o[1] = af * xb[i1] + dx[i1];
            o[2] = af * xb[i2] + dx[i2];
        } else if constexpr (GRP == G_RATE) {
            cgptr dx = v + 14 * q; cdptr xb = xbar + 14 * q;
            o[0] = af * C.omMax;
            for (int j = 0; j < 3; j++) o[1 + j] = af * xb[11 + j] + dx[11 + j];
        } else if constexpr (GRP == G_MASS) {
            const int k = q + 1;
            o[0] = (af * xbar[14 * k] + v[14 * k]) - af * C.mdry;
        } else if constexpr (GRP == G_T4) {
            const bool tb = q <= K;
            const int k = tb ? q : q - (K + 1);
            cgptr du = v + L.nx + NU * k; cdptr ub = ubar + NU * k;
            double u[3];
            for (int c = 0; c < 3; c++) u[c] = af * ub[c] + du[c];
            o[0] = tb ? af * C.Tmax : u[0] * C.icos;
            o[1] = u[0]; o[2] = u[1]; o[3] = u[2];
        } else if constexpr (GRP == G_LB) {
            cgptr du = v + L.nx + NU * q;
            o[0] = uhat[3 * q] * du[0] + uhat[3 * q + 1] * du[1] + uhat[3 * q + 2] * du[2] - af * lb0[q];
        } else if constexpr (GRP == G_FIN) {
            cgptr du = v + L.nx + NU * q; cdptr ub = ubar + NU * q;
            o[0] = af * C.finmxf;
            o[1] = af * ub[3] + du[3];
            o[2] = af * ub[4] + du[4];
        } else if constexpr (GRP == G_DP) {
            cgptr dx = v + 14 * q; cdptr xb = xbar + 14 * q;
            o[0] = af * C.vmax;
            for (int j = 0; j < 3; j++) o[1 + j] = af * xb[4 + j] + dx[4 + j];
        } else if constexpr (GRP == G_SG) {
            o[0] = v[L.iTS]; o[1] = v[L.iS];
        } else {
            o[0] = af * rk - v[L.iTTR];
        }
    }
    // the two big cones: offset / dimension / cone index in the cone vector; head and body of a(v) = J v in a var vector
    struct BigCone { int off, dim, cidx, head, body; };
    SCVX_HD BigCone big_cone(int q) const {
        return q == 0 ? BigCone{L.o_nu, 14 * L.K + 1, L.c_nu, L.iTNU, L.nx + L.nu_}
                      : BigCone{L.o_tr, NXU * (L.K + 1) + 1, L.c_tr, L.iTTR, 0};
    }
    struct D6 { double a, b, c, d, e, f; };

    // ---- sweep 1: scaling, residual, predictor right-hand side ----
    SCVX_HD_NI void scale_pass(double& gap_out, double& nrz2_out) {
        SCVX_THIS_LDS();
        SCVX_T0();
        double gap = 0, nrz2 = 0;
        const cgptr S_ = S; const cgptr Z_ = Z; const cgptr V_ = V;
        const gptr Wv_ = Wv; const gptr lam_ = lam; const gptr Wirz_ = Wirz; const gptr tmpc_ = tmpc; const gptr Wbeta_ = Wbeta;
        for_small([&](auto G_, auto Dt_, int off, int c, int q) {
            constexpr int GRP = decltype(G_)::value, D = decltype(Dt_)::value;
            double s[D], z[D], a[D], r[D];
            for (int i = 0; i < D; i++) { s[i] = S_[off + i]; z[i] = Z_[off + i]; }
            small_gather<GRP, D>(q, V_, 1.0, a);
            for (int i = 0; i < D; i++) { r[i] = s[i] - a[i]; gap += s[i] * z[i]; nrz2 += r[i] * r[i]; }
            if constexpr (D == 1) {
                const double beta = sqrt(s[0] / z[0]), l = sqrt(s[0] * z[0]);
                const double wr = r[0] / beta;
                Wbeta_[c] = beta; Wv_[off] = 1.0; lam_[off] = l; Wirz_[off] = wr; tmpc_[off] = (l - wr) / beta;
            } else {
                double beta, v[D], l[D], wr[D], m[D], t[D];
                soc_nt_small(s, z, D, v, beta);
                soc_W_small(v, beta, D, z, l, false);
                soc_W_small(v, beta, D, r, wr, true);
                for (int i = 0; i < D; i++) m[i] = l[i] - wr[i];
                soc_W_small(v, beta, D, m, t, true);
                Wbeta_[c] = beta;
                for (int i = 0; i < D; i++) { Wv_[off + i] = v[i]; lam_[off + i] = l[i]; Wirz_[off + i] = wr[i]; tmpc_[off + i] = t[i]; }
            }
        });
        for (int q = 0; q < 2; q++) {
            const BigCone bc = big_cone(q);
            const cgptr s = S_ + bc.off; const cgptr z = Z_ + bc.off; const cgptr ab = V_ + bc.body - 1;   // ab[i] = a_i, i >= 1
            double a = 0, b = 0, c = 0;
            const bool carried = SCVX_CARRY_BIGSUMS != 0 && bigs_ok;
            if (carried) { a = bigs[q][0]; b = bigs[q][1]; c = bigs[q][2]; }
            else {
                stream(1, bc.dim, [&](int i) { return D2{s[i], z[i]}; },
                       [&](int, const D2& v) { a += v.a * v.a; b += v.b * v.b; c += v.a * v.b; });
                a = ex.sum(a); b = ex.sum(b); c = ex.sum(c);
            }
            const double s0 = s[0], z0 = z[0], r0 = s0 - V_[bc.head];
            const double sj = sqrt(s0 * s0 - a), zj = sqrt(z0 * z0 - b);
            const double isj = 1.0 / sj, izj = 1.0 / zj;
            const double gam = sqrt(0.5 * (1.0 + (s0 * z0 + c) * isj * izj));
            const double ig = 0.5 / gam;
            const double wb0 = (s0 * isj + z0 * izj) * ig;
            const double den = 1.0 / sqrt(2.0 * (wb0 + 1.0));
            const double v0 = (wb0 + 1.0) * den, beta = sqrt(sj * izj), ibeta = 1.0 / beta;
            const double igd = ig * den;
            // second reduction: products with v (v_i = (s_i / sj - z_i / zj) ig den needs the scalars above)
            double vz = 0, vr = 0, vv = 0, g1 = 0, n1 = 0;
            if (carried) {
                // v_i = (s_i / sj - z_i / zj) ig den: its products with z, r and itself from the carried sums (at a near-complementary pair
                // <s,z> < 0 and every term below has one sign: nothing cancels)
                const double sr = bigs[q][3], zr = bigs[q][4];
                vz = (c * isj - b * izj) * igd;
                vr = (sr * isj - zr * izj) * igd;
                vv = ((a * isj) * isj - 2.0 * (c * isj) * izj + (b * izj) * izj) * (igd * igd);
                if (ex.lane() == 0) { gap += c; nrz2 += bigs[q][5]; }   // lane-local partials: counted once
            } else {
            stream(1, bc.dim, [&](int i) { return D3{s[i], z[i], ab[i]}; },
                   [&](int, const D3& w) {
                       const double vi = (w.a * isj - w.b * izj) * igd, ri = w.a - w.c;
                       vz += vi * w.b; vr += vi * ri; vv += vi * vi; g1 += w.a * w.b; n1 += ri * ri;
                   });
            vz = ex.sum(vz); vr = ex.sum(vr); vv = ex.sum(vv);
            }
            bigvv[q] = vv;
            gap += g1; nrz2 += n1;
            if (ex.lane() == 0) { gap += s0 * z0; nrz2 += r0 * r0; }
            const double vxz = v0 * z0 + vz;                       // lam = W z
            const double l0 = (2.0 * vxz * v0 - z0) * beta;
            const double vxr = v0 * r0 - vr;                       // Wirz = W^-1 rz
            const double wr0 = (2.0 * vxr * v0 - r0) * ibeta;
            const double m0 = l0 - wr0;                            // tmpc = W^-1 (lam - Wirz)
            const double vl = beta * (2.0 * vxz * vv + vz);        // <v, lam>_1
            const double vw = (-2.0 * vxr * vv + vr) * ibeta;      // <v, Wirz>_1
            const double vxm = v0 * m0 - (vl - vw);
            const double t0 = (2.0 * vxm * v0 - m0) * ibeta;
            const gptr vo = Wv_ + bc.off; const gptr lo = lam_ + bc.off; const gptr wo = Wirz_ + bc.off; const gptr to = tmpc_ + bc.off;
            stream(1, bc.dim, [&](int i) { return D3{s[i], z[i], ab[i]}; },
                   [&](int i, const D3& w) {
                       const double vi = (w.a * isj - w.b * izj) * igd, ri = w.a - w.c;
                       const double li = (2.0 * vxz * vi + w.b) * beta;
                       const double wi = (-2.0 * vxr * vi + ri) * ibeta;
                       vo[i] = vi; lo[i] = li; wo[i] = wi;
                       to[i] = (-2.0 * vxm * vi + (li - wi)) * ibeta;
                   });
            if (ex.lane() == 0) { vo[0] = v0; lo[0] = l0; wo[0] = wr0; to[0] = t0; Wbeta_[bc.cidx] = beta; }
        }
        ex.sync();
        gap_out = ex.sum(gap);
        nrz2_out = ex.sum(nrz2);
        SCVX_T1(11);
    }

    // ---- sweeps 2 and 4: scaled directions from the Newton step dw, step length to the cone boundary ----
    // wij = W^-1 J dw,  W^-1 ds = wij - Wirz,  W dz = -(wij + Wibz)   (predictor: Wibz = lam - Wirz, so W dz = -lam - W^-1 ds).
    // PRED: stores sd = W^-1 ds (the corrector's right-hand side needs it).  Otherwise stores sd = W dz (update_pass needs
    // it) and, for the big cones, <v, W dz>_1 in bigvz[].
    // CHECK (corrector only): also leaves u = W^-1 wij in tmpc for the refinement's operator-form residual H dw = J' W^-1 (W^-1 J dw)
    // -- the first scaling is computed here anyway.  Small cones and the heads of the big cones: final; bodies of the big cones: wij
    // itself, with bigvx2[] = v0 wij_0 - <v, wij>_1 (summed from the computed wij) for cone_map_t(hbig) to finish.
    template <bool PRED, bool CHECK = false>
    SCVX_HD_NI double dir_pass() {
        static_assert(!(PRED && CHECK), "the operator-form check belongs to the corrector");
        SCVX_THIS_LDS();
        SCVX_T0();
        double amax = INFINITY;
        const cgptr dw_ = dw; const cgptr Wv_ = Wv; const cgptr lam_ = lam; const cgptr Wirz_ = Wirz; const cgptr Wibz_ = Wibz;
        const cgptr Wbeta_ = Wbeta; const gptr sd_ = sd; const gptr tc_ = tmpc;
        for_small([&](auto G_, auto Dt_, int off, int c, int q) {
            constexpr int GRP = decltype(G_)::value, D = decltype(Dt_)::value;
            double ds[D], l[D], wr[D], a[D], b[D];
            small_gather<GRP, D>(q, dw_, 0.0, ds);
            const double beta = Wbeta_[c];
            for (int i = 0; i < D; i++) { l[i] = lam_[off + i]; wr[i] = Wirz_[off + i]; }
            if constexpr (D == 1) {
                const double wij = ds[0] / beta;
                if (CHECK) tc_[off] = wij / beta;
                a[0] = wij - wr[0];
                b[0] = PRED ? -l[0] - a[0] : -(wij + Wibz_[off]);
                const double sa = a[0] < 0.0 ? -l[0] / a[0] : INFINITY, sb = b[0] < 0.0 ? -l[0] / b[0] : INFINITY;
                if (sa < amax) amax = sa;
                if (sb < amax) amax = sb;
            } else {
                double v[D], wij[D];
                for (int i = 0; i < D; i++) v[i] = Wv_[off + i];
                soc_W_small(v, beta, D, ds, wij, true);
                if (CHECK) {
                    double uu[D];
                    soc_W_small(v, beta, D, wij, uu, true);
                    for (int i = 0; i < D; i++) tc_[off + i] = uu[i];
                }
                for (int i = 0; i < D; i++) { a[i] = wij[i] - wr[i]; b[i] = PRED ? -l[i] - a[i] : -(wij[i] + Wibz_[off + i]); }
                double ll = 0, la = 0, aa = 0, lb = 0, bb = 0;
                for (int i = 1; i < D; i++) { ll += l[i] * l[i]; la += l[i] * a[i]; aa += a[i] * a[i]; lb += l[i] * b[i]; bb += b[i] * b[i]; }
                const double sa = soc_maxstep_parts(l[0], a[0], l[0] * l[0] - ll, l[0] * a[0] - la, a[0] * a[0] - aa);
                const double sb = soc_maxstep_parts(l[0], b[0], l[0] * l[0] - ll, l[0] * b[0] - lb, b[0] * b[0] - bb);
                if (sa < amax) amax = sa;
                if (sb < amax) amax = sb;
            }
            for (int i = 0; i < D; i++) sd_[off + i] = PRED ? a[i] : b[i];
        });
        amax = ex.min(amax);
        for (int q = 0; q < 2; q++) {
            const BigCone bc = big_cone(q);
            const cgptr v = Wv_ + bc.off; const cgptr l = lam_ + bc.off; const cgptr wr = Wirz_ + bc.off; const cgptr wb = Wibz_ + bc.off;
            const cgptr db = dw_ + bc.body - 1;   // db[i] = (J dw)_i, i >= 1
            const gptr so = sd_ + bc.off;
            double vd = 0;
            stream(1, bc.dim, [&](int i) { return D2{v[i], db[i]}; }, [&](int, const D2& w) { vd += w.a * w.b; });
            vd = ex.sum(vd);
            const double beta = Wbeta_[bc.cidx], ibeta = 1.0 / beta, v0 = v[0], d0 = dw_[bc.head], l0 = l[0];
            const double vx = v0 * d0 - vd;
            const double wij0 = (2.0 * vx * v0 - d0) * ibeta;
            const double a0 = wij0 - wr[0];
            const double b0 = PRED ? -l0 - a0 : -(wij0 + wb[0]);
            double ll = 0, la = 0, aa = 0, lb = 0, bb = 0, vb = 0, vw = 0;
            const gptr to = tc_ + bc.off;
            double pvl = 0, pvw = 0, pva = 0;
            if (PRED) {
                stream(1, bc.dim, [&](int i) { return D4{v[i], db[i], wr[i], l[i]}; },
                       [&](int i, const D4& w) {
                           const double ai = (-2.0 * vx * w.a + w.b) * ibeta - w.c, bi = -w.d - ai;
                           so[i] = ai;
                           ll += w.d * w.d; la += w.d * ai; aa += ai * ai; lb += w.d * bi; bb += bi * bi;
                           pvl += w.a * w.d; pvw += w.a * w.c; pva += w.a * ai;
                       });
            } else {
                stream(1, bc.dim, [&](int i) { return D5{v[i], db[i], wr[i], l[i], wb[i]}; },
                       [&](int i, const D5& w) {
                           const double wij = (-2.0 * vx * w.a + w.b) * ibeta;
                           const double ai = wij - w.c, bi = -(wij + w.e);
                           so[i] = bi;
                           if (CHECK) { to[i] = wij; vw += w.a * wij; }
                           ll += w.d * w.d; la += w.d * ai; aa += ai * ai; lb += w.d * bi; bb += bi * bi; vb += w.a * bi;
                       });
            }
            ll = ex.sum(ll); la = ex.sum(la); aa = ex.sum(aa); lb = ex.sum(lb); bb = ex.sum(bb);
            if (PRED) {
                bigq[q][0] = ll; bigq[q][1] = la; bigq[q][2] = aa;
                bigq[q][3] = ex.sum(pvl); bigq[q][4] = ex.sum(pvw); bigq[q][5] = ex.sum(pva);
            }
            if (!PRED) bigvz[q] = ex.sum(vb);
            if (CHECK) {
                const double vx2 = v0 * wij0 - ex.sum(vw);
                bigvx2[q] = vx2;
                if (ex.lane() == 0) to[0] = (2.0 * vx2 * v0 - wij0) * ibeta;
            }
            if (ex.lane() == 0) so[0] = PRED ? a0 : b0;
            const double sa = soc_maxstep_parts(l0, a0, l0 * l0 - ll, l0 * a0 - la, a0 * a0 - aa);
            const double sb = soc_maxstep_parts(l0, b0, l0 * l0 - ll, l0 * b0 - lb, b0 * b0 - bb);
            if (sa < amax) amax = sa;
            if (sb < amax) amax = sb;
        }
        ex.sync();
        SCVX_T1(22);
        return amax;
    }

    // ---- sweep 3: the corrector's cone right-hand side ----
    // With a = W^-1 ds_aff (in sd) and b = W dz_aff = -lam - a:   t = lam \ (-lam o lam - a o b + sigmu e),
    // Wibz = -Wirz - t,  tmpc = W^-1 Wibz.
    SCVX_HD_NI void corr_rhs_pass(double sigmu) {
        SCVX_THIS_LDS();
        SCVX_T0();
        const cgptr Wv_ = Wv; const cgptr lam_ = lam; const cgptr Wirz_ = Wirz; const cgptr sd_ = sd; const cgptr Wbeta_ = Wbeta;
        const gptr Wibz_ = Wibz; const gptr tmpc_ = tmpc;
        for_small([&](auto, auto Dt_, int off, int c, int) {
            constexpr int D = decltype(Dt_)::value;
            double l[D], a[D], b[D], wr[D];
            for (int i = 0; i < D; i++) { l[i] = lam_[off + i]; a[i] = sd_[off + i]; wr[i] = Wirz_[off + i]; b[i] = -l[i] - a[i]; }
            const double beta = Wbeta_[c];
            if constexpr (D == 1) {
                const double rhs = -l[0] * l[0] + (-a[0] * b[0] + sigmu);
                const double wb = -wr[0] - rhs / l[0];
                Wibz_[off] = wb; tmpc_[off] = wb / beta;
            } else {
                double rhs[D], t[D], wb[D], v[D], o[D];
                double ll = 0, ab = 0;
                for (int i = 0; i < D; i++) { ll += l[i] * l[i]; ab += a[i] * b[i]; }
                for (int i = 1; i < D; i++) rhs[i] = -2.0 * l[0] * l[i] - (a[0] * b[i] + b[0] * a[i]);
                rhs[0] = -ll + (-ab + sigmu);
                double ld = 0, l1 = 0;
                for (int i = 1; i < D; i++) { ld += l[i] * rhs[i]; l1 += l[i] * l[i]; }
                const double x0 = (l[0] * rhs[0] - ld) / (l[0] * l[0] - l1);
                const double il0 = 1.0 / l[0];
                t[0] = x0;
                for (int i = 1; i < D; i++) t[i] = (rhs[i] - x0 * l[i]) * il0;
                for (int i = 0; i < D; i++) { wb[i] = -wr[i] - t[i]; v[i] = Wv_[off + i]; }
                soc_W_small(v, beta, D, wb, o, true);
                for (int i = 0; i < D; i++) { Wibz_[off + i] = wb[i]; tmpc_[off + i] = o[i]; }
            }
        });
        for (int q = 0; q < 2; q++) {
            const BigCone bc = big_cone(q);
            const cgptr v = Wv_ + bc.off; const cgptr l = lam_ + bc.off; const cgptr wr = Wirz_ + bc.off; const cgptr a = sd_ + bc.off;
            const gptr wbo = Wibz_ + bc.off; const gptr to = tmpc_ + bc.off;
            const double l0 = l[0], a0 = a[0], b0 = -l0 - a0, v0 = v[0], wr0 = wr[0];
            const double beta = Wbeta_[bc.cidx], ibeta = 1.0 / beta;
            // no reduction sweep: rhs_i = -2 l0 l_i - (a0 b_i + b0 a_i) = c1 l_i + c2 a_i (b_i = -l_i - a_i), so every sum over the body is a
            // combination of the six the predictor's direction pass took on its way (bigq)
            const double c1 = -2.0 * l0 + a0, c2 = a0 - b0;
            const double ll = bigq[q][0], la = bigq[q][1], aa = bigq[q][2], vl = bigq[q][3], vw = bigq[q][4], va = bigq[q][5];
            const double ab = -la - aa, lr = c1 * ll + c2 * la, vr = c1 * vl + c2 * va;
            const double rhs0 = -(ll + l0 * l0) + (-(ab + a0 * b0) + sigmu);
            const double x0 = (l0 * rhs0 - lr) / (l0 * l0 - ll);
            const double il0 = 1.0 / l0;
            const double wb0 = -wr0 - x0;
            const double vt = (vr - x0 * vl) * il0;                 // <v, t>_1
            const double vx = v0 * wb0 - (-vw - vt);                // inverse scaling of Wibz
            stream(1, bc.dim, [&](int i) { return D4{l[i], a[i], v[i], wr[i]}; },
                   [&](int i, const D4& w) {
                       const double bi = -w.a - w.b, ri = -2.0 * l0 * w.a - (a0 * bi + b0 * w.b);
                       const double wbi = -w.d - (ri - x0 * w.a) * il0;
                       wbo[i] = wbi;
                       to[i] = (-2.0 * vx * w.c + wbi) * ibeta;
                   });
            if (ex.lane() == 0) { wbo[0] = wb0; to[0] = (2.0 * vx * v0 - wb0) * ibeta; }
        }
        ex.sync();
        SCVX_T1(22);
    }

    // ---- sweep 5: S += alpha (J dw - rz) with rz = S - a(V) recomputed from the old V,  Z += alpha W^-1 (W dz) ----
    // ... and V += alpha dw on the way: the bodies of the two big cones ARE the local part of V (Vn: where the new iterate goes -- V
    // itself, or the other buffer while V holds the best iterate so far)
    SCVX_HD_NI void update_pass(double alpha, gptr Vn) {
        SCVX_THIS_LDS();
        SCVX_T0();
        const gptr S_ = S; const gptr Z_ = Z; const cgptr V_ = V; const cgptr dw_ = dw; const cgptr Wv_ = Wv; const cgptr sd_ = sd;
        const cgptr Wbeta_ = Wbeta;
        for_small([&](auto G_, auto Dt_, int off, int c, int q) {
            constexpr int GRP = decltype(G_)::value, D = decltype(Dt_)::value;
            double s[D], z[D], a[D], jd[D], b[D];
            for (int i = 0; i < D; i++) { s[i] = S_[off + i]; z[i] = Z_[off + i]; b[i] = sd_[off + i]; }
            small_gather<GRP, D>(q, V_, 1.0, a);
            small_gather<GRP, D>(q, dw_, 0.0, jd);
            const double beta = Wbeta_[c];
            if constexpr (D == 1) {
                S_[off] = s[0] + alpha * (jd[0] - (s[0] - a[0]));
                Z_[off] = z[0] + alpha * (b[0] / beta);
            } else {
                double v[D], dz[D];
                for (int i = 0; i < D; i++) v[i] = Wv_[off + i];
                soc_W_small(v, beta, D, b, dz, true);
                for (int i = 0; i < D; i++) { S_[off + i] = s[i] + alpha * (jd[i] - (s[i] - a[i])); Z_[off + i] = z[i] + alpha * dz[i]; }
            }
        });
        for (int q = 0; q < 2; q++) {
            const BigCone bc = big_cone(q);
            const gptr s = S_ + bc.off; const gptr z = Z_ + bc.off; const cgptr v = Wv_ + bc.off; const cgptr b = sd_ + bc.off;
            const cgptr ab = V_ + bc.body - 1; const cgptr db = dw_ + bc.body - 1;
            const double beta = Wbeta_[bc.cidx], ibeta = 1.0 / beta, v0 = v[0], b0 = b[0];
            const double vx = v0 * b0 - bigvz[q];
            const double s0 = s[0], z0 = z[0], a0 = V_[bc.head], d0 = dw_[bc.head];
            ex.sync();   // every lane holds the heads before lane 0 overwrites them
            const gptr vnb = Vn + bc.body - 1;
            double q0 = 0, q1 = 0, q2 = 0, q3 = 0, q4 = 0, q5 = 0;
            stream(1, bc.dim, [&](int i) { return D6{s[i], z[i], v[i], b[i], ab[i], db[i]}; },
                   [&](int i, const D6& w) {
                       const double sn = w.a + alpha * (w.f - (w.a - w.e));
                       const double zn = w.b + alpha * ((-2.0 * vx * w.c + w.d) * ibeta);
                       const double vn = w.e + alpha * w.f;
                       s[i] = sn; z[i] = zn; vnb[i] = vn;
                       const double rn = sn - vn;
                       q0 += sn * sn; q1 += zn * zn; q2 += sn * zn; q3 += sn * rn; q4 += zn * rn; q5 += rn * rn;
                   });
            bigs[q][0] = ex.sum(q0); bigs[q][1] = ex.sum(q1); bigs[q][2] = ex.sum(q2);
            bigs[q][3] = ex.sum(q3); bigs[q][4] = ex.sum(q4); bigs[q][5] = ex.sum(q5);
            if (ex.lane() == 0) {
                s[0] = s0 + alpha * (d0 - (s0 - a0));
                z[0] = z0 + alpha * ((2.0 * vx * v0 - b0) * ibeta);
            }
        }
        ex.sync();
        for (int i = L.nloc + ex.lane(); i < L.nv; i += ex.nlanes()) Vn[i] = V_[i] + alpha * dw_[i];   // the four global variables
        bigs_ok = true;
        {   // ... and the multipliers
            const gptr y_ = y; const cgptr dy_ = dy;
            stream(0, L.ny, [&](int i) { return D2{y_[i], dy_[i]}; }, [&](int i, const D2& v) { y_[i] = v.a + alpha * v.b; });
        }
        ex.sync();
        SCVX_T1(23);
    }

    // ---- Hb^-1 on a local vector (dx, du, nu); g and out must not alias ----
    // One lane per output ROW: row j of the compact node inverse has at most three entries (hxi_apply's formulas), so
    // every lane loads three coefficients and three inputs that sit next to those of its neighbours.  (One lane per
    // node, as build_kkt assembles the blocks, reads 51 elements at a stride of 33 doubles: no two lanes share a line.)
    SCVX_HD_NI void Hb_inv(cgptr g, gptr out, bool with_nu = true) {
        SCVX_THIS_LDS();
        SCVX_T0();
        const int K = L.K;
        const cgptr hx_ = hx; const cgptr hu_ = hu;
        stream(0, L.nx,
               [&](int t) {
                   const int k = t / 14, j = t - 14 * k;
                   // first coefficient / first input / number of terms of row j
                   const int hb = j == 0 ? HX_M : j < 4 ? HX_R + 3 * (j - 1) : j < 7 ? HX_V + 3 * (j - 4) : j < 9 ? HX_Q
                                  : j < 11 ? HX_Q34 + 2 * (j - 9) : HX_W + 3 * (j - 11);
                   const int gb = j == 0 ? 0 : j < 4 ? 1 : j < 7 ? 4 : j < 9 ? j : j < 11 ? 9 : 11;
                   const int n = (j == 0 || j == 7 || j == 8) ? 1 : (j == 9 || j == 10) ? 2 : 3;
                   cgptr h = hx_ + (size_t)k * HX_SZ + hb; cgptr x = g + 14 * k + gb;
                   const int i1 = n > 1 ? 1 : 0, i2 = n > 2 ? 2 : 0;
                   return D6{h[0], n > 1 ? h[i1] : 0.0, n > 2 ? h[i2] : 0.0, x[0], x[i1], x[i2]};
               },
               [&](int t, const D6& w) { out[t] = w.a * w.d + w.b * w.e + w.c * w.f; });
        {
            cgptr gu = g + L.nx; gptr ou = out + L.nx;
            stream(0, L.nu_,
                   [&](int t) {
                       const int k = t / NU, c = t - NU * k;
                       if (NU == 3 || c < 3) { cgptr h = hu_ + HU_SZ * k + 3 * c; cgptr x = gu + NU * k; return D6{h[0], h[1], h[2], x[0], x[1], x[2]}; }
                       cgptr h = hu_ + HU_SZ * k + 9 + 2 * (c - 3); cgptr x = gu + NU * k + 3;   // row c - 3 of the 2x2 fin block
                       return D6{h[0], h[1], 0.0, x[0], x[1], 0.0};
                   },
                   [&](int t, const D6& w) { ou[t] = w.a * w.d + w.b * w.e + w.c * w.f; });
        }
        if (with_nu) {
            cgptr gn = g + L.nx + L.nu_; gptr on = out + L.nx + L.nu_;
            const double hn = hnui;
            stream<8>(0, 14 * K, [&](int i) { return gn[i]; }, [&](int i, double v) { on[i] = hn * v; });
        }
        ex.sync();
        SCVX_T1(3);
    }

    // ---- block-tridiagonal solve S x = r (r, x: [K][14], distinct buffers) ----
    // S = L L' with L block lower bidiagonal (diagonal blocks L_k, sub-diagonal blocks Wb_k).  Stored per segment:
    // Linv_k = L_k^-1 and the coupling tile Nf_k = -Linv_k Wb_{k-1} (negated, transposed: element (i,j) at 14 j + i).
    //     forward   t_k = Linv_k r_k + Nf_k t_{k-1}
    //     backward  w_k = t_k + Nf_{k+1}' w_{k+1},   x_k = Linv_k' w_k
    // (L'^-1 in terms of w = L_k' x_k: Wb_k' x_{k+1} = Wb_k' Linv_{k+1}' w_{k+1} = -Nf_{k+1}' w_{k+1}, so the SAME tile
    // serves both sweeps and no second coupling matrix is formed, stored or read.)
    // The Linv products are chain-free (all k in parallel); only the 14x14 matrix-vector recurrences are sequential
    // and run inside the executor (ex.chain: FP64 matrix pipe on the device, no barriers).
    // The two recurrences of a solve: on entry t holds z = L^-1 r, on exit t holds w (x is scratch: the forward result).
    // TWO-ENDED form (Ex::kTwisted, Solver::factor_twisted): nodes 0..m-1 are eliminated downwards, K-1..m+1 upwards, the
    // middle node m last.  Tile slot j couples nodes j-1 and j: for j <= m it is N_j = -L_j^-1 Wb_{j-1} in the standard
    // (transposed) layout; for j > m it holds N'_{j-1} = -L_{j-1}^-1 Wb'_j untransposed -- so the executor's `reverse`
    // recurrence IS the bottom half's forward substitution and its forward recurrence the bottom half's back substitution,
    // and the two halves run side by side on wavefronts 0 and 2.
    SCVX_HD bool twisted() const { return Ex::kTwisted && L.K >= 8; }
    // forward recurrences: x = z + (coupling) x_prev over the factor's elimination order (two-ended: both halves, then the middle node)
    template <int N>
    SCVX_HD void chains_fwd(const cgptr (&tz)[N], const gptr (&x)[N]) {
        const int K = L.K;
        const cfptr Nf = this->Nf;
        if constexpr (Ex::kTwisted) {
            if (twisted()) {
                const int m = K / 2;
                ex.template chain_range_n<N>(0, K, tz, Nf, x, false, 0, m, true);
                ex.template chain_range_n<N>(2, K, tz, Nf, x, true, K - 1, K - 1 - m, true);
                ex.sync();
                for (int e = ex.lane(); e < 14 * N; e += ex.nlanes()) {   // the middle node: both neighbours feed it
                    const int q = e / 14, i = e - 14 * q;
                    cfptr Na = Nf + (size_t)m * 196 + i;            // N_m(i, j) at 14 j + i
                    cfptr Nb = Nf + (size_t)(m + 1) * 196 + 14 * i;   // N'_m(i, j) at 14 i + j
                    cgptr xa = x[q] + 14 * (m - 1); cgptr xb = x[q] + 14 * (m + 1);
                    double a = tz[q][14 * m + i];
                    for (int j = 0; j < 14; j++) a += Na[14 * j] * xa[j] + Nb[j] * xb[j];
                    x[q][14 * m + i] = a;
                }
                ex.sync();
                return;
            }
        }
        ex.template chain_n<N>(K, tz, Nf, x, false);
        ex.sync();
    }
    // backward recurrences (the transposed couplings, in reverse elimination order): t = x + (coupling)' t_next
    template <int N>
    SCVX_HD void chains_bwd(const cgptr (&xz)[N], const gptr (&t)[N]) {
        const int K = L.K;
        const cfptr Nf = this->Nf;
        if constexpr (Ex::kTwisted) {
            if (twisted()) {
                const int m = K / 2;
                ex.template chain_range_n<N>(0, K, xz, Nf, t, true, m, m + 1, true);
                ex.template chain_range_n<N>(2, K, xz, Nf, t, false, m, K - m, false);
                ex.sync();
                return;
            }
        }
        ex.template chain_n<N>(K, xz, Nf, t, true);
        ex.sync();
    }
    template <int N>
    SCVX_HD void solve_chains(const gptr (&t)[N], const gptr (&x)[N]) {
        cgptr tz[N], xz[N];
        for (int q = 0; q < N; q++) { tz[q] = t[q]; xz[q] = x[q]; }
        chains_fwd<N>(tz, x);
        chains_bwd<N>(xz, t);
    }
    SCVX_HD_NI void S_solve(cgptr r, gptr x) {
        SCVX_THIS_LDS();
        SCVX_T0();
        const int K = L.K;
        const cfptr Linv = this->Linv;
        const gptr tchain = this->tchain;
        for (int t = ex.lane(); t < 14 * K; t += ex.nlanes()) {
            const int k = t / 14, i = t - 14 * k;
            cfptr Li = Linv + (size_t)k * LINV_SZ + linv_row(i);
            cgptr rk_ = r + 14 * k;
            double a = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
            for (int j = 0; j < 14; j++) a += (j <= i ? Li[j <= i ? j : i] : 0.0) * rk_[j];   // fixed trip count: the loads batch
            tchain[t] = a;
        }
        ex.sync();
        SCVX_TE(t0_, 16);
        SCVX_TS(tc1_);
        {
            const gptr ts[1] = {tchain};
            const gptr xs[1] = {x};
            solve_chains<1>(ts, xs);              // t -> x -> w in tchain
        }
        SCVX_TE(tc1_, 17);
        SCVX_TS(tp2_);
        for (int t = ex.lane(); t < 14 * K; t += ex.nlanes()) {
            const int k = t / 14, i = t - 14 * k;
            cfptr Lk = Linv + (size_t)k * LINV_SZ;
            cgptr wk = tchain + 14 * k;
            double a = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
            for (int j = 0; j < 14; j++) a += (j >= i ? Lk[linv_row(j) + (j >= i ? i : j)] : 0.0) * wk[j];   // column i of L^-1
            x[t] = a;
        }
        ex.sync();
        SCVX_TE(tp2_, 18);
        SCVX_T1(0);
    }

    // forward half of S_solve: t = L^-1 r (tchain is scratch; r and t distinct buffers)
    SCVX_HD_NI void S_fwd(cgptr r, gptr t) {
        SCVX_THIS_LDS();
        SCVX_T0();
        const int K = L.K;
        const cfptr Linv = this->Linv;
        const gptr tchain = this->tchain;
        for (int e = ex.lane(); e < 14 * K; e += ex.nlanes()) {
            const int k = e / 14, i = e - 14 * k;
            cfptr Li = Linv + (size_t)k * LINV_SZ + linv_row(i);
            cgptr rk_ = r + 14 * k;
            double a = 0;
            SCVX_UNROLL
            for (int j = 0; j < 14; j++) a += (j <= i ? Li[j <= i ? j : i] : 0.0) * rk_[j];   // fixed trip count: the loads batch
            tchain[e] = a;
        }
        ex.sync();
        SCVX_TE(t0_, 16);
        SCVX_TS(tc1_);
        {
            const cgptr tz[1] = {tchain};
            const gptr xs[1] = {t};
            chains_fwd<1>(tz, xs);
        }
        SCVX_TE(tc1_, 17);
        SCVX_T1(0);
    }

    // N right-hand sides (N <= 4) through the block-tridiagonal solve with ONE pass over Linv / Nf per sweep.
    // r, x, t: N arrays of [K][14] each (t: scratch); nothing may alias.
    template <int N>
    SCVX_HD_NI void S_solveN(const cgptr (&r)[N], const gptr (&x)[N], const gptr (&t)[N]) {
        SCVX_THIS_LDS();
        const int K = L.K;
        const cfptr Linv = this->Linv;
        for (int e = ex.lane(); e < 14 * K; e += ex.nlanes()) {
            const int k = e / 14, i = e - 14 * k;
            cfptr Li = Linv + (size_t)k * LINV_SZ + linv_row(i);
            double a[N];
            for (int q = 0; q < N; q++) a[q] = 0;
            SCVX_UNROLL
            for (int j = 0; j < 14; j++) {
                const double l = j <= i ? Li[j <= i ? j : i] : 0.0;
                for (int q = 0; q < N; q++) a[q] += l * r[q][14 * k + j];
            }
            for (int q = 0; q < N; q++) t[q][e] = a[q];
        }
        ex.sync();
        solve_chains<N>(t, x);
        for (int e = ex.lane(); e < 14 * K; e += ex.nlanes()) {
            const int k = e / 14, i = e - 14 * k;
            cfptr Lk = Linv + (size_t)k * LINV_SZ;
            double a[N];
            for (int q = 0; q < N; q++) a[q] = 0;
            SCVX_UNROLL
            for (int j = 0; j < 14; j++) {
                const double l = j >= i ? Lk[linv_row(j) + (j >= i ? i : j)] : 0.0;
                for (int q = 0; q < N; q++) a[q] += l * t[q][14 * k + j];
            }
            for (int q = 0; q < N; q++) x[q][e] = a[q];
        }
        ex.sync();
    }
    // The BACKWARD half of S_solveN for right-hand sides whose forward substitution has already been done (inside the factorisation
    // loop, build_kkt): on entry x[q] holds t = L^-1 r, on exit x[q] holds the solution; t[q] is scratch.
    template <int N>
    SCVX_HD_NI void S_backN(const gptr (&x)[N], const gptr (&t)[N]) {
        SCVX_THIS_LDS();
        const int K = L.K;
        const cfptr Linv = this->Linv;
        cgptr xz[N];
        for (int q = 0; q < N; q++) xz[q] = x[q];
        chains_bwd<N>(xz, t);
        for (int e = ex.lane(); e < 14 * K; e += ex.nlanes()) {
            const int k = e / 14, i = e - 14 * k;
            cfptr Lk = Linv + (size_t)k * LINV_SZ;
            double a[N];
            for (int q = 0; q < N; q++) a[q] = 0;
            SCVX_UNROLL
            for (int j = 0; j < 14; j++) {
                const double l = j >= i ? Lk[linv_row(j) + (j >= i ? i : j)] : 0.0;
                for (int q = 0; q < N; q++) a[q] += l * t[q][14 * k + j];
            }
            for (int q = 0; q < N; q++) x[q][e] = a[q];
        }
        ex.sync();
    }
    // out0 = E v0, out1 = E v1 + add1 (both without the s column) with one pass over D
    SCVX_HD_NI void E_apply2(cgptr v0, cgptr v1, gptr out0, gptr out1, cgptr add1) {
        SCVX_THIS_LDS();
        SCVX_T0();
        const int K = L.K;
        for (int r = ex.lane(); r < 14 * K; r += ex.nlanes()) {
            const int k = r / 14, i = r - 14 * k;
            dcptr Dk = D + (size_t)k * DSZ + i;
            const double ad = add1[r];
            double a = 0, b = 0;
            for (int j = 0; j < 14; j++) { const double d = Dk[14 * j]; a += d * v0[14 * k + j]; b += d * v1[14 * k + j]; }
            for (int j = 0; j < 2 * NU; j++) { const double d = Dk[14 * (14 + j)]; a += d * v0[L.nx + NU * k + j]; b += d * v1[L.nx + NU * k + j]; }
            a += v0[L.nx + L.nu_ + r] - v0[14 * (k + 1) + i];
            b += v1[L.nx + L.nu_ + r] - v1[14 * (k + 1) + i];
            out0[r] = a;
            out1[r] = b + ad;
        }
        ex.sync();
        SCVX_T1(1);
    }

    // ---- the K-step factorisation loop as a two-wavefront pipeline (multi-wavefront executors only) ----
    // The sequential loop (build_kkt) spends ~16 k cycles per segment in one wavefront: half of it assembles the Schur
    // blocks Sd_k, So_k from the linearisation and the node inverses -- work that does not depend on the Cholesky chain
    // -- and half is the chain itself (pivot update, Cholesky + inverse, coupling tile).  With more than one wavefront per
    // trajectory the two halves run side by side: wavefront 1 PRODUCES (Sd_k, So_k) into a two-slot LDS ring, wavefront 0
    // CONSUMES them one segment behind; one workgroup barrier per segment.  Same arithmetic, same results.
    //   producer k:  TBp_k;  Sd_k = Hxi_{k+1} + hnui I + [TA|TBm|TBp]_k D_k';  TA_{k+1}, TBm_{k+1};  So_k = -TA_{k+1} + TBm_{k+1} Bp_k'
    //   consumer k:  M = Sd_k - Wb_{k-1} Wb_{k-1}';  L^-1 = chol_inv(M);  store L^-1;  Nf_k = -L^-1 Wb_{k-1};  Wb_k = So_k L^-T
    // Since round 4 the pipeline also carries the border (as the sequential loop of build_kkt does, see there): the assembly wavefront
    // forms r_k = (E Hb^-1 g)_k for the four right-hand sides from the tiles it holds, and the CHAIN wavefront -- which used to wait
    // two thirds of every step at the hand-over barrier -- stores L_k^-1 and forms and stores N_k; the separate post stage is gone.
    // (Measured at B = 1,024, two wavefronts per trajectory: the assembly wavefront was the bottleneck with 7.5 k cycles per step of
    // which 4.3 k post stage; profiles/r04_k4_sections_small.txt.)  The forward substitution t_k = L_k^-1 r_k + N_k t_{k-1} ran on the chain
    // wavefront too in round 4; round 5 made the assembly stage cheap enough that it went back there (see fwd_subst below).
    template <class E2 = Ex>
    SCVX_HD_NI bool factor_pipelined(bool with_pred) {
        SCVX_THIS_LDS();
        const int K = L.K;
        const dcptr D_ = D; const cgptr hx_ = hx; const cgptr hu_ = hu;
        const fptr Linv_ = Linv; const fptr Nf_ = Nf;
        const double hnui_ = hnui;
        double* sc = ex.pipe_scratch();
        double* Sd = sc;                 // 2 x 196: ring; the chain factorises slot k in place (pivot tile)
        double* So = Sd + 392;           // 2 x 196
        double* Wp = So + 392;           // 3 x 196: Wb ring (chain writes k, reads k-1; the post stage reads k-2)
        double* Li = Wp + 588;           // 2 x 196: Linv ring (chain writes k; the post stage reads k-1)
        double* Mq = Li + 392;           // 2 x 196: N_k ring (chain writes k; the assembly wavefront's forward substitution reads k one step later)
        double* Dt = Mq + 392;           // producer: D_k tile
        double* T = Dt + DSZ;            // producer: [TA | TBm | TBp], row stride 22
        double* Bp = T + 14 * TS;            // producer: Bp_k kept across the tile swap
        double* Hh = Bp + BPN;            // producer: node inverses k | k+1
        double* Hd = Hh + 2 * NODE_SZ;   // producer: dense Hxi_{k+1}
        double* Gn = Hd + 196;           // producer: 2 x NXU x 4 node slices of the border right-hand sides (columns 1 = Ptr, 3 = gx)
        double* Rr = Gn + 2 * NXU * 4;   // 3 x 56: r_k ring (producer writes k, the chain reads k one step later)
        double* Tt = Rr + 168;           // forward substitution (assembly wavefront): 2 x 56  t_{k-1}, t_k
        double* Sg = Tt + 112;           // producer: 42 segment scalars gx_nu,k | ry_k | Pnu_k
        const int w = ex.wave(), l = ex.wlane();
        bool ok = true;
        auto node_elem = [&](int node, int e) -> double {
            return e < HX_SZ ? hx_[(size_t)node * HX_SZ + e] : hu_[HU_SZ * node + (e - HX_SZ)];
        };
        const cgptr Wtr_ = Wv + L.o_tr + 1; const cgptr Wnu_ = Wv + L.o_nu + 1; const cgptr gx_ = gx; const cgptr ry_ = ry;
        const int nx_ = L.nx, nxu_ = L.nx + L.nu_;
        auto gnode_elem = [&](int nd, int e) -> double {   // element e = NXU vec + row of node nd (vec 0: Ptr, 1: gx)
            const int vec = e >= NXU ? 1 : 0, row = e - NXU * vec;
            if (vec == 1 && !with_pred) return 0.0;
            cgptr src = vec ? gx_ : Wtr_;
            return row < 14 ? src[14 * nd + row] : src[nx_ + NU * nd + (row - 14)];
        };
        auto gseg_elem = [&](int sgk, int e) -> double {
            if (e >= 28) return Wnu_[14 * sgk + (e - 28)];
            if (!with_pred) return 0.0;
            return e < 14 ? gx_[nxu_ + 14 * sgk + e] : ry_[14 * sgk + (e - 14)];
        };
        const gptr xq_[4] = {ys, ytr, ynu, dy};
        if (w == 1) {   // producer prologue: D_0, node 0 -> TA_0, TBm_0
            for (int e = l; e < 2 * NXU * 4; e += 64) Gn[e] = 0.0;   // columns 0 and 2 stay zero
            ex.w_sync_lds();
            for (int e = l; e < 2 * NXU; e += 64) {
                const int vec = e >= NXU ? 1 : 0, row = e - NXU * vec;
                Gn[NXU * 4 + 4 * row + (vec ? 3 : 1)] = gnode_elem(0, e);
            }
            for (int e = l; e < DSZ; e += 64) Dt[e] = D_[e];
            for (int e = l; e < NODE_SZ; e += 64) Hh[NODE_SZ + e] = node_elem(0, e);
            ex.w_sync_lds();
            for (int e = l; e < 196; e += 64) Hd[e] = hxi_entry(Hh + NODE_SZ, e / 14, e % 14);
            ex.w_sync_lds();
            ex.w_tile_gemm(T, TS, 1, Dt, 1, 14, Hd, 14, 1, 14, 1.0, false);
            for (int q = l; q < BPN; q += 64) {
                const int i = q / NU, c = q - NU * i;
                T[TS * i + 14 + c] = bhu(Dt, 14, i, c, Hh + NODE_SZ + HX_SZ);
            }
            ex.w_sync_lds();
        }
        const int hpos_lane = hx_dense_pos(l <= HX_SZ ? l : 0);
        typename E2::WAcc cg;   // chain wavefront: Gram matrix of the forward-substituted right-hand sides (Solver::gram)
        ex.w_acc_zero(cg);
        double gnacc = 0.0;     // assembly wavefront: <Pnu, gx_nu>
        double hnext = (w == 1 && l < NODE_SZ) ? node_elem(1, l) : 0.0;   // the assembly wavefront keeps the next node's inverses one step ahead
        double gnext = (w == 1 && l < 2 * NXU) ? gnode_elem(1, l) : 0.0;  // ... and the right-hand sides' slices and segment scalars
        double sgnext = (w == 1 && l < 42) ? gseg_elem(0, l) : 0.0;
        // Round 5: the forward substitution t_k = L_k^-1 r_k + N_k t_{k-1} of the four border right-hand sides (with their Gram matrix)
        // is the ASSEMBLY wavefront's again, one barrier behind the chain: with the tile products unpredicated and the dense node inverse
        // scattered, the assembly stage shrank to 7.2 k cycles per segment while the chain wavefront (pivot update, Cholesky + inverse,
        // coupling tile, L^-1 / N_k stores AND this substitution) took 11.4 k and the assembly waited 4.2 k of every step.
        // Round 6: the substitution itself (two small products on tiles the chain wavefront has just written) is the CHAIN's again -- the DPP
        // Cholesky took 1.4 k cycles off its step and left the assembly wavefront the longer one (9.0 k against 7.0 k at B = 1,024) -- while the
        // Gram matrix and the stores of t_k stay here, a barrier later.
        auto fwd_subst = [&](int kp) {
            const double* Tc = Tt + 56 * (kp & 1);
            ex.w_acc_mac(cg, Tc, 1, 4, Tc, 4, 1, 14, 1.0, 4);   // Gram matrix of the forward-substituted right-hand sides
            if (l < 56) {
                const int q = l / 14, i = l - 14 * q;
                if (q < 3 || with_pred) xq_[q][14 * kp + i] = Tc[4 * i + q];
            }
        };
        for (int t = 0; t <= K; t++) {
            if (w == 1 && t < K) {
                const int k = t;
                SCVX_TS(tp0_);
                double* Sdk = Sd + 196 * (k & 1); double* Sok = So + 196 * (k & 1);
                // next tile into registers while this segment is assembled
                double pre[NPW];
                dcptr Dn = D_ + (size_t)(k + 1 < K ? k + 1 : k) * DSZ;
                SCVX_UNROLL
                for (int q = 0; q < NPW; q++) { const int e = l + 64 * q; pre[q] = e < DSZ ? Dn[e] : 0.0; }
                for (int e = l; e < NODE_SZ; e += 64) { Hh[e] = Hh[NODE_SZ + e]; }
                if (l < 2 * NXU) {   // slot 0 <- slot 1 (node k)
                    const int vec = l >= NXU ? 1 : 0, row = l - NXU * vec, at = 4 * row + (vec ? 3 : 1);
                    Gn[at] = Gn[NXU * 4 + at];
                }
                if (l < 42) Sg[l] = sgnext;
                ex.w_sync_lds();
                if (l < NODE_SZ) Hh[NODE_SZ + l] = hnext;                          // node k + 1, requested a step ago
                hnext = l < NODE_SZ ? node_elem(k + 2 <= K ? k + 2 : K, l) : 0.0;   // node k + 2 for the next step
                if (l < 2 * NXU) {
                    const int vec = l >= NXU ? 1 : 0, row = l - NXU * vec;
                    Gn[NXU * 4 + 4 * row + (vec ? 3 : 1)] = gnext;
                }
                gnext = l < 2 * NXU ? gnode_elem(k + 2 <= K ? k + 2 : K, l) : 0.0;
                sgnext = l < 42 ? gseg_elem(k + 1 < K ? k + 1 : k, l) : 0.0;
                ex.w_sync_lds();
                if (l <= HX_SZ) Hd[hpos_lane] = Hh[NODE_SZ + (l < HX_SZ ? l : HX_Q)];   // the 34 non-zeros of the dense Hxi_{k+1}
                for (int q = l; q < BPN; q += 64) {
                    const int i = q / NU, c = q - NU * i;
                    T[TS * i + 14 + NU + c] = bhu(Dt, 14 + NU, i, c, Hh + NODE_SZ + HX_SZ);
                }
                ex.w_sync_lds();
                SCVX_TE(tp0_, 24);
                SCVX_TS(tp1_);
                {
                    double* Rk = Rr + 56 * (k % 3);
                    typename E2::WAcc cm, cr;
                    ex.w_acc_zero(cm); ex.w_acc_zero(cr);
                    ex.w_acc_mac(cm, T, TS, 1, Dt, 14, 1, TW, 1.0);
                    // r_k (columns 1 and 3): [TA | TBm]_k [g_x,k; g_u,k] + TBp_k g_u,k+1 - Hxi_{k+1} g_x,k+1
                    ex.w_acc_mac(cr, T, TS, 1, Gn, 4, 1, 14 + NU, 1.0, 4);
                    ex.w_acc_mac(cr, T + 14 + NU, TS, 1, Gn + NXU * 4 + 14 * 4, 4, 1, NU, 1.0, 4);
                    ex.w_acc_mac(cr, Hd, 14, 1, Gn + NXU * 4, 4, 1, 14, -1.0, 4);
                    ex.w_acc_store_init(cm, Sdk, Hd, hnui_);   // Sd_k = Hxi_{k+1} + hnui I + the products
                    ex.w_acc_store(cr, Rk, 4, 1, false, 4);
                    ex.w_sync_lds();
                    // the plain parts: column 0 = Sg_k, column 2 = hnui Pnu_k, column 3 += hnui gx_nu,k + ry_k
                    if (l < 14) {
                        Rk[4 * l] = Dt[14 * CS + l];
                        Rk[4 * l + 2] = hnui_ * Sg[28 + l];
                        Rk[4 * l + 3] += hnui_ * Sg[l] + Sg[14 + l];
                        gnacc += Sg[28 + l] * Sg[l];
                    }
                }
                SCVX_TE(tp1_, 25);
                SCVX_TS(tp2_);
                if (k + 1 < K) {
                    for (int q = l; q < BPN; q += 64) Bp[q] = Dt[14 * (14 + NU) + q];
                    ex.w_sync_lds();
                    SCVX_UNROLL
                    for (int q = 0; q < NPW; q++) { const int e = l + 64 * q; if (e < DSZ) Dt[e] = pre[q]; }
                    ex.w_sync_lds();
                    ex.w_tile_gemm(T, TS, 1, Dt, 1, 14, Hd, 14, 1, 14, 1.0, false);
                    for (int q = l; q < BPN; q += 64) {
                        const int i = q / NU, c = q - NU * i;
                        T[TS * i + 14 + c] = bhu(Dt, 14, i, c, Hh + NODE_SZ + HX_SZ);
                    }
                    ex.w_sync_lds();
                    for (int e = l; e < 196; e += 64) {
                        const int i = e / 14, j = e - 14 * i;
                        Sok[e] = so_elem(T, Bp, i, j);
                    }
                }
                ex.w_sync_lds();
                SCVX_TE(tp2_, 26);
            }
            if (w == 0 && t >= 1 && t <= K) {
                // ---- chain: pivot tile in place on the ring slot, Cholesky + inverse, coupling tile ----
                const int k = t - 1;
                SCVX_TS(tc0_);
                double* M = Sd + 196 * (k & 1); const double* Sok = So + 196 * (k & 1);
                double* Lik = Li + 196 * (k & 1);
                const double* Wpm = Wp + 196 * ((k + 2) % 3);   // Wb[k-1]
                if (k > 0) { ex.w_tile_gemm(M, 14, 1, Wpm, 14, 1, Wpm, 1, 14, 14, -1.0, true); ex.w_sync_lds(); }
                SCVX_TE(tc0_, 24);
                SCVX_TS(tc1_);
                ok = ex.w_chol_inv14(M, Lik) && ok;
                ex.w_sync_lds();
                SCVX_TE(tc1_, 25);
                SCVX_TS(tc3_);
                if (k + 1 < K) { ex.w_tile_gemm(Wp + 196 * (k % 3), 14, 1, Sok, 14, 1, Lik, 1, 14, 14, 1.0, false); ex.w_sync_lds(); }
                SCVX_TE(tc3_, 27);
                // ---- what used to be the post stage, now in the chain wavefront's idle time: L^-1 (packed) and N_k = -L_k^-1 Wb_{k-1} to
                // HBM, then the forward substitution of the four right-hand sides, t_k = L_k^-1 r_k + N_k t_{k-1} ----
                SCVX_TS(tc2_);
                for (int e = l; e < LINV_SZ; e += 64) {
                    const int p = e / 15, q = e - 15 * p;
                    const int i = q <= p ? p : 13 - p, jj = q <= p ? q : q - (p + 1);
                    Linv_[(size_t)k * LINV_SZ + e] = Lik[14 * i + jj];
                }
                if (k > 0) {
                    double* Mqk = Mq + 196 * (k & 1);
                    ex.w_tile_gemm(Mqk, 1, 14, Lik, 14, 1, Wpm, 14, 1, 14, -1.0, false);
                    ex.w_sync_lds();
                    for (int e = l; e < 196; e += 64) Nf_[(size_t)k * 196 + e] = Mqk[e];
                }
                {   // t_k = L_k^-1 r_k + N_k t_{k-1}: r_k was formed by the assembly wavefront a barrier ago, L_k^-1 and N_k just now
                    const double* Rk = Rr + 56 * (k % 3);
                    double* Tc = Tt + 56 * (k & 1); const double* Tp = Tt + 56 * ((k + 1) & 1);
                    typename E2::WAcc ct;
                    ex.w_acc_zero(ct);
                    ex.w_acc_mac(ct, Lik, 14, 1, Rk, 4, 1, 14, 1.0, 4);
                    if (k > 0) ex.w_acc_mac(ct, Mq + 196 * (k & 1), 1, 14, Tp, 4, 1, 14, 1.0, 4);
                    ex.w_acc_store(ct, Tc, 4, 1, false, 4);
                }
                SCVX_TE(tc2_, 26);
            }
            if (w == 1 && t >= 2) fwd_subst(t - 2);   // the assembly wavefront: Gram matrix and stores of t of segment t - 2, finished by the chain a barrier ago
            SCVX_TS(tb_);
            ex.sync();   // hand-over: producer's slot k is complete, consumer has finished with slot k - 1
            SCVX_TE(tb_, 28);
        }
        if (w == 1) {
            fwd_subst(K - 1);
            ex.w_sync_lds();
            ex.w_acc_store(cg, Rr, 4, 1, false, 4);
        }
        ex.sync();
        for (int q = 0; q < 16; q++) gram[q] = Rr[q];
        gn_pred = ex.sum(gnacc);
        return ex.all(ok);
    }

    // TWO-ENDED (twisted) form of the pipelined factorisation, four wavefronts per trajectory: wavefronts 1 / 0 assemble and
    // eliminate nodes 0 .. m-1 downwards exactly as factor_pipelined does, wavefronts 3 / 2 assemble and eliminate nodes
    // K-1 .. m+1 UPWARDS at the same time, and the middle node m receives both corrections at the end:
    //     bottom node k:   M'_k = Sd_k - Wb'_{k+1} Wb'_{k+1}',  L_k = chol(M'_k),  Wb'_k = So_{k-1}' L_k^-T   (couples k to k-1)
    //     middle:          M_m  = Sd_m - Wb_{m-1} Wb_{m-1}' - Wb'_{m+1} Wb'_{m+1}'
    // The chain is half as long; what the solve does with the factor is in solve_chains.  Tile slot j > m holds
    // N'_{j-1} = -L_{j-1}^-1 Wb'_j UNtransposed, slot j <= m the usual N_j = -L_j^-1 Wb_{j-1} transposed.
    // Round 6: the two-ended form carries the border too, in t-space.  S = L L' holds for the twisted factor as for the plain one, so the
    // Gram identity <r_a, S^-1 r_b> = <t_a, t_b>, t = L^-1 r, does; the twisted forward substitution runs downwards in the top half,
    // upwards in the bottom half and ends at the middle node,
    //     top    j < m:  t_j = L_j^-1 r_j + N_j  t_{j-1}         (N_j  = -L_j^-1 Wb_{j-1},  slot j, transposed)
    //     bottom k > m:  t_k = L_k^-1 r_k + N'_k t_{k+1}         (N'_k = -L_k^-1 Wb'_{k+1}, slot k + 1, untransposed)
    //     middle:        t_m = L_m^-1 r_m + N_m t_{m-1} + N'_m t_{m+1}
    // with r_k = (E Hb^-1 g)_k formed by the assembly wavefront of each half from the tiles it holds (as factor_pipelined does), the two small
    // products of the substitution done by the CHAIN wavefront right after it has formed L_k^-1 and the coupling tile of that node, and the
    // Gram contribution and the stores of t_k by the assembly wavefront a barrier later.  Until round 5
    // this executor back-substituted four border systems after the loop (Hb^-1 twice, a pass over D, S_solveN<4>, a coefficient pass:
    // 17-19 % of a solve at B <= 512).
    // The four roles are four routines (one per wavefront: each gets a register allocation of its own -- as one routine the union of their
    // loop-carried values spilled 50 registers inside the loop), meeting at the same nsteps + 2 workgroup barriers.
    struct TwTiles { double *Sd, *So, *Wp, *Li, *Mq, *Dt, *T, *Bp, *Hh, *Hd, *Mp, *Gn, *Rr, *Tt, *Sg; };
    static SCVX_HD TwTiles tw_tiles(double* sc) {
        TwTiles q;
        q.Sd = sc;                  // 2 x 196 ring, factorised in place
        q.So = q.Sd + 392;          // 2 x 196 ring
        q.Wp = q.So + 392;          // 3 x 196: Wb ring
        q.Li = q.Wp + 588;          // 2 x 196: Linv ring
        q.Mq = q.Li + 392;          // coupling-tile ring, slot 0 (the chain writes the tile of step v into slot v & 1, the assembly reads it a step later)
        q.Dt = q.Mq + 196;          // assembly: D_k tile
        q.T = q.Dt + DSZ;           // assembly: [TA | TBm | TBp], row stride TS
        q.Bp = q.T + 14 * TS;       // assembly: Bp of the neighbouring tile
        q.Hh = q.Bp + BPN;          // assembly: node inverses (two slots)
        q.Hd = q.Hh + 2 * NODE_SZ;  // assembly: dense Hxi (bottom: two of them, by step parity)
        q.Mp = q.Hd + 392;          // coupling-tile ring, slot 1
        q.Gn = q.Mp + 196;          // assembly: 2 x NXU x 4 node slices of the border right-hand sides: slot 0 = node k, slot 1 = node k + 1
        q.Rr = q.Gn + 2 * NXU * 4;  // 3 x 56: r_k ring (written at the node's assembly step, read two steps later)
        q.Tt = q.Rr + 168;          // 2 x 56: the running t of this half
        q.Sg = q.Tt + 112;          // 42 segment scalars gx_nu,k | ry_k | Pnu_k (+ 14: the Gram product's over-read)
        return q;
    }
    static constexpr int kTwTileDoubles = 392 + 392 + 588 + 392 + 196 + DSZ + 14 * TS + BPN + 2 * NODE_SZ + 392 + 196 + 2 * NXU * 4 + 168 + 112 + 42 + 14;
    SCVX_HD double tw_node_elem(int node, int e) const { return e < HX_SZ ? hx[(size_t)node * HX_SZ + e] : hu[HU_SZ * node + (e - HX_SZ)]; }
    // element e = NXU vec + row of node nd (vec 0: Ptr = v1 of the trust-region cone; vec 1: the predictor's gx)
    SCVX_HD double tw_gnode_elem(int nd, int e, bool with_pred) const {
        const int vec = e >= NXU ? 1 : 0, row = e - NXU * vec;
        if (vec == 1 && !with_pred) return 0.0;
        cgptr src = vec ? (cgptr)gx : (cgptr)(Wv + L.o_tr + 1);
        return row < 14 ? src[14 * nd + row] : src[L.nx + NU * nd + (row - 14)];
    }
    SCVX_HD double tw_gseg_elem(int sgk, int e, bool with_pred) const {
        if (e >= 28) return Wv[L.o_nu + 1 + 14 * sgk + (e - 28)];
        if (!with_pred) return 0.0;
        return e < 14 ? gx[L.nx + L.nu_ + 14 * sgk + e] : ry[14 * sgk + (e - 14)];
    }
    // r_k (4 columns: 0 Sg, 1 Ptr, 2 Pnu, 3 predictor) from the tiles the calling assembly wavefront holds: T = [TA | TBm | TBp]_k, Hn = dense
    // Hxi_{k+1}, Gn slots = nodes k | k + 1
    template <class E2>
    SCVX_HD void tw_form_r(typename E2::WAcc& cr, const TwTiles& q, const double* Hn) {
        ex.w_acc_mac(cr, q.T, TS, 1, q.Gn, 4, 1, 14 + NU, 1.0, 4);
        ex.w_acc_mac(cr, q.T + 14 + NU, TS, 1, q.Gn + NXU * 4 + 14 * 4, 4, 1, NU, 1.0, 4);
        ex.w_acc_mac(cr, Hn, 14, 1, q.Gn + NXU * 4, 4, 1, 14, -1.0, 4);
    }
    // column 0 = Sg_k, column 2 = hnui Pnu_k, column 3 += hnui gx_nu,k + ry_k; returns this lane's share of <Pnu_k, gx_nu,k>
    SCVX_HD double tw_plain_r(const TwTiles& q, double* Rk, int l, double hnui_) {
        double g = 0.0;
        if (l < 14) {
            Rk[4 * l] = q.Dt[14 * CS + l];
            Rk[4 * l + 2] = hnui_ * q.Sg[28 + l];
            Rk[4 * l + 3] += hnui_ * q.Sg[l] + q.Sg[14 + l];
            g = q.Sg[28 + l] * q.Sg[l];
        }
        return g;
    }
    // Forward substitution of ring index idx (top: node j; bottom: step v), in two parts.  The CHAIN wavefront, which has just written L^-1
    // and the coupling tile Nt of this node: t = L^-1 r + Nt t_prev (untr: Nt is stored untransposed -- the bottom half) ...
    template <class E2>
    SCVX_HD void tw_fwd_product(const TwTiles& q, int idx, const double* Lik, const double* Nt, bool coupled, bool untr) {
        const double* Rk = q.Rr + 56 * (idx % 3);
        double* Tc = q.Tt + 56 * (idx & 1); const double* Tp = q.Tt + 56 * ((idx + 1) & 1);
        typename E2::WAcc ct;
        ex.w_acc_zero(ct);
        ex.w_acc_mac(ct, Lik, 14, 1, Rk, 4, 1, 14, 1.0, 4);
        if (coupled) {
            if (untr) ex.w_acc_mac(ct, Nt, 14, 1, Tp, 4, 1, 14, 1.0, 4);
            else ex.w_acc_mac(ct, Nt, 1, 14, Tp, 4, 1, 14, 1.0, 4);
        }
        ex.w_acc_store(ct, Tc, 4, 1, false, 4);
    }
    // ... and the ASSEMBLY wavefront, a barrier later: its Gram contribution and the rows of `node` of the four t-vectors
    template <class E2>
    SCVX_HD void tw_fwd_gram_store(typename E2::WAcc& cg, const TwTiles& q, int idx, int node, bool with_pred, int l) {
        const double* Tc = q.Tt + 56 * (idx & 1);
        ex.w_acc_mac(cg, Tc, 1, 4, Tc, 4, 1, 14, 1.0, 4);
        if (l < 56) {
            const int c = l / 14, i = l - 14 * c;
            const gptr dst = c == 0 ? ys : (c == 1 ? ytr : (c == 2 ? ynu : dy));
            if (c < 3 || with_pred) dst[14 * node + i] = Tc[4 * i + c];
        }
    }
    SCVX_HD void tw_store_linv(int k, const double* Lik, int l) {
        const fptr Linv_ = Linv;
        for (int e = l; e < LINV_SZ; e += 64) {
            const int p = e / 15, c = e - 15 * p;
            const int i = c <= p ? p : 13 - p, jj = c <= p ? c : c - (p + 1);
            Linv_[(size_t)k * LINV_SZ + e] = Lik[14 * i + jj];
        }
    }

    // ---- role: top assembly (wavefront 1): Sd_k, r_k, So_k for nodes 0 .. m; Gram matrix and stores of t two nodes behind ----
    template <class E2 = Ex>
    SCVX_HD_NI double tw_top_assembly(bool with_pred) {
        SCVX_THIS_LDS();
        const int K = L.K, m = K / 2, nb = K - 1 - m, l = ex.wlane();
        const int nsteps = (m > nb ? m : nb) + 1;
        const dcptr D_ = D; const double hnui_ = hnui;
        const TwTiles q = tw_tiles(ex.pipe_scratch());
        const int hpos_lane = hx_dense_pos(l <= HX_SZ ? l : 0);
        const int gat = l < 2 * NXU ? 4 * (l >= NXU ? l - NXU : l) + (l >= NXU ? 3 : 1) : 0;   // this lane's element of a Gn slot
        typename E2::WAcc cg;
        ex.w_acc_zero(cg);
        double gnacc = 0.0;
        // prologue: D_0, node 0 -> TA_0, TBm_0 (as factor_pipelined)
        for (int e = l; e < 2 * NXU * 4 + 168 + 112 + 42 + 14; e += 64) q.Gn[e] = 0.0;   // Gn (columns 0 and 2 stay zero), Rr, Tt, Sg
        ex.w_sync_lds();
        if (l < 2 * NXU) q.Gn[NXU * 4 + gat] = tw_gnode_elem(0, l, with_pred);
        for (int e = l; e < DSZ; e += 64) q.Dt[e] = D_[e];
        for (int e = l; e < NODE_SZ; e += 64) q.Hh[NODE_SZ + e] = tw_node_elem(0, e);
        ex.w_sync_lds();
        for (int e = l; e < 196; e += 64) q.Hd[e] = hxi_entry(q.Hh + NODE_SZ, e / 14, e % 14);
        ex.w_sync_lds();
        ex.w_tile_gemm(q.T, TS, 1, q.Dt, 1, 14, q.Hd, 14, 1, 14, 1.0, false);
        for (int c0 = l; c0 < BPN; c0 += 64) {
            const int i = c0 / NU, c = c0 - NU * i;
            q.T[TS * i + 14 + c] = bhu(q.Dt, 14, i, c, q.Hh + NODE_SZ + HX_SZ);
        }
        ex.w_sync_lds();
        double hnext = l < NODE_SZ ? tw_node_elem(1, l) : 0.0;                 // the next node's inverses one step ahead
        double gnext = l < 2 * NXU ? tw_gnode_elem(1, l, with_pred) : 0.0;    // ... and the right-hand sides' slices and segment scalars
        double sgnext = l < 42 ? tw_gseg_elem(0, l, with_pred) : 0.0;
        for (int t = 0; t < nsteps; t++) {
            if (t <= m) {
                const int k = t;
                SCVX_TS(ta_);
                double* Sdk = q.Sd + 196 * (k & 1); double* Sok = q.So + 196 * (k & 1);
                double pre[NPW];
                dcptr Dn = D_ + (size_t)(k + 1 < K ? k + 1 : k) * DSZ;
                SCVX_UNROLL
                for (int c = 0; c < NPW; c++) { const int e = l + 64 * c; pre[c] = e < DSZ ? Dn[e] : 0.0; }
                for (int e = l; e < NODE_SZ; e += 64) { q.Hh[e] = q.Hh[NODE_SZ + e]; }
                if (l < 2 * NXU) q.Gn[gat] = q.Gn[NXU * 4 + gat];   // slot 0 <- slot 1 (node k)
                if (l < 42) q.Sg[l] = sgnext;
                ex.w_sync_lds();
                if (l < NODE_SZ) q.Hh[NODE_SZ + l] = hnext;                            // node k + 1, requested a step ago
                hnext = l < NODE_SZ ? tw_node_elem(k + 2 <= K ? k + 2 : K, l) : 0.0;   // node k + 2 for the next step
                if (l < 2 * NXU) q.Gn[NXU * 4 + gat] = gnext;
                gnext = l < 2 * NXU ? tw_gnode_elem(k + 2 <= K ? k + 2 : K, l, with_pred) : 0.0;
                sgnext = l < 42 ? tw_gseg_elem(k + 1 < K ? k + 1 : k, l, with_pred) : 0.0;
                ex.w_sync_lds();
                if (l <= HX_SZ) q.Hd[hpos_lane] = q.Hh[NODE_SZ + (l < HX_SZ ? l : HX_Q)];   // the 34 non-zeros of the dense Hxi_{k+1}
                for (int c0 = l; c0 < BPN; c0 += 64) {
                    const int i = c0 / NU, c = c0 - NU * i;
                    q.T[TS * i + 14 + NU + c] = bhu(q.Dt, 14 + NU, i, c, q.Hh + NODE_SZ + HX_SZ);
                }
                ex.w_sync_lds();
                {
                    double* Rk = q.Rr + 56 * (k % 3);
                    typename E2::WAcc cm, cr;
                    ex.w_acc_zero(cm); ex.w_acc_zero(cr);
                    ex.w_acc_mac(cm, q.T, TS, 1, q.Dt, 14, 1, TW, 1.0);
                    tw_form_r<E2>(cr, q, q.Hd);
                    ex.w_acc_store_init(cm, Sdk, q.Hd, hnui_);   // Sd_k = Hxi_{k+1} + hnui I + [TA | TBm | TBp]_k D_k'
                    ex.w_acc_store(cr, Rk, 4, 1, false, 4);
                    ex.w_sync_lds();
                    gnacc += tw_plain_r(q, Rk, l, hnui_);
                }
                if (k < m) {
                    for (int c0 = l; c0 < BPN; c0 += 64) q.Bp[c0] = q.Dt[14 * (14 + NU) + c0];
                    ex.w_sync_lds();
                    SCVX_UNROLL
                    for (int c = 0; c < NPW; c++) { const int e = l + 64 * c; if (e < DSZ) q.Dt[e] = pre[c]; }
                    ex.w_sync_lds();
                    ex.w_tile_gemm(q.T, TS, 1, q.Dt, 1, 14, q.Hd, 14, 1, 14, 1.0, false);
                    for (int c0 = l; c0 < BPN; c0 += 64) {
                        const int i = c0 / NU, c = c0 - NU * i;
                        q.T[TS * i + 14 + c] = bhu(q.Dt, 14, i, c, q.Hh + NODE_SZ + HX_SZ);
                    }
                    ex.w_sync_lds();
                    for (int e = l; e < 196; e += 64) {
                        const int i = e / 14, j = e - 14 * i;
                        Sok[e] = so_elem(q.T, q.Bp, i, j);
                    }
                }
                ex.w_sync_lds();
                SCVX_TE(ta_, 24);
            }
            if (t >= 2 && t - 2 <= m - 2) {   // two nodes behind: Gram contribution and stores of t of node t - 2 (formed by the chain a step ago)
                SCVX_TS(tp_);
                tw_fwd_gram_store<E2>(cg, q, t - 2, t - 2, with_pred, l);
                SCVX_TE(tp_, 26);
            }
            SCVX_TS(tbar_);
            ex.sync();
            SCVX_TE(tbar_, 28);
        }
        // the last node of the half, beside the middle node's factorisation
        tw_fwd_gram_store<E2>(cg, q, m - 1, m - 1, with_pred, l);
        ex.w_sync_lds();
        ex.w_acc_store(cg, q.Gn, 4, 1, false, 4);   // this half's Gram matrix (the slices are no longer needed)
        return gnacc;
    }

    // ---- role: bottom assembly (wavefront 3): Sd_k, r_k, So_{k-1} for nodes K-1 .. m+1 upwards; Gram matrix and stores of t two steps behind ----
    template <class E2 = Ex>
    SCVX_HD_NI double tw_bot_assembly(bool with_pred) {
        SCVX_THIS_LDS();
        const int K = L.K, m = K / 2, nb = K - 1 - m, l = ex.wlane();
        const int nsteps = (m > nb ? m : nb) + 1;
        const dcptr D_ = D; const double hnui_ = hnui;
        const TwTiles q = tw_tiles(ex.pipe_scratch2());
        const int hpos_lane = hx_dense_pos(l <= HX_SZ ? l : 0);
        const int gat = l < 2 * NXU ? 4 * (l >= NXU ? l - NXU : l) + (l >= NXU ? 3 : 1) : 0;
        typename E2::WAcc cg;
        ex.w_acc_zero(cg);
        double gnacc = 0.0;
        // prologue: tile K-1, nodes K-1 (slot 0) and K (slot 1), Bp of tile K-2, dense Hxi_K
        for (int e = l; e < 2 * NXU * 4 + 168 + 112 + 42 + 14; e += 64) q.Gn[e] = 0.0;
        ex.w_sync_lds();
        {
            dcptr Dk = D_ + (size_t)(K - 1) * DSZ;
            if (l < 2 * NXU) { q.Gn[gat] = tw_gnode_elem(K - 1, l, with_pred); q.Gn[NXU * 4 + gat] = tw_gnode_elem(K, l, with_pred); }
            if (l < 42) q.Sg[l] = tw_gseg_elem(K - 1, l, with_pred);
            for (int e = l; e < DSZ; e += 64) q.Dt[e] = Dk[e];
            for (int e = l; e < NODE_SZ; e += 64) { q.Hh[e] = tw_node_elem(K - 1, e); q.Hh[NODE_SZ + e] = tw_node_elem(K, e); }
            for (int c0 = l; c0 < BPN; c0 += 64) q.Bp[c0] = D_[(size_t)(K - 2) * DSZ + 14 * (14 + NU) + c0];
            ex.w_sync_lds();
            for (int e = l; e < 196; e += 64) { q.Hd[196 + e] = hxi_entry(q.Hh + NODE_SZ, e / 14, e % 14); q.Hd[e] = 0.0; }   // parity 1 = "step -1"; parity 0: the structural zeros
            ex.w_sync_lds();
        }
        for (int t = 0; t < nsteps; t++) {
            if (t < nb) {
                const int u = t, k = K - 1 - u;
                SCVX_TS(tc_);
                double* Sdk = q.Sd + 196 * (u & 1); double* Sok = q.So + 196 * (u & 1);
                double* Hd0 = q.Hd + 196 * (u & 1);            // dense Hxi_k
                const double* Hd1 = q.Hd + 196 * ((u + 1) & 1);   // dense Hxi_{k+1}: the previous step's Hd0
                // the next step's inputs into registers: tile k-1, node k-1, Bp of tile k-2 (indices clamped at the end of the half)
                const int kn = k - 1 > m ? k - 1 : k, kb2 = kn - 1;
                double pre[NPW];
                dcptr Dn = D_ + (size_t)kn * DSZ;
                SCVX_UNROLL
                for (int c = 0; c < NPW; c++) { const int e = l + 64 * c; pre[c] = e < DSZ ? Dn[e] : 0.0; }
                const double hn = l < NODE_SZ ? tw_node_elem(kn, l) : 0.0;
                const double gn = l < 2 * NXU ? tw_gnode_elem(kn, l, with_pred) : 0.0;
                const double sgn = l < 42 ? tw_gseg_elem(kn, l, with_pred) : 0.0;
                double bpn[(BPN + 63) / 64];
                SCVX_UNROLL
                for (int c = 0; c < (BPN + 63) / 64; c++) { const int e = l + 64 * c; bpn[c] = e < BPN ? (double)D_[(size_t)kb2 * DSZ + 14 * (14 + NU) + e] : 0.0; }
                if (l <= HX_SZ) Hd0[hpos_lane] = q.Hh[l < HX_SZ ? l : HX_Q];   // dense Hxi_k: its 34 non-zeros (both parities were zeroed once)
                ex.w_sync_lds();
                ex.w_tile_gemm(q.T, TS, 1, q.Dt, 1, 14, Hd0, 14, 1, 14, 1.0, false);     // TA_k = A_k Hxi_k
                for (int c0 = l; c0 < 2 * BPN; c0 += 64) {
                    const bool pls = c0 >= BPN;
                    const int qq = pls ? c0 - BPN : c0, i = qq / NU, c = qq - NU * i;
                    const double* h = q.Hh + (pls ? NODE_SZ : 0) + HX_SZ;            // Hui_k for TBm_k, Hui_{k+1} for TBp_k
                    const int cc = pls ? 14 + NU : 14;
                    q.T[TS * i + cc + c] = bhu(q.Dt, cc, i, c, h);
                }
                ex.w_sync_lds();
                {
                    double* Rk = q.Rr + 56 * (u % 3);
                    typename E2::WAcc cm, cr;
                    ex.w_acc_zero(cm); ex.w_acc_zero(cr);
                    ex.w_acc_mac(cm, q.T, TS, 1, q.Dt, 14, 1, TW, 1.0);
                    tw_form_r<E2>(cr, q, Hd1);
                    ex.w_acc_store_init(cm, Sdk, Hd1, hnui_);   // Sd_k = Hxi_{k+1} + hnui I + [TA | TBm | TBp]_k D_k'
                    ex.w_acc_store(cr, Rk, 4, 1, false, 4);
                    ex.w_sync_lds();
                    gnacc += tw_plain_r(q, Rk, l, hnui_);
                }
                for (int e = l; e < 196; e += 64) {
                    const int i = e / 14, j = e - 14 * i;
                    Sok[e] = so_elem(q.T, q.Bp, i, j);   // So_{k-1}
                }
                ex.w_sync_lds();
                // rotate: node k becomes "k+1" of the next step, the prefetched tile / node / Bp / right-hand-side slices move in
                if (l < NODE_SZ) { q.Hh[NODE_SZ + l] = q.Hh[l]; }
                if (l < 2 * NXU) q.Gn[NXU * 4 + gat] = q.Gn[gat];
                ex.w_sync_lds();
                if (l < NODE_SZ) q.Hh[l] = hn;
                if (l < 2 * NXU) q.Gn[gat] = gn;
                if (l < 42) q.Sg[l] = sgn;
                SCVX_UNROLL
                for (int c = 0; c < (BPN + 63) / 64; c++) { const int e = l + 64 * c; if (e < BPN) q.Bp[e] = bpn[c]; }
                SCVX_UNROLL
                for (int c = 0; c < NPW; c++) { const int e = l + 64 * c; if (e < DSZ) q.Dt[e] = pre[c]; }
                ex.w_sync_lds();
                SCVX_TE(tc_, 24);
            }
            if (t >= 2 && t - 2 <= nb - 1) {   // two steps behind: Gram contribution and stores of t of step v = t - 2
                SCVX_TS(tq_);
                tw_fwd_gram_store<E2>(cg, q, t - 2, K - 1 - (t - 2), with_pred, l);
                SCVX_TE(tq_, 26);
            }
            SCVX_TS(tbar_);
            ex.sync();
            SCVX_TE(tbar_, 28);
        }
        if (nb - 1 > nsteps - 3) {   // the last node of the half, if the loop did not reach it
            tw_fwd_gram_store<E2>(cg, q, nb - 1, m + 1, with_pred, l);
        }
        ex.w_sync_lds();
        ex.w_acc_store(cg, q.Gn, 4, 1, false, 4);
        return gnacc;
    }

    // ---- role: the two chains (wavefront 0: nodes 0 .. m-1 downwards, then the middle node; wavefront 2: nodes K-1 .. m+1 upwards) ----
    template <class E2 = Ex>
    SCVX_HD_NI bool tw_chain(bool bot) {
        SCVX_THIS_LDS();
        const int K = L.K, m = K / 2, nb = K - 1 - m, l = ex.wlane();
        const int nsteps = (m > nb ? m : nb) + 1;
        const fptr Nf_ = Nf;
        const TwTiles q = tw_tiles(bot ? ex.pipe_scratch2() : ex.pipe_scratch());
        const int nn = bot ? nb : m;   // nodes of this half
        bool ok = true;
        for (int t = 0; t < nsteps; t++) {
            if (t >= 1 && t <= nn) {
                // step v = t - 1: top node k = v, bottom node k = K-1-v
                const int v = t - 1, k = bot ? K - 1 - v : v;
                SCVX_TS(tb0_);
                double* M = q.Sd + 196 * (v & 1); const double* Sok = q.So + 196 * (v & 1);   // top: So_k; bottom: So_{k-1}
                double* Lik = q.Li + 196 * (v & 1);
                const double* Wpm = q.Wp + 196 * ((v + 2) % 3);   // top: Wb_{k-1}; bottom: Wb'_{k+1}
                if (v > 0) { ex.w_tile_gemm(M, 14, 1, Wpm, 14, 1, Wpm, 1, 14, 14, -1.0, true); ex.w_sync_lds(); }
                ok = ex.w_chol_inv14(M, Lik) && ok;
                ex.w_sync_lds();
                if (bot) ex.w_tile_gemm(q.Wp + 196 * (v % 3), 14, 1, Sok, 1, 14, Lik, 1, 14, 14, 1.0, false);   // Wb'_k = So_{k-1}' L_k^-T
                else ex.w_tile_gemm(q.Wp + 196 * (v % 3), 14, 1, Sok, 14, 1, Lik, 1, 14, 14, 1.0, false);       // Wb_k = So_k L_k^-T
                tw_store_linv(k, Lik, l);
                if (v > 0) {
                    // coupling tile of this node into the two-slot ring (the assembly wavefront's forward substitution reads it a step later)
                    // and to the factor: top N_k = -L_k^-1 Wb_{k-1}, transposed, slot k; bottom N'_k = -L_k^-1 Wb'_{k+1}, untransposed, slot k + 1
                    double* Nt = (v & 1) ? q.Mp : q.Mq;
                    if (bot) ex.w_tile_gemm(Nt, 14, 1, Lik, 14, 1, Wpm, 14, 1, 14, -1.0, false);
                    else ex.w_tile_gemm(Nt, 1, 14, Lik, 14, 1, Wpm, 14, 1, 14, -1.0, false);
                    ex.w_sync_lds();
                    const size_t slot = bot ? (size_t)(k + 1) : (size_t)k;
                    for (int e = l; e < 196; e += 64) Nf_[slot * 196 + e] = Nt[e];
                }
                // t of this node (r_k: the assembly wavefront's, a barrier ago); its Gram contribution and stores are the assembly's, a barrier on
                tw_fwd_product<E2>(q, v, Lik, (v & 1) ? q.Mp : q.Mq, v > 0, bot);
                ex.w_sync_lds();
                SCVX_TE(tb0_, 24);
            }
            SCVX_TS(tbar_);
            ex.sync();
            SCVX_TE(tbar_, 28);
        }
        if (!bot) {
            // ---- the middle node: both corrections, its two coupling tiles ----
            SCVX_TS(tm_);
            double* sc2 = ex.pipe_scratch2();
            double* Na_ = q.So;          // N_m and N'_m: the So ring, free since this chain's last step (the coupling-tile ring is still being
            double* Nb_ = q.So + 196;    // read by the top assembly's last forward substitution)
            const double* WbB = sc2 + 784 + 196 * ((nb - 1) % 3);    // Wb'_{m+1}: the bottom chain's last coupling tile
            const double* WbT = q.Wp + 196 * ((m - 1) % 3);            // Wb_{m-1}
            double* M = q.Sd + 196 * (m & 1);
            double* Lik = q.Li + 196 * (m & 1);
            ex.w_tile_gemm(M, 14, 1, WbT, 14, 1, WbT, 1, 14, 14, -1.0, true);
            ex.w_sync_lds();
            ex.w_tile_gemm(M, 14, 1, WbB, 14, 1, WbB, 1, 14, 14, -1.0, true);
            ex.w_sync_lds();
            ok = ex.w_chol_inv14(M, Lik) && ok;
            ex.w_sync_lds();
            tw_store_linv(m, Lik, l);
            ex.w_tile_gemm(Na_, 1, 14, Lik, 14, 1, WbT, 14, 1, 14, -1.0, false);      // N_m, transposed, slot m
            ex.w_tile_gemm(Nb_, 14, 1, Lik, 14, 1, WbB, 14, 1, 14, -1.0, false);      // N'_m, untransposed, slot m + 1
            ex.w_sync_lds();
            for (int e = l; e < 196; e += 64) { Nf_[(size_t)m * 196 + e] = Na_[e]; Nf_[(size_t)(m + 1) * 196 + e] = Nb_[e]; }
            SCVX_TE(tm_, 27);
        }
        return ok;
    }

    // ---- the two-ended form for TWO wavefronts per trajectory (B <= 4 per CU; SCVX_K4_TWISTED2): each wavefront runs ONE half on its own --
    // assembly, chain, coupling tile, substitution and Gram contribution of a node one after the other, no hand-over barrier, no rings -- so
    // the sequential part is 26 node steps instead of the 51 of the assembly / chain pipeline (factor_pipelined).  The tile set of a half
    // is COMPACT (2,068 doubles at control_dim 3: both halves and the solver frames fit the 40 KB a block may take when four share a CU):
    // the pivot tile is dead once chol_inv14 has read it, so the coupling tile takes its place; Wb_k overwrites Wb_{k-1} in place after the
    // coupling tile has been formed; one dense Hxi tile, re-scattered (same 34 non-zeros) when the bottom half needs the other node's.
    struct TwC { double *Sd, *So, *Wb, *Li, *Dt, *T, *Bp, *Hh, *Hd, *Gn, *Rk, *Tt, *Sg; };
    static SCVX_HD TwC twc_tiles(double* sc) {
        TwC q;
        q.Sd = sc;                  // pivot tile, then the coupling tile N of the same node
        q.So = q.Sd + 196;
        q.Wb = q.So + 196;          // Wb of the previous node, replaced by this node's
        q.Li = q.Wb + 196;
        q.Dt = q.Li + 196;
        q.T = q.Dt + DSZ;
        q.Bp = q.T + 14 * TS;
        q.Hh = q.Bp + BPN;
        q.Hd = q.Hh + 2 * NODE_SZ;
        q.Gn = q.Hd + 196;
        q.Rk = q.Gn + 2 * NXU * 4;
        q.Tt = q.Rk + 56;           // 2 x 56: t of the previous node | of this one, by step parity
        q.Sg = q.Tt + 112;          // 42 (+ 14: the Gram product's over-read)
        return q;
    }
    static constexpr int kTwCDoubles = 4 * 196 + DSZ + 14 * TS + BPN + 2 * NODE_SZ + 196 + 2 * NXU * 4 + 56 + 112 + 42 + 14;
    // chain part of node step v of a half (top: node v; bottom: node K-1-v): pivot update, Cholesky + inverse, L^-1 to HBM, coupling tile (into
    // the pivot tile's place) to HBM, this node's Wb in place of the previous one's, t = L^-1 r + N t_prev, Gram contribution, rows of t to HBM
    template <class E2>
    SCVX_HD bool twc_chain_step(typename E2::WAcc& cg, const TwC& q, int v, int node, bool bot, bool last, bool with_pred, int l) {
        const fptr Nf_ = Nf;
        double* M = q.Sd;
        if (v > 0) { ex.w_tile_gemm(M, 14, 1, q.Wb, 14, 1, q.Wb, 1, 14, 14, -1.0, true); ex.w_sync_lds(); }
        const bool ok = ex.w_chol_inv14(M, q.Li);
        ex.w_sync_lds();
        tw_store_linv(node, q.Li, l);
        if (v > 0) {
            if (bot) ex.w_tile_gemm(M, 14, 1, q.Li, 14, 1, q.Wb, 14, 1, 14, -1.0, false);   // N'_k = -L_k^-1 Wb'_{k+1}, untransposed, slot k + 1
            else ex.w_tile_gemm(M, 1, 14, q.Li, 14, 1, q.Wb, 14, 1, 14, -1.0, false);       // N_k = -L_k^-1 Wb_{k-1}, transposed, slot k
            ex.w_sync_lds();
            const size_t slot = bot ? (size_t)(node + 1) : (size_t)node;
            for (int e = l; e < 196; e += 64) Nf_[slot * 196 + e] = M[e];
        }
        {
            double* Tc = q.Tt + 56 * (v & 1); const double* Tp = q.Tt + 56 * ((v + 1) & 1);
            typename E2::WAcc ct;
            ex.w_acc_zero(ct);
            ex.w_acc_mac(ct, q.Li, 14, 1, q.Rk, 4, 1, 14, 1.0, 4);
            if (v > 0) {
                if (bot) ex.w_acc_mac(ct, M, 14, 1, Tp, 4, 1, 14, 1.0, 4);
                else ex.w_acc_mac(ct, M, 1, 14, Tp, 4, 1, 14, 1.0, 4);
            }
            ex.w_acc_store(ct, Tc, 4, 1, false, 4);
            ex.w_sync_lds();
            ex.w_acc_mac(cg, Tc, 1, 4, Tc, 4, 1, 14, 1.0, 4);
            if (l < 56) {
                const int c = l / 14, i = l - 14 * c;
                const gptr dst = c == 0 ? ys : (c == 1 ? ytr : (c == 2 ? ynu : dy));
                if (c < 3 || with_pred) dst[14 * node + i] = Tc[4 * i + c];
            }
        }
        if (!last) {   // this node's Wb = So L^-T (top: So_k; bottom: So_{k-1}'), in place of the previous node's
            if (bot) ex.w_tile_gemm(q.Wb, 14, 1, q.So, 1, 14, q.Li, 1, 14, 14, 1.0, false);
            else ex.w_tile_gemm(q.Wb, 14, 1, q.So, 14, 1, q.Li, 1, 14, 14, 1.0, false);
        }
        ex.w_sync_lds();
        return ok;
    }
    // top half on one wavefront: nodes 0 .. m-1 (assembly + chain each), then the assembly of the middle node (Sd_m, r_m)
    template <class E2 = Ex>
    SCVX_HD_NI bool twc_top(bool with_pred, double& gnacc_out) {
        SCVX_THIS_LDS();
        const int K = L.K, m = K / 2, l = ex.wlane();
        const dcptr D_ = D; const double hnui_ = hnui;
        const TwC q = twc_tiles(ex.pipe_scratch());
        const int hpos_lane = hx_dense_pos(l <= HX_SZ ? l : 0);
        const int gat = l < 2 * NXU ? 4 * (l >= NXU ? l - NXU : l) + (l >= NXU ? 3 : 1) : 0;
        typename E2::WAcc cg;
        ex.w_acc_zero(cg);
        double gnacc = 0.0;
        bool ok = true;
        for (int e = l; e < 2 * NXU * 4 + 56 + 112 + 42 + 14; e += 64) q.Gn[e] = 0.0;
        ex.w_sync_lds();
        if (l < 2 * NXU) q.Gn[NXU * 4 + gat] = tw_gnode_elem(0, l, with_pred);
        for (int e = l; e < DSZ; e += 64) q.Dt[e] = D_[e];
        for (int e = l; e < NODE_SZ; e += 64) q.Hh[NODE_SZ + e] = tw_node_elem(0, e);
        ex.w_sync_lds();
        for (int e = l; e < 196; e += 64) q.Hd[e] = hxi_entry(q.Hh + NODE_SZ, e / 14, e % 14);
        ex.w_sync_lds();
        ex.w_tile_gemm(q.T, TS, 1, q.Dt, 1, 14, q.Hd, 14, 1, 14, 1.0, false);
        for (int c0 = l; c0 < BPN; c0 += 64) {
            const int i = c0 / NU, c = c0 - NU * i;
            q.T[TS * i + 14 + c] = bhu(q.Dt, 14, i, c, q.Hh + NODE_SZ + HX_SZ);
        }
        ex.w_sync_lds();
        double hnext = l < NODE_SZ ? tw_node_elem(1, l) : 0.0;
        double gnext = l < 2 * NXU ? tw_gnode_elem(1, l, with_pred) : 0.0;
        double sgnext = l < 42 ? tw_gseg_elem(0, l, with_pred) : 0.0;
        for (int k = 0; k <= m; k++) {
            SCVX_TS(ta_);
            double pre[NPW];
            dcptr Dn = D_ + (size_t)(k + 1 < K ? k + 1 : k) * DSZ;
            SCVX_UNROLL
            for (int c = 0; c < NPW; c++) { const int e = l + 64 * c; pre[c] = e < DSZ ? Dn[e] : 0.0; }
            for (int e = l; e < NODE_SZ; e += 64) { q.Hh[e] = q.Hh[NODE_SZ + e]; }
            if (l < 2 * NXU) q.Gn[gat] = q.Gn[NXU * 4 + gat];
            if (l < 42) q.Sg[l] = sgnext;
            ex.w_sync_lds();
            if (l < NODE_SZ) q.Hh[NODE_SZ + l] = hnext;
            hnext = l < NODE_SZ ? tw_node_elem(k + 2 <= K ? k + 2 : K, l) : 0.0;
            if (l < 2 * NXU) q.Gn[NXU * 4 + gat] = gnext;
            gnext = l < 2 * NXU ? tw_gnode_elem(k + 2 <= K ? k + 2 : K, l, with_pred) : 0.0;
            sgnext = l < 42 ? tw_gseg_elem(k + 1 < K ? k + 1 : k, l, with_pred) : 0.0;
            ex.w_sync_lds();
            if (l <= HX_SZ) q.Hd[hpos_lane] = q.Hh[NODE_SZ + (l < HX_SZ ? l : HX_Q)];   // dense Hxi_{k+1}
            for (int c0 = l; c0 < BPN; c0 += 64) {
                const int i = c0 / NU, c = c0 - NU * i;
                q.T[TS * i + 14 + NU + c] = bhu(q.Dt, 14 + NU, i, c, q.Hh + NODE_SZ + HX_SZ);
            }
            ex.w_sync_lds();
            {
                typename E2::WAcc cm, cr;
                ex.w_acc_zero(cm); ex.w_acc_zero(cr);
                ex.w_acc_mac(cm, q.T, TS, 1, q.Dt, 14, 1, TW, 1.0);
                ex.w_acc_mac(cr, q.T, TS, 1, q.Gn, 4, 1, 14 + NU, 1.0, 4);
                ex.w_acc_mac(cr, q.T + 14 + NU, TS, 1, q.Gn + NXU * 4 + 14 * 4, 4, 1, NU, 1.0, 4);
                ex.w_acc_mac(cr, q.Hd, 14, 1, q.Gn + NXU * 4, 4, 1, 14, -1.0, 4);
                ex.w_acc_store_init(cm, q.Sd, q.Hd, hnui_);
                ex.w_acc_store(cr, q.Rk, 4, 1, false, 4);
                ex.w_sync_lds();
                if (l < 14) {
                    q.Rk[4 * l] = q.Dt[14 * CS + l];
                    q.Rk[4 * l + 2] = hnui_ * q.Sg[28 + l];
                    q.Rk[4 * l + 3] += hnui_ * q.Sg[l] + q.Sg[14 + l];
                    gnacc += q.Sg[28 + l] * q.Sg[l];
                }
            }
            if (k < m) {
                for (int c0 = l; c0 < BPN; c0 += 64) q.Bp[c0] = q.Dt[14 * (14 + NU) + c0];
                ex.w_sync_lds();
                SCVX_UNROLL
                for (int c = 0; c < NPW; c++) { const int e = l + 64 * c; if (e < DSZ) q.Dt[e] = pre[c]; }
                ex.w_sync_lds();
                ex.w_tile_gemm(q.T, TS, 1, q.Dt, 1, 14, q.Hd, 14, 1, 14, 1.0, false);
                for (int c0 = l; c0 < BPN; c0 += 64) {
                    const int i = c0 / NU, c = c0 - NU * i;
                    q.T[TS * i + 14 + c] = bhu(q.Dt, 14, i, c, q.Hh + NODE_SZ + HX_SZ);
                }
                ex.w_sync_lds();
                for (int e = l; e < 196; e += 64) {
                    const int i = e / 14, j = e - 14 * i;
                    q.So[e] = so_elem(q.T, q.Bp, i, j);
                }
            }
            ex.w_sync_lds();
            SCVX_TE(ta_, 24);
            if (k < m) {
                SCVX_TS(tb_);
                ok = twc_chain_step<E2>(cg, q, k, k, false, false, with_pred, l) && ok;   // (Wb_{m-1} is needed by the middle node: never `last`)
                SCVX_TE(tb_, 26);
            }
        }
        ex.w_sync_lds();
        ex.w_acc_store(cg, q.Gn, 4, 1, false, 4);   // this half's Gram matrix
        gnacc_out = gnacc;
        return ok;
    }
    // bottom half on one wavefront: nodes K-1 .. m+1 upwards
    template <class E2 = Ex>
    SCVX_HD_NI bool twc_bot(bool with_pred, double& gnacc_out) {
        SCVX_THIS_LDS();
        const int K = L.K, m = K / 2, nb = K - 1 - m, l = ex.wlane();
        const dcptr D_ = D; const double hnui_ = hnui;
        const TwC q = twc_tiles(ex.pipe_scratch() + kTwCDoubles);
        const int hpos_lane = hx_dense_pos(l <= HX_SZ ? l : 0);
        const int gat = l < 2 * NXU ? 4 * (l >= NXU ? l - NXU : l) + (l >= NXU ? 3 : 1) : 0;
        typename E2::WAcc cg;
        ex.w_acc_zero(cg);
        double gnacc = 0.0;
        bool ok = true;
        for (int e = l; e < 2 * NXU * 4 + 56 + 112 + 42 + 14; e += 64) q.Gn[e] = 0.0;
        for (int e = l; e < 196; e += 64) q.Hd[e] = 0.0;   // the structural zeros of a dense Hxi tile
        ex.w_sync_lds();
        {
            dcptr Dk = D_ + (size_t)(K - 1) * DSZ;
            if (l < 2 * NXU) { q.Gn[gat] = tw_gnode_elem(K - 1, l, with_pred); q.Gn[NXU * 4 + gat] = tw_gnode_elem(K, l, with_pred); }
            if (l < 42) q.Sg[l] = tw_gseg_elem(K - 1, l, with_pred);
            for (int e = l; e < DSZ; e += 64) q.Dt[e] = Dk[e];
            for (int e = l; e < NODE_SZ; e += 64) { q.Hh[e] = tw_node_elem(K - 1, e); q.Hh[NODE_SZ + e] = tw_node_elem(K, e); }
            for (int c0 = l; c0 < BPN; c0 += 64) q.Bp[c0] = D_[(size_t)(K - 2) * DSZ + 14 * (14 + NU) + c0];
            ex.w_sync_lds();
        }
        for (int u = 0; u < nb; u++) {
            const int k = K - 1 - u;
            SCVX_TS(tc_);
            const int kn = k - 1 > m ? k - 1 : k, kb2 = kn - 1;
            double pre[NPW];
            dcptr Dn = D_ + (size_t)kn * DSZ;
            SCVX_UNROLL
            for (int c = 0; c < NPW; c++) { const int e = l + 64 * c; pre[c] = e < DSZ ? Dn[e] : 0.0; }
            const double hn = l < NODE_SZ ? tw_node_elem(kn, l) : 0.0;
            const double gn = l < 2 * NXU ? tw_gnode_elem(kn, l, with_pred) : 0.0;
            const double sgn = l < 42 ? tw_gseg_elem(kn, l, with_pred) : 0.0;
            double bpn[(BPN + 63) / 64];
            SCVX_UNROLL
            for (int c = 0; c < (BPN + 63) / 64; c++) { const int e = l + 64 * c; bpn[c] = e < BPN ? (double)D_[(size_t)kb2 * DSZ + 14 * (14 + NU) + e] : 0.0; }
            if (l <= HX_SZ) q.Hd[hpos_lane] = q.Hh[l < HX_SZ ? l : HX_Q];   // dense Hxi_k
            ex.w_sync_lds();
            ex.w_tile_gemm(q.T, TS, 1, q.Dt, 1, 14, q.Hd, 14, 1, 14, 1.0, false);     // TA_k = A_k Hxi_k
            for (int c0 = l; c0 < 2 * BPN; c0 += 64) {
                const bool pls = c0 >= BPN;
                const int qq = pls ? c0 - BPN : c0, i = qq / NU, c = qq - NU * i;
                const double* h = q.Hh + (pls ? NODE_SZ : 0) + HX_SZ;
                const int cc = pls ? 14 + NU : 14;
                q.T[TS * i + cc + c] = bhu(q.Dt, cc, i, c, h);
            }
            ex.w_sync_lds();
            if (l <= HX_SZ) q.Hd[hpos_lane] = q.Hh[NODE_SZ + (l < HX_SZ ? l : HX_Q)];   // ... re-scattered: dense Hxi_{k+1} (same non-zeros)
            ex.w_sync_lds();
            {
                typename E2::WAcc cm, cr;
                ex.w_acc_zero(cm); ex.w_acc_zero(cr);
                ex.w_acc_mac(cm, q.T, TS, 1, q.Dt, 14, 1, TW, 1.0);
                ex.w_acc_mac(cr, q.T, TS, 1, q.Gn, 4, 1, 14 + NU, 1.0, 4);
                ex.w_acc_mac(cr, q.T + 14 + NU, TS, 1, q.Gn + NXU * 4 + 14 * 4, 4, 1, NU, 1.0, 4);
                ex.w_acc_mac(cr, q.Hd, 14, 1, q.Gn + NXU * 4, 4, 1, 14, -1.0, 4);
                ex.w_acc_store_init(cm, q.Sd, q.Hd, hnui_);
                ex.w_acc_store(cr, q.Rk, 4, 1, false, 4);
                ex.w_sync_lds();
                if (l < 14) {
                    q.Rk[4 * l] = q.Dt[14 * CS + l];
                    q.Rk[4 * l + 2] = hnui_ * q.Sg[28 + l];
                    q.Rk[4 * l + 3] += hnui_ * q.Sg[l] + q.Sg[14 + l];
                    gnacc += q.Sg[28 + l] * q.Sg[l];
                }
            }
            for (int e = l; e < 196; e += 64) {
                const int i = e / 14, j = e - 14 * i;
                q.So[e] = so_elem(q.T, q.Bp, i, j);   // So_{k-1}
            }
            ex.w_sync_lds();
            if (l < NODE_SZ) { q.Hh[NODE_SZ + l] = q.Hh[l]; }
            if (l < 2 * NXU) q.Gn[NXU * 4 + gat] = q.Gn[gat];
            ex.w_sync_lds();
            if (l < NODE_SZ) q.Hh[l] = hn;
            if (l < 2 * NXU) q.Gn[gat] = gn;
            SCVX_UNROLL
            for (int c = 0; c < (BPN + 63) / 64; c++) { const int e = l + 64 * c; if (e < BPN) q.Bp[e] = bpn[c]; }
            // (Dt and Sg still hold this node's tile and scalars: the chain step below does not read them, the swap can go first)
            SCVX_UNROLL
            for (int c = 0; c < NPW; c++) { const int e = l + 64 * c; if (e < DSZ) q.Dt[e] = pre[c]; }
            if (l < 42) q.Sg[l] = sgn;
            ex.w_sync_lds();
            SCVX_TE(tc_, 24);
            SCVX_TS(td_);
            ok = twc_chain_step<E2>(cg, q, u, k, true, false, with_pred, l) && ok;   // (Wb'_{m+1} is needed by the middle node)
            SCVX_TE(td_, 26);
        }
        ex.w_sync_lds();
        ex.w_acc_store(cg, q.Gn, 4, 1, false, 4);
        gnacc_out = gnacc;
        return ok;
    }
    template <class E2 = Ex>
    SCVX_HD bool factor_twisted2(bool with_pred) {
        static_assert(2 * kTwCDoubles <= E2::kPipe1Doubles, "the compact tile sets of both halves share the first pipeline symbol");
        const int K = L.K, m = K / 2, nb = K - 1 - m;
        const int w = ex.wave(), l = ex.wlane();
        bool ok = true;
        double gnacc = 0.0;
        if (w == 0) ok = twc_top<E2>(with_pred, gnacc);
        else ok = twc_bot<E2>(with_pred, gnacc);
        ex.sync();   // both halves are complete: Sd_m, r_m, Wb_{m-1}, t_{m-1} (top set); Wb'_{m+1}, t_{m+1} (bottom set)
        const TwC q1 = twc_tiles(ex.pipe_scratch()), q2 = twc_tiles(ex.pipe_scratch() + kTwCDoubles);
        double* Gm = q1.T;   // free now: the middle node's t (56) and its Gram matrix (at + 64)
        if (w == 0) {
            SCVX_TS(tm_);
            const fptr Nf_ = Nf;
            double* M = q1.Sd; double* Lik = q1.Li;
            double* Na_ = q1.So;          // N_m, transposed, slot m
            double* Nb_ = q1.Dt;          // N'_m, untransposed, slot m + 1 (the tile buffer is free)
            ex.w_tile_gemm(M, 14, 1, q1.Wb, 14, 1, q1.Wb, 1, 14, 14, -1.0, true);
            ex.w_sync_lds();
            ex.w_tile_gemm(M, 14, 1, q2.Wb, 14, 1, q2.Wb, 1, 14, 14, -1.0, true);
            ex.w_sync_lds();
            ok = ex.w_chol_inv14(M, Lik) && ok;
            ex.w_sync_lds();
            tw_store_linv(m, Lik, l);
            ex.w_tile_gemm(Na_, 1, 14, Lik, 14, 1, q1.Wb, 14, 1, 14, -1.0, false);
            ex.w_tile_gemm(Nb_, 14, 1, Lik, 14, 1, q2.Wb, 14, 1, 14, -1.0, false);
            ex.w_sync_lds();
            for (int e = l; e < 196; e += 64) { Nf_[(size_t)m * 196 + e] = Na_[e]; Nf_[(size_t)(m + 1) * 196 + e] = Nb_[e]; }
            const double* Ta = q1.Tt + 56 * ((m - 1) & 1);       // t_{m-1}
            const double* Tb = q2.Tt + 56 * ((nb - 1) & 1);      // t_{m+1}
            typename E2::WAcc ct, cgm;
            ex.w_acc_zero(ct); ex.w_acc_zero(cgm);
            ex.w_acc_mac(ct, Lik, 14, 1, q1.Rk, 4, 1, 14, 1.0, 4);
            ex.w_acc_mac(ct, Na_, 1, 14, Ta, 4, 1, 14, 1.0, 4);
            ex.w_acc_mac(ct, Nb_, 14, 1, Tb, 4, 1, 14, 1.0, 4);
            for (int e = l; e < 128; e += 64) Gm[e] = 0.0;
            ex.w_sync_lds();
            ex.w_acc_store(ct, Gm, 4, 1, false, 4);
            ex.w_sync_lds();
            ex.w_acc_mac(cgm, Gm, 1, 4, Gm, 4, 1, 14, 1.0, 4);
            if (l < 56) {
                const int c = l / 14, i = l - 14 * c;
                const gptr dst = c == 0 ? ys : (c == 1 ? ytr : (c == 2 ? ynu : dy));
                if (c < 3 || with_pred) dst[14 * m + i] = Gm[4 * i + c];
            }
            ex.w_sync_lds();
            ex.w_acc_store(cgm, Gm + 64, 4, 1, false, 4);
            SCVX_TE(tm_, 27);
        }
        ex.sync();
        for (int c = 0; c < 16; c++) gram[c] = (q1.Gn[c] + q2.Gn[c]) + Gm[64 + c];
        gn_pred = ex.sum(gnacc);
        return ex.all(ok);
    }

    template <class E2 = Ex>
    SCVX_HD bool factor_twisted(bool with_pred) {
        if constexpr (E2::kLanes == 128) return factor_twisted2<E2>(with_pred);
        else {
        static_assert(E2::kPipeDoubles >= kTwTileDoubles, "two-ended tile sets");
        const int K = L.K, m = K / 2, nb = K - 1 - m;
        const int w = ex.wave(), l = ex.wlane();
        bool ok = true;
        double gnacc = 0.0;
        if (w == 0) ok = tw_chain<E2>(false);
        else if (w == 1) gnacc = tw_top_assembly<E2>(with_pred);
        else if (w == 2) ok = tw_chain<E2>(true);
        else gnacc = tw_bot_assembly<E2>(with_pred);
        ex.sync();   // t_{m-1} (top assembly), t_{m+1} (bottom assembly) and the middle factor are complete
        const TwTiles q1 = tw_tiles(ex.pipe_scratch()), q2 = tw_tiles(ex.pipe_scratch2());
        double* Gm = q1.T;   // the top half's T tile: free now; the middle node's t (56) and its Gram matrix (at + 64)
        if (w == 0) {
            const double* Rm = q1.Rr + 56 * (m % 3);             // r_m: formed by the top assembly at step m
            const double* Ta = q1.Tt + 56 * ((m - 1) & 1);       // t_{m-1}
            const double* Tb = q2.Tt + 56 * ((nb - 1) & 1);      // t_{m+1}
            const double* Lik = q1.Li + 196 * (m & 1);
            const double* Na_ = q1.So; const double* Nb_ = q1.So + 196;
            typename E2::WAcc ct, cgm;
            ex.w_acc_zero(ct); ex.w_acc_zero(cgm);
            ex.w_acc_mac(ct, Lik, 14, 1, Rm, 4, 1, 14, 1.0, 4);
            ex.w_acc_mac(ct, Na_, 1, 14, Ta, 4, 1, 14, 1.0, 4);
            ex.w_acc_mac(ct, Nb_, 14, 1, Tb, 4, 1, 14, 1.0, 4);
            for (int e = l; e < 128; e += 64) Gm[e] = 0.0;
            ex.w_sync_lds();
            ex.w_acc_store(ct, Gm, 4, 1, false, 4);
            ex.w_sync_lds();
            ex.w_acc_mac(cgm, Gm, 1, 4, Gm, 4, 1, 14, 1.0, 4);
            if (l < 56) {
                const int c = l / 14, i = l - 14 * c;
                const gptr dst = c == 0 ? ys : (c == 1 ? ytr : (c == 2 ? ynu : dy));
                if (c < 3 || with_pred) dst[14 * m + i] = Gm[4 * i + c];
            }
            ex.w_sync_lds();
            ex.w_acc_store(cgm, Gm + 64, 4, 1, false, 4);
        }
        ex.sync();
        for (int c = 0; c < 16; c++) gram[c] = (q1.Gn[c] + q2.Gn[c]) + Gm[64 + c];
        gn_pred = ex.sum(gnacc);
        return ex.all(ok);
        }
    }

    // ---- factorisation for the current scaling (Wv, Wbeta) ----
    // with_pred: gx holds the (masked) predictor right-hand side and ry the equality residual; their banded solution
    // [Hb E'; E 0][dw; dy] = [gx; -ry] is produced alongside the three border systems (dw, dy), so the predictor's
    // solve adds no pass of its own over the factor and over D.
    // res (sequential loop only, needs with_pred): rx / gx hold their D-independent parts on entry; the loop completes them with E'y,
    // forms ry = E V + dk and leaves |rx_local|^2, |ry|^2, Sg . y in res_nrx2 / res_nry2 / res_sgy (see the loop)
    SCVX_HD_NI bool build_kkt(bool with_pred = false, bool res = false) {
        SCVX_THIS_LDS();
        const int K = L.K;
        SCVX_COUNT(3);
        // big-cone scalars
        {
            const double n1 = bigvv[0], n2 = bigvv[1];   // |v1|^2 of the nu / trust-region cone: summed by the pass that formed the scaling
            soc_w2(Wv[L.o_nu], n1, Wbeta[L.c_nu], h_nu[0], h_nu[1], h_nu[2], h_nu[3]);
            soc_w2(Wv[L.o_tr], n2, Wbeta[L.c_tr], h_tr[0], h_tr[1], h_tr[2], h_tr[3]);
            hnui = 1.0 / h_nu[3];
            q_nu[0] = Wv[L.o_nu]; q_nu[1] = n1; q_tr[0] = Wv[L.o_tr]; q_tr[1] = n2;
            double h00, h01, h11, b2;
            const double vs = Wv[L.o_sg + 1];
            soc_w2(Wv[L.o_sg], vs * vs, Wbeta[L.c_sg], h00, h01, h11, b2);
            q_sg[0] = Wv[L.o_sg]; q_sg[1] = vs; q_sg[2] = b2; q_sg[3] = h01 * vs / h00; h00s = h00;
            hrk = 1.0 / (Wbeta[L.c_rk] * Wbeta[L.c_rk]);
        }
        // node blocks -> compact inverses
        SCVX_T0();
        const double dtr = h_tr[3];
        for (int k = ex.lane(); k <= K; k += ex.nlanes()) {
            gptr h = hx + (size_t)k * HX_SZ;
            for (int i = 0; i < HX_SZ; i++) h[i] = 0.0;
            const bool first = (k == 0), last = (k == K);
            // mass
            if (!first) {
                double hm = dtr;
                const double wm = Wbeta[L.c_mass + (k - 1)];
                hm += 1.0 / (wm * wm);
                h[HX_M] = 1.0 / hm;
            }
            if (!first && !last) {
                // r block with glideslope cone
#if SCVX_NODE_QR
                // M = dtr I + Jc' W^-2 Jc, Jc = diag(1 / tan(gamma), 1, 1): square-root form (see qr_inv3)
                cgptr v = Wv + L.o_gs + 3 * k;
                double h00, h01, h11, b2;
                {
                    const double vv[3] = {v[0], v[1], v[2]};
                    const double ib = 1.0 / Wbeta[L.c_gs + k], sq = sqrt(dtr);
                    double A[6][3] = {{sq, 0, 0}, {0, sq, 0}, {0, 0, sq}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
                    for (int c = 0; c < 3; c++) {
                        const double x[3] = {c == 0 ? C.itan : 0.0, c == 1 ? 1.0 : 0.0, c == 2 ? 1.0 : 0.0};
                        double y[3];
                        soc_Winv_col<3>(vv, ib, x, y);
                        A[3][c] = y[0]; A[4][c] = y[1]; A[5][c] = y[2];
                    }
                    qr_inv3<6>(A, h + HX_R);
                }
#else
                cgptr v = Wv + L.o_gs + 3 * k;
                double h00, h01, h11, b2;
                soc_w2(v[0], v[1] * v[1] + v[2] * v[2], Wbeta[L.c_gs + k], h00, h01, h11, b2);
                double M[9];
                M[0] = dtr + h00 * C.itan * C.itan;
                M[1] = M[3] = h01 * v[1] * C.itan;
                M[2] = M[6] = h01 * v[2] * C.itan;
                M[4] = dtr + b2 + h11 * v[1] * v[1];
                M[5] = M[7] = h11 * v[1] * v[2];
                M[8] = dtr + b2 + h11 * v[2] * v[2];
                inv3(M, h + HX_R);
#endif
                // v block: dtr I, plus the dynamic-pressure cone (vmax; v_k) when it is enforced
                if (k < L.ndp) {
                    cgptr vd = Wv + L.o_dp + 4 * k;
                    soc_w2(vd[0], vd[1] * vd[1] + vd[2] * vd[2] + vd[3] * vd[3], Wbeta[L.c_dp + k], h00, h01, h11, b2);
                    inv_iso_rank1_3(dtr + b2, h11, vd[1], vd[2], vd[3], h + HX_V);
                } else {
                    h[HX_V] = h[HX_V + 4] = h[HX_V + 8] = 1.0 / dtr;
                }
            }
            if (!last) {
                h[HX_Q] = 1.0 / dtr;
                cgptr v = Wv + L.o_tilt + 3 * k;
                double h00, h01, h11, b2;
                soc_w2(v[0], v[1] * v[1] + v[2] * v[2], Wbeta[L.c_tilt + k], h00, h01, h11, b2);
                inv_iso_rank1_2(dtr + b2, h11, v[1], v[2], h + HX_Q34);
            }
            if (!first && !last) {
                cgptr v = Wv + L.o_rate + 4 * k;
                double h00, h01, h11, b2;
                soc_w2(v[0], v[1] * v[1] + v[2] * v[2] + v[3] * v[3], Wbeta[L.c_rate + k], h00, h01, h11, b2);
                inv_iso_rank1_3(dtr + b2, h11, v[1], v[2], v[3], h + HX_W);
            }
#if SCVX_NODE_QR
            // u block (thrust part): M = dtr I + [Tmax cone: b2 I + h11 v1 v1'] + [gimbal cone: Jc' W^-2 Jc, Jc = [e1' / cos(delta); I]]
            //                         + (1 / wl^2) uhat uhat'   -- never assembled: M = A'A, see qr_inv3
            {
                cgptr v = Wv + L.o_tb + 4 * k;
                double h00, h01, h11, b2;
                soc_w2(v[0], v[1] * v[1] + v[2] * v[2] + v[3] * v[3], Wbeta[L.c_tb + k], h00, h01, h11, b2);
                const double sq = sqrt(dtr + b2), sh = sqrt(h11);
                const double wl = Wbeta[L.c_lb + k], iwl = 1.0 / wl;
                double A[9][3] = {{sq, 0, 0}, {0, sq, 0}, {0, 0, sq},
                                  {sh * v[1], sh * v[2], sh * v[3]},
                                  {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0},
                                  {iwl * uhat[3 * k], iwl * uhat[3 * k + 1], iwl * uhat[3 * k + 2]}};
                {
                    cgptr vc = Wv + L.o_tc + 4 * k;
                    const double vv[4] = {vc[0], vc[1], vc[2], vc[3]};
                    const double ib = 1.0 / Wbeta[L.c_tc + k];
                    for (int c = 0; c < 3; c++) {
                        const double x[4] = {c == 0 ? C.icos : 0.0, c == 0 ? 1.0 : 0.0, c == 1 ? 1.0 : 0.0, c == 2 ? 1.0 : 0.0};
                        double y[4];
                        soc_Winv_col<4>(vv, ib, x, y);
                        for (int r = 0; r < 4; r++) A[4 + r][c] = y[r];
                    }
                }
                gptr hi = hu + HU_SZ * k;
                if (last) {   // only u1 is free at the last node (u[2:3, K+1] = 0): its 1x1 block = |column 0 of A|^2
                    double m00 = 0;
                    for (int r = 0; r < 9; r++) m00 += A[r][0] * A[r][0];
                    for (int i = 0; i < 9; i++) hi[i] = 0.0;
                    hi[0] = 1.0 / m00;
                } else qr_inv3<9>(A, hi);
#else
            // u block
            {
                double M[9];
                for (int i = 0; i < 9; i++) M[i] = 0.0;
                M[0] = M[4] = M[8] = dtr;
                cgptr v = Wv + L.o_tb + 4 * k;
                double h00, h01, h11, b2;
                soc_w2(v[0], v[1] * v[1] + v[2] * v[2] + v[3] * v[3], Wbeta[L.c_tb + k], h00, h01, h11, b2);
                for (int a = 0; a < 3; a++)
                    for (int b = 0; b < 3; b++) M[3 * a + b] += h11 * v[1 + a] * v[1 + b] + (a == b ? b2 : 0.0);
                // Jc' W^-2 Jc with Jc = [icos e1'; I] (the cone's head is u_1 / cos(deltaMax)).  Written through
                // W^-2 = b2 (2 wt wt' - J), wt = (2 v0^2 - 1; -2 v0 v1):   b2 (I - icos^2 e1 e1' + 2 p p'),  p = Jc' wt.
                // The h00 / h01 / h11 form adds three terms of size v0^4 b2 that cancel to p_1^2 when the gimbal cone is
                // active (v1 along +e1): at mu ~ 1e-9 the block then stops being positive definite in double precision.
                v = Wv + L.o_tc + 4 * k;
                {
                    const double wb = Wbeta[L.c_tc + k], bt2 = 1.0 / (wb * wb);
                    const double w0 = 2.0 * v[0] * v[0] - 1.0, tv = -2.0 * v[0];
                    const double pc[3] = {C.icos * w0 + tv * v[1], tv * v[2], tv * v[3]};
                    for (int a = 0; a < 3; a++)
                        for (int b = 0; b < 3; b++) M[3 * a + b] += bt2 * (2.0 * pc[a] * pc[b] + (a == b ? 1.0 : 0.0));
                    M[0] -= bt2 * C.icos * C.icos;
                }
                const double wl = Wbeta[L.c_lb + k];
                const double il2 = 1.0 / (wl * wl);
                for (int a = 0; a < 3; a++)
                    for (int b = 0; b < 3; b++) M[3 * a + b] += il2 * uhat[3 * k + a] * uhat[3 * k + b];
                gptr hi = hu + HU_SZ * k;
                if (last) {
                    for (int i = 0; i < 9; i++) hi[i] = 0.0;
                    hi[0] = 1.0 / M[0];
                } else inv3(M, hi);
#endif
#if defined(SCVX_IPM_DEBUG) && !defined(__HIPCC__)
                if (k == SCVX_DBG_NODE) {
                    cgptr vb = Wv + L.o_tb + 4 * k; cgptr vc = Wv + L.o_tc + 4 * k;
                    SCVX_DBG("UBLK dtr %.17g tb %.17g %.17g %.17g %.17g %.17g tc %.17g %.17g %.17g %.17g %.17g lb %.17g uhat %.17g %.17g %.17g icos %.17g HI %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n",
                             dtr, (double)vb[0], (double)vb[1], (double)vb[2], (double)vb[3], (double)Wbeta[L.c_tb + k], (double)vc[0], (double)vc[1], (double)vc[2], (double)vc[3],
                             (double)Wbeta[L.c_tc + k], (double)wl, (double)uhat[3 * k], (double)uhat[3 * k + 1], (double)uhat[3 * k + 2], C.icos,
                             (double)hi[0], (double)hi[1], (double)hi[2], (double)hi[3], (double)hi[4], (double)hi[5], (double)hi[6], (double)hi[7], (double)hi[8]);
                }
#endif
                if constexpr (NU == 5) {
                    // fin block: trust region + the cone (finmxf; u4, u5); free at every node (u[2:3, K+1] = 0 fixes thrust only)
                    cgptr vf = Wv + L.o_fin + 3 * k;
                    soc_w2(vf[0], vf[1] * vf[1] + vf[2] * vf[2], Wbeta[L.c_fin + k], h00, h01, h11, b2);
                    inv_iso_rank1_2(dtr + b2, h11, vf[1], vf[2], hi + 9);
                }
            }
        }
        ex.sync();
        SCVX_T1(4);
        SCVX_TS(tC_);
        // ---- fused Schur-complement assembly + block Cholesky, sequential in k, every tile in scratch (LDS) ----
        //   Sd[k] = TA_k A_k' + TBm_k Bm_k' + TBp_k Bp_k' + Hxi_{k+1} + hnui I      TA_k  = A_k  Hxi_k   (14x14)
        //   So[k] = -TA_{k+1} + TBm_{k+1} Bp_k'   (block (k+1,k))                    TBm_k = Bm_k Hui_k   (14x3)
        //   pivot M_k = Sd[k] - Wb_{k-1} Wb_{k-1}',  Wb_k = So[k] L_k^-T             TBp_k = Bp_k Hui_{k+1}
        // The D_{k+1} tile is fetched (coalesced, into registers on the device) while segment k is processed.
        // members hoisted into locals: the Solver object sits in scratch memory on the device and would be
        // re-read after every barrier
        const dcptr D_ = D;
        const cgptr hx_ = hx;
        const cgptr hu_ = hu;
        const fptr Linv_ = Linv;
        const fptr Nf_ = Nf;
        const double hnui_ = hnui;
        double* sc = ex.scratch();
        double* M = sc + 32;            // 196  pivot tile / So / scratch product
        double* Wp = M + 196;           // 196  Wb[k-1]
        double* Li = Wp + 196;          // 196  Linv[k]
        double* Dt = Li + 196;          // 294  D_k tile (column-major 14x21: element (i,j) at 14 j + i)
        double* T = Dt + DSZ;           // 308  [TA | TBm | TBp] = [A_k Hxi_k | Bm_k Hui_k | Bp_k Hui_{k+1}], 14 x 20, row stride 22
        double* Bp = T + 14 * TS;           // 42   copy of Bp_k (column-major 14x3) kept across the D tile swap
        double* Hh = Bp + BPN;           // 84   hx_k (33) hu_k (9) | hx_{k+1} (33) hu_{k+1} (9)
        double* Hd = Hh + 2 * NODE_SZ;           // 196  dense Hxi of the node being multiplied
        // FUSED BORDER (sequential loop only).  The three border systems and the predictor's banded system need, per segment,
        //     r_k = (E Hb^-1 g)_k = [TA | TBm | TBp]_k [g_x,k; g_u,k; g_u,k+1] - Hxi_{k+1} g_x,k+1 + hnui g_nu,k   (g = Ptr, gx)
        // -- products with tiles this loop has in LDS anyway -- and their forward substitution t_k = L_k^-1 r_k + N_k t_{k-1} needs
        // L_k^-1 and N_k at the moment they are formed.  So the loop carries four right-hand sides along (columns: 0 Sg, 1 Ptr, 2 Pnu,
        // 3 predictor) and leaves L^-1 r in ys / ytr / ynu / dy; what used to follow it -- a pass materialising Ptr, two Hb^-1 passes, a
        // pass over D (E_apply2), a pass for the two plain columns, and the forward half of S_solveN (L^-1 and N streamed once
        // more) -- is gone: per interior-point iteration ~0.45 of 3.5 MB and seven of ~50 dependent passes.
        double* Gn = Hd + 196;                   // 2 x NXU x 4   node slices of the right-hand sides: [slot][row: x (14) | u (NU)][column]
        double* Rk = Gn + 2 * NXU * 4;           // 14 x 4        r_k
        double* Tt = Rk + 56;                    // 2 x 14 x 4    t_{k-1}, t_k (alternating)
        double* Sg = Tt + 112;                   // 84            segment scalars: gx_nu,k | ry_k | Pnu_k | res: nu_k | dk_k | rx_nu,k (14 each)
        double* Sk = Sg + (SCVX_FUSED_RES ? 84 : 42);   // 14     Sg_k, the sigma column of D_k (kept across the tile swap); the res tiles behind it exist only with SCVX_FUSED_RES
        double* Vn = Sk + 14;                    // 2 x NXU       res: node slices of V, slots k | k+1
        double* Pn = Vn + 2 * NXU;               // 2 x NXU       res: node slices of the D-independent part of rx
        double* Yv = Pn + 2 * NXU;               // 2 x 14        res: y_k | y_{k+1}
        // (the two-ended factorisation of the four-wavefront blocks kept a separate, back-substituted border until round 6: SCVX_TWISTED_TSPACE = 0)
        const bool kFusedBorder = SCVX_TWISTED_TSPACE != 0 || !(Ex::kTwisted && twisted());
        bool ok = true;
        if constexpr (Ex::kPipelineFactor) {
            if constexpr (Ex::kTwisted) {
                ok = twisted() ? factor_twisted(with_pred) : factor_pipelined(with_pred);
            } else
            ok = factor_pipelined(with_pred);   // two wavefronts: Schur-block assembly one segment ahead of the Cholesky chain
        } else {
        // All 14x14xK products below go through ex.tile_gemm: FP64 MFMA (v_mfma_f64_16x16x4) on the device — one A and
        // one B element per lane per instruction instead of 2 LDS reads per multiply-add — plain loops on the host.
        //
        // RESIDUALS IN THE LOOP (round 5, `res`).  An iteration used to read the linearisation D once for E V (equality residual), once
        // (plus the A_k' copies) for E'y (dual residual) and once more here.  With `res` the loop, which stages every D_k tile in LDS
        // anyway, forms both products itself:
        //     ry_k = (E V + dk)_k                                   while D_k is staged,
        //     (E'y) at node k+1 = [A | B-]_{k+1}' y_{k+1} + B+_k' y_k - y_k      right after D_{k+1} has been swapped in,
        // completes  rx = (c - J'Z) + E'y  and  gx = q - E'y  node by node from the D-independent parts the cone passes left in rx / gx
        // (rx = c - J'Z, gx = -(c - J'Z) - J' W^-1 (lam - W^-1 rz)), writes rx, gx, ry back and accumulates |rx|^2, |ry|^2 and Sg . y.
        // Because g of node k+1 is only known after the swap, r_k is formed in two parts -- [TA | TBm]_k g_k early, TBp_k g_u,k+1 -
        // Hxi_{k+1} g_x,k+1 late -- and the forward substitution t_k = L_k^-1 r_k + N_k t_{k-1} closes the step (its N_k t_{k-1} part
        // waits in the accumulator registers while the swap overwrites the tile N_k sits in).
        // prologue: D_0, hx_0/hu_0 -> LDS; TA_0, TBm_0
        for (int e = ex.lane(); e < DSZ; e += ex.nlanes()) Dt[e] = D_[e];
        for (int e = ex.lane(); e < NODE_SZ; e += ex.nlanes()) Hh[NODE_SZ + e] = e < HX_SZ ? hx_[e] : hu_[e - HX_SZ];
        ex.sync_lds();
        for (int e = ex.lane(); e < 196; e += ex.nlanes()) Hd[e] = hxi_entry(Hh + NODE_SZ, e / 14, e % 14);
        ex.sync_lds();
        ex.tile_gemm(T, TS, 1, Dt, 1, 14, Hd, 14, 1, 14, 1.0, false);          // TA_0 = A_0 Hxi_0
        for (int q = ex.lane(); q < BPN; q += ex.nlanes()) {
            const int i = q / NU, c = q - NU * i;
            T[TS * i + 14 + c] = bhu(Dt, 14, i, c, Hh + NODE_SZ + HX_SZ);
        }
        // node slices of the border right-hand sides: element e = NXU vec + row of node nd (vec 0: Ptr = v1 of the trust-region
        // cone, column 1; vec 1: the predictor's gx, column 3); segment scalars: element e of segment sgk
        const cgptr Wtr_ = Wv + L.o_tr + 1; const cgptr Wnu_ = Wv + L.o_nu + 1; const gptr gx_ = gx; const gptr ry_ = ry;
        const gptr rx_ = rx; const cgptr V_ = V; const cgptr y_ = y; const cgptr dk_ = dk;
        const int nx_ = L.nx, nxu_ = L.nx + L.nu_;
        auto gnode_elem = [&](int nd, int e) -> double {
            const int vec = e >= NXU ? 1 : 0, row = e - NXU * vec;
            if (vec == 1 && !with_pred) return 0.0;
            cgptr src = vec ? (cgptr)gx_ : Wtr_;
            return row < 14 ? src[14 * nd + row] : src[nx_ + NU * nd + (row - 14)];
        };
        // res: element e = NXU vec + row of node nd (vec 0: V, vec 1: the D-independent part of rx)
        auto vnode_elem = [&](int nd, int e) -> double {
            const int vec = e >= NXU ? 1 : 0, row = e - NXU * vec;
            cgptr src = vec ? (cgptr)rx_ : V_;
            return row < 14 ? src[14 * nd + row] : src[nx_ + NU * nd + (row - 14)];
        };
        // segment scalars: [0,14) gx_nu,k | [14,28) ry_k (read only without res) | [28,42) Pnu_k | res: [42,56) nu_k | [56,70) dk_k | [70,84) rx_nu,k
        auto gseg_elem = [&](int sgk, int e) -> double {
            if (e >= 70) return rx_[nxu_ + 14 * sgk + (e - 70)];
            if (e >= 56) return dk_[14 * sgk + (e - 56)];
            if (e >= 42) return V_[nxu_ + 14 * sgk + (e - 42)];
            if (e >= 28) return Wnu_[14 * sgk + (e - 28)];
            if (!with_pred) return 0.0;
            return e < 14 ? gx_[nxu_ + 14 * sgk + e] : (res ? 0.0 : (double)ry_[14 * sgk + (e - 14)]);
        };
        for (int e = ex.lane(); e < 2 * NXU * 4 + 56 + 112; e += ex.nlanes()) Gn[e] = 0.0;   // Gn, Rk, Tt (columns 0 and 2 of Gn stay zero)
        ex.sync_lds();
        for (int e = ex.lane(); e < 2 * NXU; e += ex.nlanes()) {
            const int vec = e >= NXU ? 1 : 0, row = e - NXU * vec;
            Gn[NXU * 4 + 4 * row + (vec ? 3 : 1)] = gnode_elem(0, e);
            if (res) (vec ? Pn : Vn)[NXU + row] = vnode_elem(0, e);
        }
        double nrx2 = 0.0, nry2 = 0.0, sgacc = 0.0;   // lane-local partial sums of |rx|^2, |ry|^2, Sg . y
        const double s_var = res ? (double)V_[L.iS] : 0.0;
        if (res) {
            for (int e = ex.lane(); e < 14; e += ex.nlanes()) Yv[14 + e] = y_[e];
            ex.sync_lds();
            // node 0: (E'y)_x,0 = A_0' y_0, (E'y)_u,0 = B-_0' y_0; fixed rows (rocketland.jl:109-113) are zero in rx and gx
            for (int row = ex.lane(); row < NXU; row += ex.nlanes()) {
                double ev = 0.0;
                for (int i = 0; i < 14; i++) ev += Dt[14 * row + i] * Yv[14 + i];
                const bool fx = row < 14 && fixed_x(0, row);
                const double rxv = fx ? 0.0 : Pn[NXU + row] + ev;
                const double gxv = fx ? 0.0 : Gn[NXU * 4 + 4 * row + 3] - ev;
                Gn[NXU * 4 + 4 * row + 3] = gxv;
                const int at = row < 14 ? row : nx_ + (row - 14);
                rx_[at] = rxv; gx_[at] = gxv;
                nrx2 += rxv * rxv;
            }
        }
        ex.sync_lds();
        // the compact node inverses (42 doubles) are requested one segment ahead as well: lane e holds element e of node
        // k + 1 while segment k - 1 is processed (register-prefetching executors have at least 42 lanes)
        auto node_elem = [&](int node, int e) -> double {
            return e < HX_SZ ? hx_[(size_t)node * HX_SZ + e] : hu_[HU_SZ * node + (e - HX_SZ)];
        };
        double hn = 0.0, gn = 0.0, sgn = 0.0, sgnb = 0.0, vpn = 0.0, yn = 0.0;
        if (Ex::kPrefetchRegs > 0 && ex.lane() < NODE_SZ) hn = node_elem(1, ex.lane());
        if (Ex::kPrefetchRegs > 0 && ex.lane() < 2 * NXU) gn = gnode_elem(1, ex.lane());
        if (Ex::kPrefetchRegs > 0 && ex.lane() < 42) sgn = gseg_elem(0, ex.lane());
        if (Ex::kPrefetchRegs > 0 && res) {
            if (ex.lane() < 42) sgnb = gseg_elem(0, 42 + ex.lane());
            if (ex.lane() < 2 * NXU) vpn = vnode_elem(1, ex.lane());
            if (ex.lane() < 14) yn = y_[(K > 1 ? 14 : 0) + ex.lane()];
        }
        const gptr xq_[4] = {ys, ytr, ynu, dy};
        const int hpos_lane = hx_dense_pos(ex.lane() <= HX_SZ ? ex.lane() : 0);
        typename Ex::Acc cg;   // Gram matrix of the forward-substituted right-hand sides (rows / columns 0..3), summed over the segments
        ex.acc_zero(cg);
        double gnacc = 0.0;    // <Pnu, gx_nu>, lane-local
        for (int k = 0; k < K; k++) {
            SCVX_TS(ta_);
            // prefetch the next segment's tile
            constexpr int NPRE = Ex::kPrefetchRegs > 0 ? (DSZ + Ex::kLanes - 1) / Ex::kLanes : 0;
            double hn2 = 0.0, gn2 = 0.0, sgn2 = 0.0, sgnb2 = 0.0, vpn2 = 0.0, yn2 = 0.0;
            if (NPRE > 0 && ex.lane() < NODE_SZ) hn2 = node_elem(k + 2 <= K ? k + 2 : K, ex.lane());
            if (NPRE > 0 && ex.lane() < 2 * NXU) gn2 = gnode_elem(k + 2 <= K ? k + 2 : K, ex.lane());
            if (NPRE > 0 && ex.lane() < 42) sgn2 = gseg_elem(k + 1 < K ? k + 1 : k, ex.lane());
            if (NPRE > 0 && res) {
                if (ex.lane() < 42) sgnb2 = gseg_elem(k + 1 < K ? k + 1 : k, 42 + ex.lane());
                if (ex.lane() < 2 * NXU) vpn2 = vnode_elem(k + 2 <= K ? k + 2 : K, ex.lane());
                if (ex.lane() < 14) yn2 = y_[14 * (k + 2 < K ? k + 2 : K - 1) + ex.lane()];
            }
            double pre[NPRE > 0 ? NPRE : 1];
            dcptr Dn = D_ + (size_t)(k + 1 < K ? k + 1 : k) * DSZ;
            if (NPRE > 0) {
                SCVX_UNROLL
                for (int q = 0; q < NPRE; q++) { const int e = ex.lane() + ex.nlanes() * q; pre[q] = e < DSZ ? Dn[e] : 0.0; }
            }
            // node inverses: slot 0 <- slot 1 (k), slot 1 <- k+1, dense tile of k+1
            for (int e = ex.lane(); e < NODE_SZ; e += ex.nlanes()) {
                const double nk1 = NPRE > 0 ? hn : node_elem(k + 1, e);
                Hh[e] = Hh[NODE_SZ + e];
                Hh[NODE_SZ + e] = nk1;
            }
            hn = hn2;
            // ... and the right-hand sides' slices: slot 0 <- slot 1 (node k), slot 1 <- node k + 1; the segment's scalars
            for (int e = ex.lane(); e < 2 * NXU; e += ex.nlanes()) {
                const int vec = e >= NXU ? 1 : 0, row = e - NXU * vec, at = 4 * row + (vec ? 3 : 1);
                const double v1 = NPRE > 0 ? gn : gnode_elem(k + 1, e);
                Gn[at] = Gn[NXU * 4 + at];
                Gn[NXU * 4 + at] = v1;
                if (res) {
                    double* dst = vec ? Pn : Vn;
                    const double w1 = NPRE > 0 ? vpn : vnode_elem(k + 1, e);
                    dst[row] = dst[NXU + row];
                    dst[NXU + row] = w1;
                }
            }
            for (int e = ex.lane(); e < 42; e += ex.nlanes()) {
                Sg[e] = NPRE > 0 ? sgn : gseg_elem(k, e);
                if (res) Sg[42 + e] = NPRE > 0 ? sgnb : gseg_elem(k, 42 + e);
            }
            if (res)
                for (int e = ex.lane(); e < 14; e += ex.nlanes()) {
                    Yv[e] = Yv[14 + e];
                    Yv[14 + e] = NPRE > 0 ? yn : (double)y_[14 * (k + 1 < K ? k + 1 : K - 1) + e];
                }
            for (int e = ex.lane(); e < 14; e += ex.nlanes()) Sk[e] = Dt[14 * CS + e];   // Sg_k: the tile is swapped before r_k is closed
            gn = gn2; sgn = sgn2; sgnb = sgnb2; vpn = vpn2; yn = yn2;
            ex.sync_lds();
            SCVX_TE(ta_, 2);
            SCVX_TS(ta2_);
            // dense Hxi_{k+1}: its 34 structural non-zeros (the rest of the tile stays zero); the pivot tile starts from it (acc_store_init)
            for (int e = ex.lane(); e <= HX_SZ; e += ex.nlanes())
                Hd[Ex::kLanes >= 64 ? hpos_lane : hx_dense_pos(e)] = Hh[NODE_SZ + (e < HX_SZ ? e : HX_Q)];
            for (int q = ex.lane(); q < BPN; q += ex.nlanes()) {   // TBp_k = Bp_k Hui_{k+1}
                const int i = q / NU, c = q - NU * i;
                T[TS * i + 14 + NU + c] = bhu(Dt, 14 + NU, i, c, Hh + NODE_SZ + HX_SZ);
            }
            if (res) {
                // the nu rows of segment k ((E'y)_nu,k = y_k), the equality residual ry_k, and Sg_k . y_k
                for (int i = ex.lane(); i < 14; i += ex.nlanes()) {
                    const double yk = Yv[i];
                    const double rxv = Sg[70 + i] + yk, gxv = Sg[i] - yk;
                    Sg[i] = gxv;
                    rx_[nxu_ + 14 * k + i] = rxv; gx_[nxu_ + 14 * k + i] = gxv;
                    nrx2 += rxv * rxv;
                    double a = 0.0;
                    for (int j = 0; j < 14; j++) a += Dt[14 * j + i] * Vn[j];
                    for (int j = 0; j < NU; j++) a += Dt[14 * (14 + j) + i] * Vn[14 + j];
                    for (int j = 0; j < NU; j++) a += Dt[14 * (14 + NU + j) + i] * Vn[NXU + 14 + j];
                    a += Dt[14 * CS + i] * s_var;
                    a += Sg[42 + i] - Vn[NXU + i];
                    a += Sg[56 + i];
                    Sg[14 + i] = a;
                    ry_[14 * k + i] = a;
                    nry2 += a * a;
                    sgacc += Dt[14 * CS + i] * yk;
                }
            }
            ex.sync_lds();
            SCVX_TE(ta2_, 24);
            SCVX_TS(tb_);
            // pivot tile: M += [TA|TBm|TBp] [A|Bm|Bp]'  -  Wb_{k-1} Wb_{k-1}'
            {   // two independent accumulators, no LDS round trip between the terms of either sum
                typename Ex::Acc cm, cr;
                ex.acc_zero(cm); ex.acc_zero(cr);
                ex.acc_mac(cm, T, TS, 1, Dt, 14, 1, TW, 1.0);
                // r_k, early part (columns 1 and 3): [TA | TBm]_k [g_x,k; g_u,k]
                ex.acc_mac(cr, T, TS, 1, Gn, 4, 1, 14 + NU, 1.0, 4);
                if (k > 0) ex.acc_mac(cm, Wp, 14, 1, Wp, 1, 14, 14, -1.0);
                ex.acc_store_init(cm, M, Hd, hnui_);   // M = Hxi_{k+1} + hnui I + the products
                ex.acc_store(cr, Rk, 4, 1, false, 4);
            }
            ex.sync_lds();
            SCVX_TE(tb_, 5);
            SCVX_TS(tc_);
            ok = ex.chol_inv14(M, Li) && ok;   // L^-1 of the pivot tile -> Li (row-major, lower, zeros above)
            ex.sync_lds();
            SCVX_TE(tc_, 12);
            SCVX_TS(td_);
            for (int e = ex.lane(); e < LINV_SZ; e += ex.nlanes()) {   // packed lower triangle (see linv_row)
                const int p = e / 15, q = e - 15 * p;
                const int i = q <= p ? p : 13 - p, j = q <= p ? q : q - (p + 1);
                Linv_[(size_t)k * LINV_SZ + e] = Li[14 * i + j];
            }
            SCVX_TE(td_, 13);
            SCVX_TS(td2_);
            typename Ex::Acc ct;   // t_k = L_k^-1 r_k + N_k t_{k-1}: the second term now, the first when r_k is complete
            ex.acc_zero(ct);
            double* Tc = Tt + 56 * (k & 1); const double* Tp = Tt + 56 * ((k + 1) & 1);
            if (k > 0) {  // Nf[k] = -Linv_k Wb_{k-1}, stored transposed (the layout the executor's chain consumes)
                ex.tile_gemm(M, 1, 14, Li, 14, 1, Wp, 14, 1, 14, -1.0, false);
                ex.sync_lds();
                for (int e = ex.lane(); e < 196; e += ex.nlanes()) Nf_[(size_t)k * 196 + e] = M[e];
                ex.acc_mac(ct, M, 1, 14, Tp, 4, 1, 14, 1.0, 4);
                ex.sync_lds();   // M is overwritten below
            }
            SCVX_TE(td2_, 25);
            SCVX_TS(te_);
            if (k + 1 < K) {
                // keep Bp_k, swap in D_{k+1}
                for (int q = ex.lane(); q < BPN; q += ex.nlanes()) Bp[q] = Dt[14 * (14 + NU) + q];
                ex.sync_lds();
                if (NPRE > 0) {
                    SCVX_UNROLL
                    for (int q = 0; q < NPRE; q++) { const int e = ex.lane() + ex.nlanes() * q; if (e < DSZ) Dt[e] = pre[q]; }
                } else {
                    for (int e = ex.lane(); e < DSZ; e += ex.nlanes()) Dt[e] = Dn[e];
                }
                ex.sync_lds();
                SCVX_TE(te_, 26);
                SCVX_TS(te2_);
                // TA_{k+1} = A_{k+1} Hxi_{k+1}, TBm_{k+1}
                ex.tile_gemm(T, TS, 1, Dt, 1, 14, Hd, 14, 1, 14, 1.0, false);
                for (int q = ex.lane(); q < BPN; q += ex.nlanes()) {
                    const int i = q / NU, c = q - NU * i;
                    T[TS * i + 14 + c] = bhu(Dt, 14, i, c, Hh + NODE_SZ + HX_SZ);
                }
                ex.sync_lds();
                SCVX_TE(te2_, 27);
                SCVX_TS(te3_);
                // So[k] = -TA_{k+1} + TBm_{k+1} Bp_k'
                for (int e = ex.lane(); e < 196; e += ex.nlanes()) {
                    const int i = e / 14, j = e - 14 * i;
                    M[e] = so_elem(T, Bp, i, j);
                }
                ex.sync_lds();
                SCVX_TE(te3_, 28);
                SCVX_TS(te4_);
                ex.tile_gemm(Wp, 14, 1, M, 14, 1, Li, 1, 14, 14, 1.0, false);      // Wb_k = So Linv'
                SCVX_TE(te4_, 29);
            }
            SCVX_TS(te5_);
            if (res) {
                // node k + 1: (E'y)_x = A_{k+1}' y_{k+1} - y_k, (E'y)_u = B-_{k+1}' y_{k+1} + B+_k' y_k (at the last node only the y_k terms,
                // B+_{K-1} still staged in the tile); rx = (c - J'Z) + E'y, gx = q - E'y, fixed rows zero (rocketland.jl:109-115)
                const bool lastn = k + 1 == K;
                for (int row = ex.lane(); row < NXU; row += ex.nlanes()) {
                    double ev = 0.0;
                    if (!lastn) {
                        for (int i = 0; i < 14; i++) ev += Dt[14 * row + i] * Yv[14 + i];
                        if (row < 14) ev -= Yv[row];
                        else for (int i = 0; i < 14; i++) ev += Bp[14 * (row - 14) + i] * Yv[i];
                    } else {
                        if (row < 14) ev = -Yv[row];
                        else for (int i = 0; i < 14; i++) ev += Dt[14 * (NU + row) + i] * Yv[i];   // column 14 + NU + (row - 14)
                    }
                    const bool fx = lastn && (row < 14 ? fixed_x(K, row) : fixed_u(K, row - 14));
                    const double rxv = fx ? 0.0 : Pn[NXU + row] + ev;
                    const double gxv = fx ? 0.0 : Gn[NXU * 4 + 4 * row + 3] - ev;
                    Gn[NXU * 4 + 4 * row + 3] = gxv;
                    const int at = row < 14 ? 14 * (k + 1) + row : nx_ + NU * (k + 1) + (row - 14);
                    rx_[at] = rxv; gx_[at] = gxv;
                    nrx2 += rxv * rxv;
                }
            }
            ex.sync_lds();
            {   // r_k, late part: TBp_k g_u,k+1 - Hxi_{k+1} g_x,k+1; then the plain parts and the forward substitution
                typename Ex::Acc cr;
                ex.acc_zero(cr);
                ex.acc_mac(cr, T + 14 + NU, TS, 1, Gn + NXU * 4 + 14 * 4, 4, 1, NU, 1.0, 4);
                ex.acc_mac(cr, Hd, 14, 1, Gn + NXU * 4, 4, 1, 14, -1.0, 4);
                ex.acc_store(cr, Rk, 4, 1, true, 4);
            }
            ex.sync_lds();
            // the plain parts of r_k: column 0 = Sg_k (the sigma column of D_k), column 2 = hnui Pnu_k, column 3 += hnui gx_nu,k + ry_k
            for (int i = ex.lane(); i < 14; i += ex.nlanes()) {
                Rk[4 * i] = Sk[i];
                Rk[4 * i + 2] = hnui_ * Sg[28 + i];
                Rk[4 * i + 3] += hnui_ * Sg[i] + Sg[14 + i];
                gnacc += Sg[28 + i] * Sg[i];
            }
            ex.sync_lds();
            SCVX_TE(te5_, 30);
            SCVX_TS(te6_);
            ex.acc_mac(ct, Li, 14, 1, Rk, 4, 1, 14, 1.0, 4);
            ex.acc_store(ct, Tc, 4, 1, false, 4);
            ex.sync_lds();
            ex.acc_mac(cg, Tc, 1, 4, Tc, 4, 1, 14, 1.0, 4);   // rows 0..3: t_a . t_b of this segment (the other rows: never read)
            for (int e = ex.lane(); e < 56; e += ex.nlanes()) {
                const int q = e / 14, i = e - 14 * q;
                if (q < 3 || with_pred) xq_[q][14 * k + i] = Tc[4 * i + q];
            }
            SCVX_TE(te6_, 31);
        }
        if (res) { res_nrx2 = ex.sum(nrx2); res_nry2 = ex.sum(nry2); res_sgy = ex.sum(sgacc); }
        ex.sync_lds();
        ex.acc_store(cg, Rk, 4, 1, false, 4);
        ex.sync_lds();
        for (int q = 0; q < 16; q++) gram[q] = Rk[q];
        gn_pred = ex.sum(gnacc);
        }   // sequential factorisation
        ex.sync();   // the factors written above are read back (by other lanes) in the border solves
        SCVX_TE(tC_, 6);
        SCVX_TS(tB_);
        // border: three banded systems [Hb E'; E 0][l; y] = [g; r] sharing every matrix read
        //   s  : g = 0,            r = -Sg   ->  S ys  = +Sg
        //   tr : g = (Ptr, 0, 0),  r = 0     ->  S ytr = E_loc Hb^-1 Ptr =: rtr
        //   nu : g = (0, 0, Pnu),  r = 0     ->  S ynu = hnui Pnu        (E_loc is the identity on the nu block)
        // and, with_pred, a fourth one on the same sweeps: the y of the predictor's banded solve, S dy = E Hb^-1 gx + ry.
        // Only the multipliers y are kept.  The local parts l = Hb^-1 (g - E'y) of the border solutions are never
        // formed: every border coefficient is an inner product in y-space (below), and a solve applies its border
        // correction to dy and to the right-hand side BEFORE its single final  Hb^-1 (g - E'dy)  (kkt_solve).
        tsp = kFusedBorder;
        if (kFusedBorder) {
            // the forward substitutions AND their Gram matrix were carried by the loop: ys / ytr / ynu (and dy, with_pred) stay in t-space;
            // what is left is Hb^-1 Ptr on (dx, du) for the two inner products that use it
            Hb_inv(Wv + L.o_tr + 1, ptl, false);
            const double ptp = dot(Wv + L.o_tr + 1, ptl, L.nx + L.nu_);
            css = gram[0]; cst = -gram[1]; csn = -gram[2];
            cts = -gram[1]; ctt = -ptp + gram[5]; ctn = gram[6];
            cns = -gram[2]; cnt_ = gram[6]; pny = gram[10] / hnui_;
            SCVX_TE(tB_, 7);
            return ex.all(ok);
        } else {
            const int nxu = L.nx + L.nu_;
            const gptr g_tr = r1;     // nloc: Ptr on (dx,du), 0 on nu   (r1: refinement scratch, idle here)
            {
                cgptr wt = Wv + L.o_tr + 1;
                stream<8>(0, L.nloc, [&](int i) { return i < nxu ? wt[i] : 0.0; }, [&](int i, double v) { g_tr[i] = v; });
            }
            ex.sync();
            Hb_inv(g_tr, ptl);
            if (with_pred) {
                Hb_inv(gx, cw);
                E_apply2(ptl, cw, rtr, rp, ry);                    // rtr, r_pred = E Hb^-1 gx + ry
            } else {
                E_apply(ptl, rtr, false);                          // rtr
            }
            {
                gptr ty = tmpy; gptr r2_ = r2; cgptr wn = Wv + L.o_nu + 1;
                stream(0, L.ny, [&](int r) { const int k = r / 14, i = r - 14 * k; return D2{D_[(size_t)k * DSZ + 14 * CS + i], wn[r]}; },
                       [&](int r, const D2& v) { ty[r] = v.a; r2_[r] = hnui_ * v.b; });   // r_s = +Sg, r_nu
            }
            ex.sync();
            if (with_pred) {
                const cgptr rr[4] = {tmpy, rtr, r2, rp};
                const gptr xx[4] = {ys, ytr, ynu, dy};
                const gptr tt4[4] = {tchain, cy, tq0, tq1};
                S_solveN<4>(rr, xx, tt4);
            } else {
                const cgptr rr[3] = {tmpy, rtr, r2};
                const gptr xx[3] = {ys, ytr, ynu};
                const gptr tt3[3] = {tchain, cy, tq0};
                S_solveN<3>(rr, xx, tt3);
            }
        }
        // border coefficients, all in y-space (Sg = the sigma column of E, Pnu = v1 of the nu cone, rtr = E Hb^-1 Ptr):
        //   <Ptr, ls>  = -<rtr, ys>          <Ptr, ltr> = <Ptr, ptl> - <rtr, ytr>      <Ptr, lnu> = -<rtr, ynu>
        //   <Pnu, ls>  = -hnui <Pnu, ys>     <Pnu, ltr> = -hnui <Pnu, ytr>             <Pnu, ynu> = pny
        {
            double a[9];
            for (int q = 0; q < 9; q++) a[q] = 0.0;
            {
                cgptr y0 = ys; cgptr y1 = ytr; cgptr y2 = ynu; cgptr rt = rtr; cgptr wn = Wv + L.o_nu + 1;
                stream(0, L.ny, [&](int r) { const int k = r / 14, i = r - 14 * k; return D6{D_[(size_t)k * DSZ + 14 * CS + i], y0[r], y1[r], y2[r], rt[r], wn[r]}; },
                       [&](int, const D6& v) {
                           a[0] += v.a * v.b; a[1] += v.a * v.c; a[2] += v.a * v.d;
                           a[3] += v.e * v.b; a[4] += v.e * v.c; a[5] += v.e * v.d;
                           a[6] += v.f * v.b; a[7] += v.f * v.c; a[8] += v.f * v.d;
                       });
            }
            for (int q = 0; q < 9; q++) a[q] = ex.sum(a[q]);
            const double ptp = dot(Wv + L.o_tr + 1, ptl, L.nx + L.nu_);
            css = a[0]; cst = -a[1]; csn = -a[2];
            cts = -a[3]; ctt = -ptp + a[4]; ctn = a[5];
            cns = -hnui_ * a[6]; cnt_ = hnui_ * a[7]; pny = a[8];
        }
        SCVX_TE(tB_, 7);
        return ex.all(ok);
    }

    // full reduced KKT: [H E'; E 0][dwv; dyv] = [g; ryv]  (g var-shaped incl. 4 globals)
    // have_band: (dwv, dyv) already hold the banded solution for (g, rsign ryv) (build_kkt(with_pred))
    // eout / sgout (optional): E_loc' dyv on the (dx, du) rows and Sg . dyv, by-products of the final E' product (newton_corr)
    SCVX_HD void kkt_solve(cgptr g, cgptr ryv, gptr dwv, gptr dyv, double rsign = 1.0, bool have_band = false, gptr eout = nullptr,
                           double* sgout = nullptr) {
        // banded multiplier: S dy = E Hb^-1 g - rsign ryv   (have_band: build_kkt(true) left it in dyv -- forward-substituted only, tsp)
        // inner products of the banded LOCAL solution dl = Hb^-1 (g - E'dy) with Sg / Ptr / Pnu, without forming dl:
        //   <Ptr, dl> = <ptl, g> - <rtr, dy>,   <Pnu, dl_nu> = hnui (<Pnu, g_nu> - <Pnu, dy>)
        double c0s, c0t, c0n;
        if (tsp) {
            // t-space: dyv holds t_g = L^-1 (E Hb^-1 g - rsign ryv);  <Sg, dy> = <t_s, t_g>, <rtr, dy> = <t_tr, t_g>, hnui <Pnu, dy> = <t_nu, t_g>
            double a0, a1, a2, gn;
            if (have_band) { a0 = gram[3]; a1 = gram[7]; a2 = gram[11]; gn = gn_pred; }
            else {
                Hb_inv(g, tmpl);
                (void)E_apply(tmpl, tmpy, false, ryv, -rsign);
                S_fwd(tmpy, dyv);
                double q0 = 0, q1 = 0, q2 = 0, q3 = 0;
                cgptr b0 = ys; cgptr b1 = ytr; cgptr b2 = ynu; cgptr wn = Wv + L.o_nu + 1; cgptr gnu = g + L.nx + L.nu_;
                stream(0, L.ny, [&](int i) { return D6{dyv[i], b0[i], b1[i], b2[i], wn[i], gnu[i]}; },
                       [&](int, const D6& v) { q0 += v.a * v.b; q1 += v.a * v.c; q2 += v.a * v.d; q3 += v.e * v.f; });
                a0 = ex.sum(q0); a1 = ex.sum(q1); a2 = ex.sum(q2); gn = ex.sum(q3);
            }
            c0s = a0;
            c0t = dot(ptl, g, L.nx + L.nu_) - a1;
            c0n = hnui * gn - a2;
        } else {
        if (!have_band) {
            Hb_inv(g, tmpl);
            (void)E_apply(tmpl, tmpy, false, ryv, -rsign);
            S_solve(tmpy, dyv);
        }
        double a = 0, bt = 0, bn = 0, gn = 0;
        {
            const dcptr D_ = D; cgptr rt = rtr; cgptr wn = Wv + L.o_nu + 1; cgptr gnu = g + L.nx + L.nu_;
            stream(0, L.ny, [&](int r) { const int k = r / 14, i = r - 14 * k; return D5{D_[(size_t)k * DSZ + 14 * CS + i], dyv[r], rt[r], wn[r], gnu[r]}; },
                   [&](int, const D5& v) { a += v.a * v.b; bt += v.c * v.b; bn += v.d * v.b; gn += v.d * v.e; });
        }
        c0s = ex.sum(a);
        bt = ex.sum(bt); bn = ex.sum(bn); gn = ex.sum(gn);
        c0t = dot(ptl, g, L.nx + L.nu_) - bt;
        c0n = hnui * (gn - bn);
        }
        // ---- border: three unknowns (s, ctr, cnu), the heads of the cones eliminated in closed form ----
        // With W^-2 = [[h00, h01 v1'], [h01 v1, b2 I + h11 v1 v1']] a cone whose head t appears in no other row gives
        //     h00 t + h01 pi = g_t,   c = h01 t + h11 pi = kappa + sigma pi,   pi = <v1, l>,
        //     kappa = h01 g_t / h00,  sigma = h11 - h01^2 / h00 = -8 v0^2 b2^2 / h00          (v0^2 - |v1|^2 = 1)
        // -- the stiff rank-one term h11 and the head coupling cancel to a SOFT direction b2 / (8 v0^2 |v1|^2 + 1).
        // Left to a 6x6 elimination on (t, a = h11 pi) that cancellation happens numerically between entries of size
        // v0^4 b2: its 2x2 blocks have condition (v0 + |v1|)^8, beyond double precision once mu < 1e-7, and the solver
        // stalled at merit ~1e-7 on 40 % of the subproblems.  In closed form nothing cancels:
        //     nu cone (Hb = b2 I on its body):  -<Pnu, lnu> - 1 / sigma = hnui (pny + 1 / (8 v0^2)),  pny = <Pnu, ynu>
        //     (ts; s):   Msg3 - Msg1 Msg2 / Msg0 = b2 / (1 + 8 v0^2 vs^2)
        //     tr cone (its head also carries the radius row hrk):  sigma = 8 v0^2 b2 (hrk - b2) / (h00 + hrk).
        const double v0n = q_nu[0], n1n = q_nu[1], h00n = h_nu[0], h01n = h_nu[1];
        const double v0t = q_tr[0], b2t = h_tr[3], h01t = h_tr[1];
        const double dtt = h_tr[0] + hrk;
        const double sig_t = 8.0 * v0t * v0t * b2t * (hrk - b2t) / dtt, kap_t = h01t * g[L.iTTR] / dtt;
        const double sig_n = -8.0 * v0n * v0n / (hnui * hnui * h00n), kap_n = h01n * g[L.iTNU] / h00n;
        const double v0s = q_sg[0], vs = q_sg[1];
        double M[9], r[3];
        M[0] = q_sg[2] / (1.0 + 8.0 * v0s * v0s * vs * vs) + css; M[1] = cst; M[2] = csn;
        r[0] = g[L.iS] - c0s - q_sg[3] * g[L.iTS];
        M[3] = -sig_t * cts; M[4] = 1.0 - sig_t * ctt; M[5] = -sig_t * ctn; r[1] = kap_t + sig_t * c0t;
        M[6] = cns; M[7] = cnt_; M[8] = hnui * (pny + 1.0 / (8.0 * v0n * v0n));
        r[2] = -c0n - (v0n * v0n + n1n) * g[L.iTNU] * hnui / (2.0 * v0n);
        // Gaussian elimination with partial pivoting (every lane redundantly)
        for (int c = 0; c < 3; c++) {
            int p = c; double best = fabs(M[c * 3 + c]);
            for (int i = c + 1; i < 3; i++) if (fabs(M[i * 3 + c]) > best) { best = fabs(M[i * 3 + c]); p = i; }
            if (p != c) { for (int j = 0; j < 3; j++) { const double t = M[c * 3 + j]; M[c * 3 + j] = M[p * 3 + j]; M[p * 3 + j] = t; } const double t = r[c]; r[c] = r[p]; r[p] = t; }
            const double ip = 1.0 / M[c * 3 + c];
            for (int i = c + 1; i < 3; i++) {
                const double f = M[i * 3 + c] * ip;
                for (int j = c; j < 3; j++) M[i * 3 + j] -= f * M[c * 3 + j];
                r[i] -= f * r[c];
            }
        }
        double b[3];
        for (int i = 2; i >= 0; i--) { double a_ = r[i]; for (int j = i + 1; j < 3; j++) a_ -= M[i * 3 + j] * b[j]; b[i] = a_ / M[i * 3 + i]; }
        const double s_ = b[0], ctr = b[1], cnu = b[2];
        // heads: pi_t from its definition (no division by sigma, which vanishes at hrk = b2), pi_n = (cnu - kappa) / sigma
        const double pi_t = c0t + cts * s_ + ctt * ctr + ctn * cnu;
        const double pi_n = (cnu - kap_n) / sig_n;
        const double ttr_ = (g[L.iTTR] - h01t * pi_t) / dtt;
        const double tnu_ = (g[L.iTNU] - h01n * pi_n) / h00n;
        const double ts_ = g[L.iTS] / h00s - q_sg[3] * s_;
        ex.sync();
        {   // the border correction (in y-space, or -- tsp -- in t-space: the same combination of the forward-substituted vectors) ...
            cgptr b0 = ys; cgptr b1 = ytr; cgptr b2 = ynu;
            stream(0, L.ny, [&](int i) { return D4{dyv[i], b0[i], b1[i], b2[i]}; },
                   [&](int i, const D4& v) { dyv[i] = v.a + v.b * s_ - v.c * ctr - v.d * cnu; });
        }
        ex.sync();
        if (tsp) {   // ... followed by the one back substitution of the solve
            const gptr xx[1] = {dyv};
            const gptr tt[1] = {tchain};
            S_backN<1>(xx, tt);
        }
        // ... and the local step in one go:  dw = Hb^-1 (g - Ptr ctr - Pnu cnu - E'dy)
        {
            const double sg_ = Et_apply(dyv, tmpl2, g, 1, ctr, cnu, eout);
            if (sgout) *sgout = sg_;
        }
        Hb_inv(tmpl2, dwv);
        if (ex.lane() == 0) { dwv[L.iS] = s_; dwv[L.iTS] = ts_; dwv[L.iTNU] = tnu_; dwv[L.iTTR] = ttr_; }
        ex.sync();
    }

    // H dwv (var-shaped, incl. globals) in operator form: J' W^-1 W^-1 J dwv
    SCVX_HD_NI void H_apply(cgptr dwv, gptr out) {
        SCVX_THIS_LDS();
        cone_map(dwv, tmpc, false);
        ex.sync();
        W_all(tmpc, tmpc, true);
        W_all(tmpc, tmpc, true);
        cone_map_t(tmpc, out);
        ex.sync();
    }

    // Newton step for the cone right-hand side held in tmpc = W^-1 Wibz (scale_pass / corr_rhs_pass): results in dw, dy
    // pred: gx and the banded part of the solve were prepared before / inside build_kkt(true)
    SCVX_HD void newton_solve(bool pred) {
        if (!pred) {
            cone_map_t(tmpc, gx, rx);      // gx = -rx - J' W^-1 Wibz
            mask_fixed(gx);
        }
        SCVX_COUNT(0);
        kkt_solve(gx, ry, dw, dy, -1.0, pred);   // equality right-hand side -ry
#if defined(SCVX_IPM_DEBUG) && !defined(__HIPCC__)
        {
            H_apply(dw, r1);
            const double sgy = Et_apply(dy, tmpl);
            double e1 = 0, e2 = 0, e3 = 0;
            for (int i = 0; i < L.nloc; i++) { double r = gx[i] - r1[i] - tmpl[i]; bool fx = false;
                if (i < 14) fx = fixed_x(0, i); else if (i >= 14 * L.K && i < L.nx) fx = fixed_x(L.K, i - 14 * L.K); else if (i == L.nx + NU * L.K + 1 || i == L.nx + NU * L.K + 2) fx = true;
                if (!fx) { if (i < L.nx + L.nu_) e1 += r * r; else e2 += r * r; } }
            E_apply(dw, tmpy2, true);
            for (int i = 0; i < L.ny; i++) { double r = -ry[i] - tmpy2[i]; e3 += r * r; }
            SCVX_DBG("      kkt res: xu %.2e nu %.2e s %.2e tnu %.2e ttr %.2e ts %.2e | E %.2e\n", sqrt(e1), sqrt(e2), gx[L.iS] - r1[L.iS] - sgy,
                     gx[L.iTNU] - r1[L.iTNU], gx[L.iTTR] - r1[L.iTTR], gx[L.iTS] - r1[L.iTS], sqrt(e3));
        }
#endif
        const int nref = (cur_gate < SCVX_REFINE_FROM && !(pred && !SCVX_REFINE_PRED)) ? C.refine : 0;
        double nr_prev = INFINITY;
        for (int it = 0; it < nref; it++) {
            SCVX_COUNT(1);
            H_apply(dw, r1);
            const double sgy = Et_apply(dy, tmpl);
            ex.sync();
            {
                gptr r1_ = r1; cgptr g_ = gx; cgptr tl = tmpl;
                stream(0, L.nloc, [&](int i) { return D3{g_[i], r1_[i], tl[i]}; }, [&](int i, const D3& v) { r1_[i] = v.a - v.b - v.c; });
            }
            ex.sync();
            if (ex.lane() == 0) {
                r1[L.iS] = gx[L.iS] - r1[L.iS] - sgy;
                r1[L.iTNU] = gx[L.iTNU] - r1[L.iTNU];
                r1[L.iTTR] = gx[L.iTTR] - r1[L.iTTR];
                r1[L.iTS] = gx[L.iTS] - r1[L.iTS];
            }
            ex.sync();
            mask_fixed(r1);
            // The first-row residual r1 = gx - H dw - E'dy is exactly what the step adds to the dual residual
            // (rx+ = (1 - alpha) rx - alpha r1), so it is measured against the dual tolerance: a solve that is already
            // accurate enough skips the correction, one that is not gets up to C.refine of them.
            const double nr1 = sqrt(sumsq(r1, L.nv));
            SCVX_DBG("      refine %d: |r1| %.3e\n", it, nr1);
#if defined(SCVX_IPM_DEBUG) && !defined(__HIPCC__)
            if (it == 0) {   // where the residual of the first solve sits: (block, node, component)
                int am = 0; double mx = 0;
                for (int i = 0; i < L.nv; i++) if (fabs((double)r1[i]) > mx) { mx = fabs((double)r1[i]); am = i; }
                if (am < L.nx) SCVX_DBG("        max at x node %d comp %d (%.3e)\n", am / 14, am % 14, mx);
                else if (am < L.nx + L.nu_) SCVX_DBG("        max at u node %d comp %d (%.3e)\n", (am - L.nx) / NU, (am - L.nx) % NU, mx);
                else SCVX_DBG("        max at index %d of nv %d (%.3e)\n", am, L.nv, mx);
            }
#endif
            if (nr1 <= SCVX_REFINE_STOP * C.tol * (C.wNu > 1.0 ? C.wNu : 1.0)) break;
            if (it > 0 && !(nr1 < 0.5 * nr_prev)) break;   // the corrections have stopped contracting: precision floor
            nr_prev = nr1;
            E_apply(dw, tmpy2, true);
            ex.sync();
            {
                gptr t2 = tmpy2; cgptr ry_ = ry;
                stream(0, L.ny, [&](int i) { return D2{ry_[i], t2[i]}; }, [&](int i, const D2& v) { t2[i] = -v.a - v.b; });
            }
            ex.sync();
            SCVX_COUNT(2);
            kkt_solve(r1, tmpy2, cw, cy);
            {
                gptr dw_ = dw; cgptr cw_ = cw; gptr dy_ = dy; cgptr cy_ = cy;
                stream(0, L.nv, [&](int i) { return D2{dw_[i], cw_[i]}; }, [&](int i, const D2& v) { dw_[i] = v.a + v.b; });
                stream(0, L.ny, [&](int i) { return D2{dy_[i], cy_[i]}; }, [&](int i, const D2& v) { dy_[i] = v.a + v.b; });
            }
            ex.sync();
        }
    }

    // The corrector's Newton step + its scaled direction + the step length, with the refinement's residual check folded in (round 5).
    // newton_solve's check costs a pass over D (E'dy), three cone-vector passes (J dw, W^-1 twice) and J' -- 0.43 MB, in every
    // iteration of the endgame.  Here: E'dy is a by-product of the solve's own final E' product (kkt_solve's eout), and W^-1 J dw is what
    // dir_pass computes anyway (CHECK: it also leaves W^-1 of it in tmpc).  The check is then one J' gather and one combination:
    // 0.12 MB.  Same operator form (J' W^-1 W^-1 J dw), same stopping rules; a correction (rare: ~2 per solve) re-runs the direction pass.
    // Returns the largest step to the cone boundary (dir_pass<false>).
    SCVX_NEWTON_CORR_ATTR double newton_corr() {
        SCVX_THIS_LDS();
        cone_map_t(tmpc, gx, rx, false, false, true);      // gx = -rx - J' W^-1 Wibz, fixed rows zero
        SCVX_COUNT(0);
        const int nref = cur_gate < SCVX_REFINE_FROM ? C.refine : 0;
        if (nref == 0) {
            kkt_solve(gx, ry, dw, dy, -1.0, false);
            return dir_pass<false>();
        }
        const gptr et = tmpv;          // E_loc' dy on the (dx, du) rows; on the nu rows it is dy itself
        double sgy = 0.0;
        kkt_solve(gx, ry, dw, dy, -1.0, false, et, &sgy);
        const int nxu = L.nx + L.nu_;
        double nr_prev = INFINITY, amax = INFINITY;
        for (int it = 0;; it++) {
            if (it >= nref) { amax = dir_pass<false>(); break; }
            amax = dir_pass<false, true>();
            SCVX_COUNT(1);
            cone_map_t(tmpc, r1, nullptr, true);     // r1 = H dw (operator form)
            {
                gptr r1_ = r1; cgptr g_ = gx; cgptr et_ = et; cgptr dy_ = dy;
                stream(0, nxu, [&](int i) { return D3{g_[i], r1_[i], et_[i]}; }, [&](int i, const D3& v) { r1_[i] = v.a - v.b - v.c; });
                stream(nxu, L.nloc, [&](int i) { return D3{g_[i], r1_[i], dy_[i - nxu]}; }, [&](int i, const D3& v) { r1_[i] = v.a - v.b - v.c; });
            }
            ex.sync();
            if (ex.lane() == 0) {
                r1[L.iS] = gx[L.iS] - r1[L.iS] - sgy;
                r1[L.iTNU] = gx[L.iTNU] - r1[L.iTNU];
                r1[L.iTTR] = gx[L.iTTR] - r1[L.iTTR];
                r1[L.iTS] = gx[L.iTS] - r1[L.iTS];
            }
            ex.sync();
            mask_fixed(r1);
            const double nr1 = sqrt(sumsq(r1, L.nv));
            SCVX_DBG("      refine %d: |r1| %.3e\n", it, nr1);
            if (nr1 <= SCVX_REFINE_STOP * C.tol * (C.wNu > 1.0 ? C.wNu : 1.0)) break;
            if (it > 0 && !(nr1 < 0.5 * nr_prev)) break;   // the corrections have stopped contracting: precision floor
            nr_prev = nr1;
            E_apply(dw, tmpy2, true);
            ex.sync();
            {
                gptr t2 = tmpy2; cgptr ry_ = ry;
                stream(0, L.ny, [&](int i) { return D2{ry_[i], t2[i]}; }, [&](int i, const D2& v) { t2[i] = -v.a - v.b; });
            }
            ex.sync();
            SCVX_COUNT(2);
            double sgc = 0.0;
            kkt_solve(r1, tmpy2, cw, cy, 1.0, false, tmpl, &sgc);   // tmpl: kkt_solve's own scratch, free again when its last E' product runs
            sgy += sgc;
            {
                gptr dw_ = dw; cgptr cw_ = cw; gptr dy_ = dy; cgptr cy_ = cy; gptr et_ = et; cgptr ec = tmpl;
                stream(0, L.nv, [&](int i) { return D2{dw_[i], cw_[i]}; }, [&](int i, const D2& v) { dw_[i] = v.a + v.b; });
                stream(0, L.ny, [&](int i) { return D2{dy_[i], cy_[i]}; }, [&](int i, const D2& v) { dy_[i] = v.a + v.b; });
                stream(0, nxu, [&](int i) { return D2{et_[i], ec[i]}; }, [&](int i, const D2& v) { et_[i] = v.a + v.b; });
            }
            ex.sync();
        }
        return amax;
    }

    // Every cone whose point is not at least SCVX_INIT_SHIFT inside gets its own multiple of e added, just enough to be that far
    // inside.  (CVXOPT's conelp adds ONE multiple, 1 + the largest violation, to every cone: on this problem the largest violation
    // is the virtual-control cone's, ~50, and 356 cones that needed nothing start 50 away from their boundaries -- an initial gap of
    // 5e4 that takes five iterations to work off.  Twin, 14-step bench mix: 14.21 -> 13.16 iterations per solve; first failures over
    // 100 random classes 1.16 % -> 1.10 %: tools/k4_fuzz.py, profiles/r03_k4_init_shift.md.)
    SCVX_HD_NI void shift_into_cone(gptr X) {
        SCVX_THIS_LDS();
        all_small([&](auto Dt_, int off, int) {
            constexpr int dm = decltype(Dt_)::value;
            double n = 0;
            for (int i = 1; i < dm; i++) n += X[off + i] * X[off + i];
            const double m = sqrt(n) - X[off];
            if (m >= -p_init_shift) X[off] += p_init_shift + m;
        }, true);
        const int offs[2] = {L.o_nu, L.o_tr};
        const int dims[2] = {14 * L.K + 1, NXU * (L.K + 1) + 1};
        for (int q = 0; q < 2; q++) {
            double n = 0;
            for (int i = 1 + ex.lane(); i < dims[q]; i += ex.nlanes()) n += X[offs[q] + i] * X[offs[q] + i];
            n = ex.sum(n);
            const double m = sqrt(n) - X[offs[q]];
            ex.sync();   // every lane has read the head before lane 0 moves it
            if (ex.lane() == 0 && m >= -p_init_shift) X[offs[q]] += p_init_shift + m;
        }
        ex.sync();
    }

    // ---- the solve.  ic: (rIi, vIi) of this trajectory.  Outputs in V (dx, du, nu, s, ...). ----
    // warm: the previous solve in this workspace was for the same (xbar, ubar, endpoint, D) -- the step it belonged to was
    // rejected -- so its saved iterate may be used as the starting point
    SCVX_HD Result solve(cdptr xbar_, cdptr ubar_, cdptr endpoint_, dcptr D_,
                         double rk_, cdptr ic, gptr work, bool warm = false) {
        xbar = xbar_; ubar = ubar_; endpoint = endpoint_; D = D_; rk = rk_;
        SCVX_TS(tTot_);
        carve(work);
        const int K = L.K;
        // constants of this subproblem
        for (int r = ex.lane(); r < L.ny; r += ex.nlanes()) dk[r] = endpoint[r] - xbar[14 + r];
        {   // A_k' once per subproblem: element (i, j) of A_k sits at 14 j + i in D (column-major), at 14 i + j here
            const dcptr D_ = D; const dptr At_ = At;
            stream(0, 196 * K, [&](int e) { const int k = e / 196, r = e - 196 * k, i = r / 14, j = r - 14 * i; return D_[(size_t)k * DSZ + 14 * j + i]; },
                   [&](int e, double v) { At_[e] = v; });
        }
        for (int k = ex.lane(); k <= K; k += ex.nlanes()) {
            cdptr u = ubar + NU * k;   // the thrust part of the control (rocketland.jl:199 indexes control[1:3])
            const double un = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
            for (int c = 0; c < 3; c++) uhat[3 * k + c] = u[c] / un;  // rocketland.jl:199 (un = 0 -> NaN, as in the reference)
            lb0[k] = C.Tmin - un;
        }
        ex.sync();
        // The ladder.  A solve that ends on its numerical floor above the tolerance (status 1 / 2 / 3) is run again from the cold
        // start with another step rule, up to C.retries times: such failures sit at the precision floor of the Newton system and
        // move with every change of the path -- over 100 random problem classes the sets of (class, trajectory, step) that fail
        // under rules 0..3 overlap in 8 of ~110 each (profiles/r03_k4_retry_ladder.md).  The reference errors unless its solver
        // reports OPTIMAL (rocketland.jl:273-276); a commercial solver's OPTIMAL is the outcome of such safeguards too.  On the
        // sample problems no solve ever fails, so none is retried.  iters = the iterations of all attempts together.
        Result res;
        int iters_all = 0, first_warmed = 0;
        // attempt_solve trades V and Vbest and may leave both names on one buffer (a failed attempt returns its best iterate through V):
        // every attempt starts from the two buffers carve() handed out
        const gptr V_first = V, Vbest_first = Vbest;
        for (int attempt = 0;; attempt++) {
            V = V_first; Vbest = Vbest_first;
            //                          attempt:      0          1     2     3     4     5     6     7
            const double r_step[8] = {SCVX_STEP_FRAC,   0.9,  0.95, SCVX_STEP_FRAC, 0.8, 0.9, 0.85, 0.7};
            const double r_shift[8] = {SCVX_INIT_SHIFT, SCVX_INIT_SHIFT, SCVX_INIT_SHIFT, 1.0, SCVX_INIT_SHIFT, 5.0, 2.0, 0.5};
            const double r_bal[8] = {SCVX_INIT_BALANCE, SCVX_INIT_BALANCE, SCVX_INIT_BALANCE, 0.5, SCVX_INIT_BALANCE, SCVX_INIT_BALANCE, 0.5, 2.0};
            const double r_floor[8] = {0.25, 0.25, 0.25, 0.25, 0.5, 0.25, 0.25, 0.5};
            const int a = attempt < 8 ? attempt : 7;
            p_step_frac = r_step[a]; p_init_shift = r_shift[a]; p_init_balance = r_bal[a]; p_mu_floor = r_floor[a];
            p_sigma_cube = a == 2 || a == 6;
            res = attempt_solve(ic, warm && attempt == 0);
            iters_all += res.iters;
            if (attempt == 0) first_warmed = res.warmed;
            if (!(res.status >= 1 && res.status <= 3) || attempt >= C.retries) { res.attempts = attempt + 1; break; }
            SCVX_DBG("attempt %d ended with status %d at merit %.3e: next rule\n", attempt, res.status, res.merit);
        }
        res.iters = iters_all; res.warmed = first_warmed;
        SCVX_TE(tTot_, 15);
        return res;
    }

    SCVX_HD_NI Result attempt_solve(cdptr ic, bool warm) {
        SCVX_THIS_LDS();
        const int K = L.K;
        zero(V, L.nv); zero(y, L.ny);
        if (ex.lane() == 0) {
            // fixed components: w = bc - xbar (rocketland.jl:109-115)
            V[0] = C.mwet - xbar[0];
            for (int i = 0; i < 3; i++) { V[1 + i] = ic[i] - xbar[1 + i]; V[4 + i] = ic[3 + i] - xbar[4 + i]; V[11 + i] = C.wBi[i] - xbar[11 + i]; }
            cdptr xK = xbar + 14 * K;
            gptr vK = V + 14 * K;
            for (int i = 0; i < 3; i++) { vK[1 + i] = C.rIf[i] - xK[1 + i]; vK[4 + i] = C.vIf[i] - xK[4 + i]; vK[11 + i] = C.wBf[i] - xK[11 + i]; }
            for (int i = 0; i < 4; i++) vK[7 + i] = C.qBIf[i] - xK[7 + i];
            V[L.nx + NU * K + 1] = 0.0 - ubar[NU * K + 1];
            V[L.nx + NU * K + 2] = 0.0 - ubar[NU * K + 2];
        }
        ex.sync();
        Result res; res.status = 1; res.iters = 0; res.merit = INFINITY; res.pobj = 0; res.warmed = 0; res.attempts = 1;
        // Infeasibility that needs no iteration to detect: at node 1 the reference fixes r, v and w (rocketland.jl:109-113) and
        // applies the glideslope and rate cones there (:142-167, k = 1..K): constants against constants.  (The dynamic-
        // pressure extension adds |vIi| <= vmax.)  A violated one has no strictly feasible point.
        {
            const double r1 = ic[0], r2 = ic[1], r3 = ic[2];
            bool bad = sqrt(r2 * r2 + r3 * r3) > r1 * C.itan * (1.0 + 1e-12) + 1e-14;
            bad = bad || sqrt(C.wBi[0] * C.wBi[0] + C.wBi[1] * C.wBi[1] + C.wBi[2] * C.wBi[2]) > C.omMax * (1.0 + 1e-12);
            if (C.vmax > 0.0) bad = bad || sqrt(ic[3] * ic[3] + ic[4] * ic[4] + ic[5] * ic[5]) > C.vmax * (1.0 + 1e-12);
            if (bad) { res.status = 5; return res; }
        }
        cur_gate = INFINITY;
        bigs_ok = false;
        // ... and only while the kept iterate lies inside the new radius (Jtr_w < SCVX_WARM_RADIUS rk): then the radius row is
        // inactive or barely active and the old central path is (nearly) the new one -- 7 iterations instead of 19.  Once the radius
        // binds in earnest the kept iterate is far from the new path and a warm start costs MORE than a cold one (33 vs 21 measured);
        // pulling the kept primal point inside the new radius (a convex combination with the reference point) and keeping
        // the duals breaks down within two iterations on 80 % of such solves.  What does work there is a BLEND of that pulled-in
        // point with the cold starting point (below, SCVX_BLEND_WARM; measured, off by default).
        const bool warmed = warm && wh[0] == 1.0 && Vw[L.iTTR] < SCVX_WARM_RADIUS * rk;
        const bool blend = warm && wh[0] == 1.0 && !warmed;   // same subproblem data, but the new radius cuts the kept optimum off
        res.warmed = warmed ? 1 : 0;
        if (warmed) {
            // same subproblem, new radius: restart from the kept iterate; its radius slack is recomputed (and kept interior)
            copy(V, Vw, L.nv); copy(y, yw, L.ny); copy(S, Sw, L.nc); copy(Z, Zw, L.nc);
            if (ex.lane() == 0) {
                const double srk = rk - V[L.iTTR], flo = 1e-3 * rk;
                S[L.o_rk] = srk > flo ? srk : flo;
            }
            ex.sync();
        } else {
        ex.sync();   // every wavefront has read wh[0] (warmed / blend above)
        if (ex.lane() == 0) wh[0] = 0.0;   // new subproblem data: whatever was kept belongs to another problem
        // ---- initial point (CVXOPT conelp style, W = I): two least-squares problems on one factorisation ----
        //   primal:  min ||s||  s.t. E w = e, s = a(w)        -> w, s     (the cost does not enter)
        //   dual:    min ||z||  s.t. -J'z + E'y + c = 0       -> y, z = -J w'
        // then each of s, z is shifted into the cone interior if it is not already there.
        identity_scaling();
        if (!build_kkt()) { res.status = 2; return res; }
        cone_map(V, S, true);   // a0
        cone_map_t(S, gx);
        for (int i = ex.lane(); i < L.nv; i += ex.nlanes()) gx[i] = -gx[i];
        ex.sync();
        mask_fixed(gx);
        E_apply(V, ry, true);
        for (int i = ex.lane(); i < L.ny; i += ex.nlanes()) r2[i] = -(ry[i] + dk[i]);
        ex.sync();
        kkt_solve(gx, r2, dw, dy);
        for (int i = ex.lane(); i < L.nv; i += ex.nlanes()) V[i] += dw[i];
        ex.sync();
        cone_map(V, S, true);
        // dual
        zero(gx, L.nv);
        if (ex.lane() == 0) { gx[14 * K] = 1.0; gx[L.iTNU] = -C.wNu; gx[L.iTTR] = -0.5; gx[L.iTS] = -1.0; }  // -cost
        ex.sync();
        mask_fixed(gx);
        zero(r2, L.ny);
        kkt_solve(gx, r2, dw, y);
        cone_map(dw, Z, false);  // J w'
        for (int i = ex.lane(); i < L.nc; i += ex.nlanes()) Z[i] = -Z[i];
        ex.sync();
        SCVX_DBG("init: |V|^2 %.12e s %.6e tnu %.6e ttr %.6e ts %.6e |y|^2 %.6e |S|^2 %.12e |Z|^2 %.6e\n", dot(V, V, L.nv), V[L.iS], V[L.iTNU], V[L.iTTR], V[L.iTS], dot(y, y, L.ny), dot(S, S, L.nc), dot(Z, Z, L.nc));
        shift_into_cone(S);
        shift_into_cone(Z);
        if (p_init_balance > 0.0) {
            // Mehrotra's second shift: S += (s'z / 2 e'z) e, Z += (s'z / 2 e's) e balances the complementarity products of
            // the starting point (the dual least-squares solution carries the 1e4 virtual-control weight in a few entries).
            // Measured at B = 8192: 20.9 -> 19.9 iterations per solve, same merit distribution of the returned iterates (factor 0.5
            // after the uniform shift; 1.0 after the per-cone shift: 13.40 -> 13.16 on the twin's bench mix).
            const double sz = dot(S, Z, L.nc);
            double es = 0, ez = 0;
            all_small([&](auto, int off, int) { es += S[off]; ez += Z[off]; }, true);
            if (ex.lane() == 0) { es += S[L.o_nu] + S[L.o_tr]; ez += Z[L.o_nu] + Z[L.o_tr]; }
            es = ex.sum(es); ez = ex.sum(ez);
            const double dsh = p_init_balance * sz / ez, dzh = p_init_balance * sz / es;
            ex.sync();
            all_small([&](auto, int off, int) { S[off] += dsh; Z[off] += dzh; }, true);
            if (ex.lane() == 0) { S[L.o_nu] += dsh; S[L.o_tr] += dsh; Z[L.o_nu] += dzh; Z[L.o_tr] += dzh; }
            ex.sync();
        }
        SCVX_DBG("shifted: |S|^2 %.12e |Z|^2 %.12e\n", dot(S, S, L.nc), dot(Z, Z, L.nc));
        if (blend && SCVX_BLEND_WARM > 0.0) {
            // The solve after a rejected step whose halved radius cuts the kept optimum off: start between the cold point and that
            // optimum pulled inside the new radius (scaled towards the reference point, which keeps it in every path cone when the
            // reference point is), the multipliers likewise; the cold point's margins, scaled, keep the blend interior.  Twin, iterations
            // per solve over the step mix, without -> with (weight 0.8; 0.5 ... 0.95 alike): exo 11.10 -> 10.71, aero 10.25 -> 9.76,
            // K = 100 11.90 -> 11.04; first failures on random classes 0.96 % -> 1.00 % (single attempt).  The same blend for the solve
            // after an ACCEPTED step (primal "stay here", kept multipliers) gained 2 % and is not in.
            const double lm = SCVX_BLEND_WARM, th = SCVX_BLEND_RADIUS * rk / Vw[L.iTTR];
            cone_map(V, tmpc, true);
            for (int i = ex.lane(); i < L.nc; i += ex.nlanes()) tmpc[i] = S[i] - tmpc[i];   // the cold point's margins
            ex.sync();
            for (int i = ex.lane(); i < L.nv; i += ex.nlanes()) {
                bool fx = false;
                if (i < 14) fx = fixed_x(0, i); else if (i >= 14 * K && i < L.nx) fx = fixed_x(K, i - 14 * K);
                else if (i == L.nx + NU * K + 1 || i == L.nx + NU * K + 2) fx = true;
                if (!fx) V[i] = lm * th * Vw[i] + (1.0 - lm) * V[i];
            }
            for (int i = ex.lane(); i < L.ny; i += ex.nlanes()) y[i] = lm * yw[i] + (1.0 - lm) * y[i];
            ex.sync();
            cone_map(V, S, true);
            for (int i = ex.lane(); i < L.nc; i += ex.nlanes()) { S[i] += (1.0 - lm) * tmpc[i]; Z[i] = lm * Zw[i] + (1.0 - lm) * Z[i]; }
            ex.sync();
        }
        }   // cold start

        double best_merit = INFINITY; int best_it = 0;
        // The best iterate is kept without a copy: V and Vbest are two buffers; while the current iterate IS the best one
        // (best_in_V) the next iterate is written into the other buffer and the two pointers trade places.
        bool best_in_V = false;
        bool kept = false;   // this solve's iterate for a possible warm start has been stored
        const int degree = L.ncones;
        bool res_carried = false;                 // rx / ry / their norms were scaled by the last step instead of being due for evaluation
        double nrx_c = 0.0, nry_c = 0.0;          // the carried norms
        for (int it = 1; it <= C.max_iter; it++) {
            res.iters = it;
            // cone residual rz = S - a(V), S'Z, and -- on the same sweep -- the scaling of this iterate
            double gap, nrz2;
            scale_pass(gap, nrz2);
            // dual and equality residuals
            SCVX_TS(tR_);
            // fused (round 5): the two products with the linearisation, E'y and E V, are formed by the factorisation loop, which stages
            // every D_k tile anyway (build_kkt(res)); the stopping tests then follow the factorisation, and the last iteration of a solve
            // factorises once for nothing (1 of ~18).  The first look at a restart from a kept optimum (usually its only one) takes the
            // separate passes: it needs no factorisation at all.
            const bool fused = SCVX_FUSED_RES != 0 && Ex::kFusedResidual && !(warmed && it == 1);
            double nrx, nry;
            bool kkt_ok = true;
            auto eval_residuals = [&]() {
                cone_map_t(Z, rx);
                double nrx2 = 0.0;
                const double sgy = Et_apply(y, rx, rx, 2, 0.0, 0.0, nullptr, &nrx2);     // rx = c + E'y - J'Z on the local part, masked, with its norm
                const double g0 = -rx[L.iS] + sgy, g1 = C.wNu - rx[L.iTNU], g2 = 0.5 - rx[L.iTTR], g3 = 1.0 - rx[L.iTS];
                ex.sync();   // every lane holds the four global entries before lane 0 overwrites them
                if (ex.lane() == 0) { rx[L.iS] = g0; rx[L.iTNU] = g1; rx[L.iTTR] = g2; rx[L.iTS] = g3; }
                ex.sync();
                const double nry2 = E_apply(V, ry, true, dk, 1.0);   // ry = E V + dk
                nrx = sqrt(nrx2 + (g0 * g0 + g1 * g1 + g2 * g2 + g3 * g3)); nry = sqrt(nry2);
            };
            if (!fused) {
                if (res_carried) { nrx = nrx_c; nry = nry_c; }
                else eval_residuals();
            } else {
                cone_map_t(Z, rx, nullptr, false, true);             // rx = -J'Z ...
                if (ex.lane() == 0) {                                // ... + c (the s row still lacks Sg . y)
                    rx[14 * K] += -1.0;
                    rx[L.iTNU] = C.wNu + rx[L.iTNU];
                    rx[L.iTTR] = 0.5 + rx[L.iTTR];
                    rx[L.iTS] = 1.0 + rx[L.iTS];
                }
                ex.sync();
                cone_map_t(tmpc, gx, rx);                            // gx = -(c - J'Z) - J' W^-1 (lam - W^-1 rz)
                kkt_ok = build_kkt(true, true);                      // + / - E'y, ry, the norms; and the factorisation
                if (ex.lane() == 0) { rx[L.iS] += res_sgy; gx[L.iS] -= res_sgy; }
                ex.sync();
                const double g0 = rx[L.iS], g1 = rx[L.iTNU], g2 = rx[L.iTTR], g3 = rx[L.iTS];
                nrx = sqrt(res_nrx2 + (g0 * g0 + g1 * g1 + g2 * g2 + g3 * g3)); nry = sqrt(res_nry2);
            }
            const double pobj = -V[14 * K] + C.wNu * V[L.iTNU] + 0.5 * V[L.iTTR] + V[L.iTS];
            const double nrz = sqrt(nrz2);
            const double relgap = gap / (fabs(pobj) > 1.0 ? fabs(pobj) : 1.0);
            double pres = nry > nrz ? nry : nrz;
            double dres = nrx / (C.wNu > 1.0 ? C.wNu : 1.0);
            double merit = pres > dres ? pres : dres;
            if (relgap > merit) merit = relgap;
            if (res_carried && !(merit >= SCVX_RESID_FRESH_FROM)) {
                // the endgame begins here (or the carried values have gone non-finite): evaluate, and decide on what was evaluated
                eval_residuals();
                res_carried = false;
                pres = nry > nrz ? nry : nrz;
                dres = nrx / (C.wNu > 1.0 ? C.wNu : 1.0);
                merit = pres > dres ? pres : dres;
                if (relgap > merit) merit = relgap;
            }
            SCVX_TE(tR_, 9);
            SCVX_DBG("%3d pobj %+.8e gap %.2e pres %.2e (ry %.2e rz %.2e) dres %.2e\n", it, pobj, gap, pres, nry, nrz, dres);
            if (!(merit == merit) || !(gap == gap)) { res.status = 3; break; }
            cur_gate = pres > relgap ? pres : relgap;
            if (merit < best_merit) {
                best_merit = merit; best_it = it; res.pobj = pobj;
                best_in_V = true;
                ex.sync();
            }
            if (!kept && merit < SCVX_WARM_SAVE * C.tol) {
                copy(Vw, V, L.nv); copy(yw, y, L.ny); copy(Sw, S, L.nc); copy(Zw, Z, L.nc);
                if (ex.lane() == 0) wh[0] = 1.0;
                ex.sync();
                kept = true;
            }
            if (pres < C.tol && dres < C.tol && relgap < C.tol) { res.status = 0; break; }
            // what the best iterate is worth if the solve has to stop here for the reason `why` (1, 2 or 3)
            auto stop_status = [&](int why) { return best_merit < C.tol ? 0 : (best_merit < C.accept ? 4 : why); };
            if (best_merit < C.accept && merit > SCVX_BLOWUP_STOP * best_merit) { res.status = stop_status(2); break; }
            if (it - best_it >= SCVX_STALL_ITERS && best_merit < 1e-5) { res.status = stop_status(2); break; }
            if (it == C.max_iter) { res.status = stop_status(1); break; }
            if (!fused) {
                cone_map_t(tmpc, gx, rx, false, false, true);   // predictor right-hand side gx = -rx - J' W^-1 (lam - W^-1 rz), fixed rows zero; solved with the border
                kkt_ok = build_kkt(true);
            }
            if (!kkt_ok) { SCVX_DBG("    factorisation failed\n"); res.status = stop_status(2); break; }
            const double mu = gap / degree;
            { SCVX_TS(tN_); newton_solve(true); SCVX_TE(tN_, 10); }   // predictor: affine right-hand side -lam o lam
            double alpha = dir_pass<true>();
            if (alpha > 1.0) alpha = 1.0;
            // centering parameter (1 - alpha_aff)^4: the usual cube needs as many iterations (19.9 vs 19.8 per solve at
            // B = 8192) but leaves 5.7 % of the returned iterates above merit 1e-7, the fourth power 2.5 %
            const double om = 1.0 - alpha;
            const double sig = p_sigma_cube ? om * om * om : (om * om) * (om * om);   // the usual cube: 19.87 vs 19.67 iterations per solve (tools/twin_stats.py)
            SCVX_DBG("    aff alpha %.6e |dw|^2 %.6e ds %.6e dtnu %.6e dttr %.6e\n", alpha, dot(dw, dw, L.nv), dw[L.iS], dw[L.iTNU], dw[L.iTTR]);
            // Do not aim below the gap the tolerance asks for: the NT scalings of the active cones grow like 1 / sqrt(mu) and the
            // Newton system (entries v0^4 / beta^2) passes the precision of a double a little below mu = tol |pobj| / degree; a
            // predictor that finds alpha_aff ~ 1 there would target 1e-4 mu and the step that follows pollutes the dual residual
            // beyond repair (the solves that used to end "stalled at merit 1.0..5e-8").
            double smu = sig * mu;
#ifndef SCVX_MU_FLOOR
#define SCVX_MU_FLOOR 0.25  // of tol |pobj| / degree.  Twin, first failures on 60 random classes: 0 -> 2.25 %, 0.25 -> 1.77 %, 0.5 -> 1.71 %, 1 -> 4.4 %;
                            // full-run parity with the oracle (14 steps, x): 6.4e-5 at 0 / 0.25, 1.35e-4 at 0.5 (the optimum is flat) -> 0.25
#endif
            {
                const double apo = fabs(pobj) > 1.0 ? fabs(pobj) : 1.0;
                const double mu_floor = p_mu_floor * C.tol * apo / degree;
                if (smu < mu_floor) smu = mu_floor < mu ? mu_floor : mu;
#if defined(SCVX_HOLD_MU)
                // gap and primal residual already meet the tolerance, only the dual residual is left: a pure centring step
                // (target = the current mu) restores feasibility without pushing the scalings further
                if (relgap < C.tol && pres < C.tol) smu = mu;
#endif
            }
            corr_rhs_pass(smu);
#if SCVX_REFINE_FUSED
            { SCVX_TS(tN_); alpha = p_step_frac * newton_corr(); SCVX_TE(tN_, 10); }
#else
            { SCVX_TS(tN_); newton_solve(false); SCVX_TE(tN_, 10); }
            alpha = p_step_frac * dir_pass<false>();
#endif
            if (alpha > 1.0) alpha = 1.0;
            SCVX_DBG("    cmb alpha %.6e |dw|^2 %.6e\n", alpha, dot(dw, dw, L.nv));
            if (!(alpha == alpha)) { res.status = stop_status(3); break; }
            if (alpha < 1e-9) { res.status = stop_status(2); break; }
            {
                const gptr Vn = best_in_V ? Vbest : V;
                update_pass(alpha, Vn);   // S, Z (from the old V), the new V and y
                if (best_in_V) { Vbest = V; V = Vn; best_in_V = false; }   // Vbest now holds the best iterate, V the new one
            }
            res_carried = false;
            if (SCVX_RESID_UPDATE != 0 && !fused && merit >= SCVX_RESID_FRESH_FROM && (it & 7) != 0) {
                // far from the optimum: the next iterate's linear residuals are (1 - alpha) times this one's (see SCVX_RESID_UPDATE)
                const double om = 1.0 - alpha;
                {
                    gptr rx_ = rx; gptr ry_ = ry;
                    stream(0, L.nv, [&](int i) { return rx_[i]; }, [&](int i, double v) { rx_[i] = om * v; });
                    stream(0, L.ny, [&](int i) { return ry_[i]; }, [&](int i, double v) { ry_[i] = om * v; });
                }
                ex.sync();
                nrx_c = om * nrx; nry_c = om * nry;
                res_carried = true;
            }
        }
        res.merit = best_merit;
        if (best_it > 0 && !best_in_V) V = Vbest;   // the caller reads the solution through V
        return res;
    }
};

}  // namespace ipm
}  // namespace scvx
