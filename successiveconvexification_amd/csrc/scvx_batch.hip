// Batched SCvx on the device: the conic subproblem kernel (K4), the glue kernels (K3/K5) and the
// batch entry points of include/scvx.h.
//
// One Rocketland.solve_step (rocketland.jl:226-321) for B independent trajectories is the kernel chain
//     socp_kernel      : MOI data update + MOI.optimize! + primal extraction     (:245-283)
//     candidate_kernel : x = about + dx, u = about + du, sigma + dsigma          (:278-280, :317)
//     propagate_kernel : K predict_state calls per trajectory (K2)               (:289)
//     tr_update_kernel : jK, lK, rho test, accept / reject, radius update        (:286-313)
//     linearize_kernel : Dynamics.linearize_dynamics on the new reference (K1)   (:318)
// all on one stream, nothing returning to the host in between.
//
// The executors, the kernel body and the kernel templates of the conic solve are in scvx_socp.hpp; this file instantiates them for
// the reference's model (control_dim = 3), scvx_socp_fin.hip for the fin extension (control_dim = 5).
// socp_kernel runs ONE 64-LANE WAVEFRONT PER TRAJECTORY over the portable interior-point core
// (scvx_ipm_core.hpp): lanes split the per-node cone work, the 14x14 tile arithmetic of the block-
// tridiagonal Schur complement and the long dot products of the two big trust-region cones; reductions
// are cross-lane shuffles; the 14x14 pivot tiles of the block Cholesky live in LDS.  Each trajectory's
// working set is one contiguous slab of HBM so every strided lane loop is a coalesced access.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <vector>

#include <cstdio>
#include "scvx_socp.hpp"

using scvx::fail;

namespace scvx {

// the reference's model (control_dim = 3) keeps its two named kernels: profiles and tools refer to them
__global__ __launch_bounds__(64, SCVX_K4_OCC) void socp_kernel(ipm::Consts C, int B, size_t work_stride,
                                                  const double* __restrict__ x, const double* __restrict__ u,
                                                  const double* __restrict__ endpoint, const double* __restrict__ deriv,
                                                  const double* __restrict__ rk, const double* __restrict__ ic,
                                                  const int* __restrict__ active, double* __restrict__ work,
                                                  double* __restrict__ sol, double* __restrict__ nu,
                                                  double* __restrict__ info, const int* __restrict__ step_status,
                                                  double* __restrict__ ttr, double* __restrict__ acc) {
    socp_body<WaveEx>(C, B, work_stride, x, u, endpoint, deriv, rk, ic, active, work, sol, nu, info, step_status, ttr, acc);
}
// the same solve on float derivative tiles (scvx_batch_set_linearization_f32)
__global__ __launch_bounds__(64, SCVX_K4_OCC) void socp_lin32_kernel(ipm::Consts C, int B, size_t work_stride,
                                                  const double* __restrict__ x, const double* __restrict__ u,
                                                  const double* __restrict__ endpoint, const float* __restrict__ deriv,
                                                  const double* __restrict__ rk, const double* __restrict__ ic,
                                                  const int* __restrict__ active, double* __restrict__ work,
                                                  double* __restrict__ sol, double* __restrict__ nu,
                                                  double* __restrict__ info, const int* __restrict__ step_status,
                                                  double* __restrict__ ttr, double* __restrict__ acc) {
    socp_body<WaveEx, float>(C, B, work_stride, x, u, endpoint, deriv, rk, ic, active, work, sol, nu, info, step_status, ttr, acc);
}


// cand = about + step (x, u in one contiguous [B][(K+1)*17+1] trajectory record, sigma last)
__global__ void candidate_kernel(int B, int nrec, const double* __restrict__ traj, const double* __restrict__ sol,
                                 double* __restrict__ cand) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (size_t)B * nrec) cand[i] = traj[i] + sol[i];
}

// split / join between the contiguous trajectory record and the (x, u, sigma) arrays the kernels read
__global__ void unpack_kernel(int B, int K, int NU, const double* __restrict__ rec, double* __restrict__ x,
                              double* __restrict__ u, double* __restrict__ sigma) {
    const int nrec = (K + 1) * (14 + NU) + 1, nx = (K + 1) * 14, nu = (K + 1) * NU;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * nrec) return;
    const size_t b = i / nrec;
    const int r = (int)(i - b * nrec);
    const double v = rec[i];
    if (r < nx) x[b * nx + r] = v;
    else if (r < nx + nu) u[b * nu + (r - nx)] = v;
    else sigma[b] = v;
}

struct TrParams {
    double wNu, rh0, rh1, rh2, alph, bet, ri, nuTol, delTol;
    int K, imax, NU, pad;
};

// One wavefront per trajectory: rocketland.jl:286-313 plus the commit of the accepted candidate.
// status codes: include/scvx.h.  out[b] = {nu_norm, dJ}
__global__ __launch_bounds__(64) void tr_update_kernel(TrParams P, int B, const double* __restrict__ cand,
                                                       const double* __restrict__ xprop, const double* __restrict__ nu,
                                                       const double* __restrict__ info, double* __restrict__ traj,
                                                       double* __restrict__ rk, double* __restrict__ cost,
                                                       int* __restrict__ iter, int* __restrict__ status,
                                                       const int* mask /* may be `active` or `live` itself */, int* active,
                                                       int* live, double* __restrict__ out, double* __restrict__ acc,
                                                       int* __restrict__ nlive, int* __restrict__ k1skip) {
    const int b = blockIdx.x;
    if (b >= B) return;
    // not stepped by this call (failed earlier, or converged inside scvx_solve): status[b] keeps saying why and
    // out[b] keeps the (nu, dJ) of the last step this trajectory did take; its iterate is unchanged: K1 skips it
    if (!mask[b]) {
        if (threadIdx.x == 0) k1skip[b] = SCVX_ST_REJECTED;
        return;
    }
    const int K = P.K, nrec = (K + 1) * (14 + P.NU) + 1;
    const double* c = cand + (size_t)b * nrec;
    const double* xp = xprop + (size_t)b * K * 14;
    const double* nv = nu + (size_t)b * K * 14;
    double d2 = 0, n2 = 0;
    for (int i = threadIdx.x; i < K * 14; i += 64) {
        const double d = c[14 + i] - xp[i];  // x_{k+1} - predict_state(x_k, u_k, u_{k+1}, sigma + dsigma)
        d2 += d * d;
        n2 += nv[i] * nv[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { d2 += __shfl_xor(d2, o, 64); n2 += __shfl_xor(n2, o, 64); }
    const double mf = c[14 * K];  // x[1, K+1]
    const double jK = -mf + P.wNu * sqrt(d2);
    const double nun = sqrt(n2);
    const double lK = -mf + P.wNu * nun;
    const int sstat = (int)info[4 * b];
    int st;
    bool accept = false;
    double next_rk = rk[b], dJ = INFINITY;
    if (sstat != 0 && sstat != 4) {   // 4 = almost optimal, inside the solver's acceptance band (scvx_solver_opts.accept_tol)
        st = sstat == 3 ? SCVX_ST_NONFINITE : (sstat == 5 ? SCVX_ST_INFEASIBLE : SCVX_ST_SOLVER);  // rocketland.jl:273-276: error(...)
    } else if (!(jK == jK) || !(lK == lK)) {
        st = SCVX_ST_NONFINITE;
    } else {
        const double jKm = cost[b];
        const double djk = jKm - jK, dlk = jKm - lK;
        const double rhk = djk / dlk;  // NaN on the first call (cost = Inf): falls through to the grow branch
        if (rk[b] == INFINITY) { next_rk = P.ri; accept = true; dJ = NAN; }
        else if (rhk < P.rh0) { next_rk = rk[b] / P.alph; accept = false; dJ = INFINITY; }
        else {
            accept = true;
            dJ = djk;
            if (rhk < P.rh1) next_rk = rk[b] / P.alph;
            else if (P.rh1 <= rhk && rhk < P.rh2) next_rk = rk[b];
            else next_rk = P.bet * rk[b];
        }
        st = accept ? SCVX_ST_RUNNING : SCVX_ST_REJECTED;
    }
    if (accept) {
        double* t = traj + (size_t)b * nrec;
        for (int i = threadIdx.x; i < nrec; i += 64) t[i] = c[i];
    }
    if (threadIdx.x == 0) {
        const int it = iter[b] + 1;
        iter[b] = it;
        const int was_live = live[b];
        if (st == SCVX_ST_SOLVER || st == SCVX_ST_NONFINITE || st == SCVX_ST_INFEASIBLE) {
            active[b] = 0;   // frozen: the reference stops with an error here
            live[b] = 0;
        } else {
            rk[b] = next_rk;
            if (accept) cost[b] = jK;
            // solve_problem's loop test (rocketland.jl:436): stop when nu and dJ are both within tolerance.  Only
            // scvx_solve acts on it (live); solve_step itself has no notion of convergence and keeps stepping.
            if (accept && nun <= P.nuTol && dJ <= P.delTol) { st = SCVX_ST_CONVERGED; live[b] = 0; }
        }
        if (was_live && !live[b]) atomicSub(nlive, 1);   // device-side count of the trajectories scvx_solve still steps
        status[b] = st;
        k1skip[b] = accept && st != SCVX_ST_SOLVER && st != SCVX_ST_NONFINITE && st != SCVX_ST_INFEASIBLE ? SCVX_ST_RUNNING : SCVX_ST_REJECTED;   // re-linearise only a new reference point (rocketland.jl:318; :301 returns before it)
        out[2 * b] = nun;
        out[2 * b + 1] = dJ;
        atomicAdd(acc + ACC_TRAJ_STEPS, 1.0);
        if (st == SCVX_ST_REJECTED) atomicAdd(acc + ACC_REJECTED, 1.0);
        else if (st == SCVX_ST_CONVERGED) atomicAdd(acc + ACC_CONVERGED, 1.0);
        else if (st != SCVX_ST_RUNNING) atomicAdd(acc + ACC_FAILED, 1.0);
    }
}

// scvx_batch_reset: the scalars of create_initial (rocketland.jl:38) for every trajectory
__global__ void reset_scalars_kernel(int B, double* __restrict__ rk, double* __restrict__ cost, int* __restrict__ iter,
                                     int* __restrict__ status, int* __restrict__ active, int* __restrict__ live,
                                     double* __restrict__ out, double* __restrict__ info) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    rk[i] = 100.0; cost[i] = INFINITY; iter[i] = 0; status[i] = SCVX_ST_RUNNING; active[i] = 1; live[i] = 1;
    out[2 * i] = 0.0; out[2 * i + 1] = 0.0;
    for (int q = 0; q < 4; q++) info[4 * i + q] = 0.0;
}

// live = active (start of a solve_problem loop); *nlive (zeroed by the caller) = how many
__global__ void copy_flags_kernel(int B, const int* __restrict__ src, int* __restrict__ dst, int* __restrict__ nlive) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int v = i < B ? src[i] : 0;
    if (i < B) dst[i] = v;
    const unsigned long long m = __ballot(v != 0);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(nlive, (int)__popcll(m));
}

}  // namespace scvx

// ---------------------------------------------------------------------------------------------------
struct scvx_batch {
    scvx_ctx* ctx = nullptr;
    int B = 0, K = 0, nrec = 0;
    int NU = 3, dsz = 294;   // control_dim of the context's model; doubles per derivative tile, 14 * (14 + 2 NU + 1)
    scvx_solver_opts opts{};
    scvx::ipm::Consts C{};
    scvx::TrParams tr{};
    size_t work_stride = 0;
    // device state
    double *traj = nullptr, *cand = nullptr, *sol = nullptr;      // [B][nrec]
    double *traj0 = nullptr;                                      // the straight-line guess of scvx_batch_init (scvx_batch_reset)
    double *x = nullptr, *u = nullptr, *sigma = nullptr;          // split views of traj (kernels' input layout)
    double *cx = nullptr, *cu = nullptr, *csigma = nullptr;       // split views of cand
    double *endpoint = nullptr, *deriv = nullptr, *xprop = nullptr, *nu = nullptr;
    float* deriv_f = nullptr;   // the derivative tiles in float (scvx_batch_set_linearization_f32): then `deriv` is not allocated
    double *rk = nullptr, *cost = nullptr, *ic = nullptr, *info = nullptr, *out = nullptr, *work = nullptr;
    double *ttr = nullptr;   // trust-region norm bound at the last optimum (reuse_inactive_tr)
    double *acc = nullptr;   // scvx::ACC_N running totals (scvx_batch_get_step_stats)
    int *k1skip = nullptr;   // per trajectory: >= SCVX_ST_REJECTED = the reference point did not change in the last step (K1 skips it)
    int *d_nlive = nullptr;  // device-side count of live trajectories (scvx_solve), mirrored asynchronously into pinned h_nlive[2]
    int *h_nlive = nullptr;
    hipEvent_t ev_nlive[2] = {nullptr, nullptr};
    int *iter = nullptr, *status = nullptr;
    int *active = nullptr;   // 0 once a trajectory has failed (solver / non-finite): never stepped again
    int *live = nullptr;     // active and not yet converged: the trajectories scvx_solve still steps
    int device = 0;          // cached: scvx_batch_destroy must not touch a context that may already be gone
    int nactive_host = -1;   // upper bound of the trajectories solve_step steps, as far as the host knows (set_flags counts them; -1 = all)
    int nlive_hint = -1;     // trajectories scvx_solve still steps (host view, a few steps old); -1 = all: picks the conic solver's executor
    bool initialised = false;
    bool profiling = false;
    std::vector<hipEvent_t> events;  // pool, 7 per profiled step, reused across scvx_batch_get_profile calls
    size_t nmarks = 0;               // marks recorded since the last scvx_batch_get_profile
};

namespace {


void rotation_between_e1(const double* b, double* q) {
    // Rotations.rotation_between([1,0,0], b) as [w,x,y,z] (initial_solve.jl:121-122)
    const double nb = std::sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
    double w = nb + b[0];
    double v[3];
    if (std::fabs(w) < 100 * std::numeric_limits<double>::epsilon()) {
        v[0] = 0; v[1] = 0; v[2] = 1;  // any vector perpendicular to e1
        w = 0;
    } else {
        v[0] = 0; v[1] = -b[2]; v[2] = b[1];  // e1 x b
    }
    const double n = std::sqrt(w * w + v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    q[0] = w / n; q[1] = v[0] / n; q[2] = v[1] / n; q[3] = v[2] / n;
}

template <class T>
int dmalloc(scvx_ctx* ctx, T** p, size_t n) {
    SCVX_HIP(ctx, hipMalloc((void**)p, n * sizeof(T)));
    return SCVX_OK;
}

int split_views(scvx_batch* b, const double* rec, double* x, double* u, double* sigma) {
    const size_t n = (size_t)b->B * b->nrec;
    hipLaunchKernelGGL(scvx::unpack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, b->ctx->stream, b->B, b->K, b->NU,
                       rec, x, u, sigma);
    SCVX_HIP(b->ctx, hipGetLastError());
    return SCVX_OK;
}

// Wavefronts per trajectory of the conic solve.  One wavefront per trajectory is the throughput form; every form is
// compiled for 2 wavefronts per SIMD (<= 256 VGPRs, nothing spills), i.e. 8 per CU are resident.  While ALL trajectories
// of the batch are resident at once a solve is latency-bound and more wavefronts per trajectory shorten it (the sweeps
// and E / E' products spread over their lanes; the 14x14 tile arithmetic stays on the first, the factorisation pipelines
// over two), so: 4 wavefronts while B <= 2 per CU (512 on 256 CUs), 2 while B <= 4 per CU (1,024), else 1 -- one more
// trajectory than that starts a second round of blocks (measured: B = 1,024 -> 1,152 with 2 wavefronts 9.9 -> 12.8 ms).
// B-sweep: profiles/r02_bsweep_occ2.md; SCVX_K4_WAVES = 1 / 2 / 4 forces.
int socp_waves(int B, int num_cus) {
    if (const char* v = std::getenv("SCVX_K4_WAVES"); v && *v) {   // an empty value means "not set"
        const int w = std::atoi(v);
        return w >= 4 ? 4 : (w >= 2 ? 2 : 1);   // 8 and 16 wavefronts per trajectory were measured: no faster than 4 at B = 1..64 (barrier cost)
    }
    const int cus = num_cus > 0 ? num_cus : 256;
    return B <= 2 * cus ? 4 : (B <= 4 * cus ? 2 : 1);
}

template <int NW>
void launch_socp_block(scvx_batch* b, const int* mask) {
    constexpr int NU = 3;
    if (b->deriv_f)
        hipLaunchKernelGGL((scvx::socp_block_kernel<NW, float, NU>), dim3(b->B), dim3(64 * NW), 0, b->ctx->stream, b->C, b->B, b->work_stride,
                           b->x, b->u, b->endpoint, b->deriv_f, b->rk, b->ic, mask, b->work, b->sol, b->nu, b->info, b->status, b->ttr, b->acc);
    else
        hipLaunchKernelGGL((scvx::socp_block_kernel<NW, double, NU>), dim3(b->B), dim3(64 * NW), 0, b->ctx->stream, b->C, b->B, b->work_stride,
                           b->x, b->u, b->endpoint, b->deriv, b->rk, b->ic, mask, b->work, b->sol, b->nu, b->info, b->status, b->ttr, b->acc);
}

// K1 for the batch's iterate, into the derivative buffer of the batch's mode
hipError_t relinearize(scvx_batch* b, const int* skip) {
    const double dt = 1.0 / (b->K + 1);
    if (b->deriv_f)
        return scvx::launch_linearize_store_f32(b->ctx, b->B, b->K, b->x, b->u, b->sigma, dt, b->endpoint, b->deriv_f, b->ctx->stream, skip);
    return scvx::launch_linearize(b->ctx, b->B, b->K, b->x, b->u, b->sigma, dt, b->endpoint, b->deriv, b->ctx->stream, skip);
}

int enqueue_socp(scvx_batch* b, const int* mask) {
    // scvx_solve's tail: once few trajectories are still live the solve is latency-bound again, and the executors with
    // several wavefronts per trajectory (dead blocks return at once) finish a step in half the time
    const int w = socp_waves(b->nlive_hint >= 0 && b->nlive_hint < b->B ? (b->nlive_hint > 0 ? b->nlive_hint : 1) : b->B, b->ctx->num_cus);
    if (b->NU == 5) {   // fin extension: the same three executors, instantiated for control_dim = 5 in scvx_socp_fin.hip
        scvx::SocpLaunch a{b->C, b->B, b->work_stride, b->x, b->u, b->endpoint, b->deriv, b->deriv_f, b->rk, b->ic, mask,
                           b->work, b->sol, b->nu, b->info, b->status, b->ttr, b->acc, b->ctx->stream};
        scvx::launch_socp_fin(a, w);
    } else if (w == 4) launch_socp_block<4>(b, mask);
    else if (w == 2) launch_socp_block<2>(b, mask);
    else if (b->deriv_f)
        hipLaunchKernelGGL(scvx::socp_lin32_kernel, dim3(b->B), dim3(64), 0, b->ctx->stream, b->C, b->B, b->work_stride, b->x, b->u,
                           b->endpoint, b->deriv_f, b->rk, b->ic, mask, b->work, b->sol, b->nu, b->info, b->status, b->ttr, b->acc);
    else
        hipLaunchKernelGGL(scvx::socp_kernel, dim3(b->B), dim3(64), 0, b->ctx->stream, b->C, b->B, b->work_stride, b->x, b->u,
                           b->endpoint, b->deriv, b->rk, b->ic, mask, b->work, b->sol, b->nu, b->info, b->status, b->ttr, b->acc);
    SCVX_HIP(b->ctx, hipGetLastError());
    return SCVX_OK;
}

constexpr size_t PROF_MAX_STEPS = 4096;   // profiled steps kept between two scvx_batch_get_profile calls

int mark(scvx_batch* b) {
    if (!b->profiling || b->nmarks >= 7 * PROF_MAX_STEPS) return SCVX_OK;
    if (b->nmarks == b->events.size()) {
        hipEvent_t e;
        SCVX_HIP(b->ctx, hipEventCreate(&e));
        b->events.push_back(e);
    }
    SCVX_HIP(b->ctx, hipEventRecord(b->events[b->nmarks], b->ctx->stream));
    b->nmarks++;
    return SCVX_OK;
}

int enqueue_step(scvx_batch* b, const int* mask) {
    scvx_ctx* ctx = b->ctx;
    hipStream_t st = ctx->stream;
    const double dt = 1.0 / (b->K + 1);
    int rc = mark(b);
    if (rc) return rc;
    rc = enqueue_socp(b, mask);
    if (rc) return rc;
    if ((rc = mark(b))) return rc;
    const size_t n = (size_t)b->B * b->nrec;
    hipLaunchKernelGGL(scvx::candidate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, b->B, b->nrec, b->traj,
                       b->sol, b->cand);
    SCVX_HIP(ctx, hipGetLastError());
    rc = split_views(b, b->cand, b->cx, b->cu, b->csigma);
    if (rc) return rc;
    if ((rc = mark(b))) return rc;
    SCVX_HIP(ctx, scvx::launch_propagate(ctx, b->B, b->K, b->cx, b->cu, b->csigma, dt, b->xprop, st));
    if ((rc = mark(b))) return rc;
    hipLaunchKernelGGL(scvx::tr_update_kernel, dim3(b->B), dim3(64), 0, st, b->tr, b->B, b->cand, b->xprop, b->nu, b->info,
                       b->traj, b->rk, b->cost, b->iter, b->status, mask, b->active, b->live, b->out, b->acc, b->d_nlive, b->k1skip);
    SCVX_HIP(ctx, hipGetLastError());
    if ((rc = mark(b))) return rc;
    rc = split_views(b, b->traj, b->x, b->u, b->sigma);
    if (rc) return rc;
    if ((rc = mark(b))) return rc;
    // rocketland.jl:318; a rejected step returns before it (:301, about / dynam kept): those trajectories are skipped
    SCVX_HIP(ctx, relinearize(b, b->k1skip));
    if ((rc = mark(b))) return rc;
    return SCVX_OK;
}

int check_batch(scvx_batch* b, bool need_init) {
    if (!b || !b->ctx) return SCVX_ERR_ARG;
    if (need_init && !b->initialised) return fail(b->ctx, SCVX_ERR_STATE, "call scvx_batch_init first");
    if (b->ctx->prob.aero_kind == 1 && !b->ctx->dyn.aero)
        return fail(b->ctx, SCVX_ERR_STATE, "AtmosphericData problem: call scvx_set_aero_table first");
    hipError_t e = hipSetDevice(b->ctx->device);
    if (e != hipSuccess) return fail(b->ctx, SCVX_ERR_HIP, "hipSetDevice failed");
    return SCVX_OK;
}

}  // namespace

extern "C" {

int scvx_solver_default_opts(scvx_solver_opts* o) {
    if (!o) return SCVX_ERR_ARG;
    o->max_iter = 60;
    o->refine = 6;
    o->tol = 1e-8;
    o->accept_tol = 1e-8;   // = tol: anything but OPTIMAL is an error, as in the reference (rocketland.jl:273-276); widen to opt in to status 4
    o->reuse_inactive_tr = 0;
    o->warm_start = 1;
    o->retries = 5;
    o->reserved0 = 0;
    return SCVX_OK;
}

int scvx_batch_create(scvx_ctx* ctx, int B, scvx_batch** out) {
    if (!ctx || !out) return SCVX_ERR_ARG;
    *out = nullptr;
    if (B < 1) return fail(ctx, SCVX_ERR_ARG, "B >= 1 required");
    SCVX_HIP(ctx, hipSetDevice(ctx->device));
    scvx_batch* b = new (std::nothrow) scvx_batch();
    if (!b) return SCVX_ERR_NOMEM;
    b->ctx = ctx;
    b->device = ctx->device;
    b->B = B;
    const scvx_problem& p = ctx->prob;
    const int K = p.K;
    b->K = K;
    b->NU = scvx_control_dim(ctx);
    b->dsz = 14 * (14 + 2 * b->NU + 1);
    b->nrec = (K + 1) * (14 + b->NU) + 1;
    scvx_solver_default_opts(&b->opts);
    scvx::ipm::Consts& C = b->C;
    const double d2r = M_PI / 180.0;
    C.K = K; C.max_iter = b->opts.max_iter; C.refine = b->opts.refine; C.pad = 0; C.tol = b->opts.tol;
    C.accept = b->opts.accept_tol;
    C.pad = b->opts.reuse_inactive_tr;
    C.warm = b->opts.warm_start;
    C.retries = b->opts.retries;
    C.itan = 1.0 / std::tan(p.gammaGs * d2r);                       // rocketland.jl:63
    C.sqcm = std::sqrt((1.0 - std::cos(p.thetaMax * d2r)) / 2.0);   // :64
    C.icos = 1.0 / std::cos(p.deltaMax * d2r);                      // :65
    C.Tmax = p.Tmax; C.Tmin = p.Tmin; C.omMax = p.omMax; C.mdry = p.mdry; C.wNu = p.wNu; C.mwet = p.mwet;
    // dynamic pressure 1/2 rho |v|^2 <= dpMax (master.jl:27,30 carry the fields, rocketland.jl:211 leaves the constraint
    // as a todo): enforced only when the problem asks for it
    C.vmax = (p.model_flags & SCVX_MODEL_DPMAX) ? std::sqrt(2.0 * p.dpMax / p.rho) : 0.0;
    C.finmxf = p.finmxf;
    for (int i = 0; i < 3; i++) { C.rIf[i] = p.rIf[i]; C.vIf[i] = p.vIf[i]; C.wBi[i] = p.wBi[i]; C.wBf[i] = p.wBf[i]; }
    for (int i = 0; i < 4; i++) C.qBIf[i] = p.qBIf[i];
    scvx::TrParams& T = b->tr;
    T.wNu = p.wNu; T.rh0 = p.rh0; T.rh1 = p.rh1; T.rh2 = p.rh2; T.alph = p.alph; T.bet = p.bet; T.ri = p.ri;
    T.nuTol = p.nuTol; T.delTol = p.delTol; T.K = K; T.imax = p.imax; T.NU = b->NU; T.pad = 0;
    scvx::ipm::Layout L;
    L.init(K, C.vmax > 0.0, b->NU);
    b->work_stride = (L.work_doubles() + 7) & ~(size_t)7;
    if (const char* v = std::getenv("SCVX_WORK_PAD"); v && *v) b->work_stride += (size_t)std::atoi(v) & ~(size_t)7;   // diagnostic: extra doubles between the slabs
    const size_t nB = (size_t)B;
    int rc = 0;
    rc |= dmalloc(ctx, &b->traj, nB * b->nrec);
    rc |= dmalloc(ctx, &b->cand, nB * b->nrec);
    rc |= dmalloc(ctx, &b->traj0, nB * b->nrec);
    rc |= dmalloc(ctx, &b->sol, nB * b->nrec);
    rc |= dmalloc(ctx, &b->x, nB * (K + 1) * 14);
    rc |= dmalloc(ctx, &b->u, nB * (K + 1) * b->NU);
    rc |= dmalloc(ctx, &b->sigma, nB);
    rc |= dmalloc(ctx, &b->cx, nB * (K + 1) * 14);
    rc |= dmalloc(ctx, &b->cu, nB * (K + 1) * b->NU);
    rc |= dmalloc(ctx, &b->csigma, nB);
    rc |= dmalloc(ctx, &b->endpoint, nB * K * 14);
    rc |= dmalloc(ctx, &b->deriv, nB * K * b->dsz);
    rc |= dmalloc(ctx, &b->xprop, nB * K * 14);
    rc |= dmalloc(ctx, &b->nu, nB * K * 14);
    rc |= dmalloc(ctx, &b->rk, nB);
    rc |= dmalloc(ctx, &b->cost, nB);
    rc |= dmalloc(ctx, &b->ic, nB * 6);
    rc |= dmalloc(ctx, &b->info, nB * 4);
    rc |= dmalloc(ctx, &b->out, nB * 2);
    rc |= dmalloc(ctx, &b->ttr, nB);
    rc |= dmalloc(ctx, &b->acc, (size_t)scvx::ACC_N);
    rc |= dmalloc(ctx, &b->d_nlive, (size_t)1);
    rc |= dmalloc(ctx, &b->k1skip, nB);
    if (!rc && (hipHostMalloc((void**)&b->h_nlive, 2 * sizeof(int), hipHostMallocDefault) != hipSuccess ||
                hipEventCreateWithFlags(&b->ev_nlive[0], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&b->ev_nlive[1], hipEventDisableTiming) != hipSuccess)) rc = SCVX_ERR_HIP;
    if (!rc && hipMemset(b->acc, 0, sizeof(double) * scvx::ACC_N) != hipSuccess) rc = SCVX_ERR_HIP;
    rc |= dmalloc(ctx, &b->work, nB * b->work_stride);
    rc |= dmalloc(ctx, &b->iter, nB);
    rc |= dmalloc(ctx, &b->status, nB);
    rc |= dmalloc(ctx, &b->active, nB);
    rc |= dmalloc(ctx, &b->live, nB);
    if (rc) {
        scvx_batch_destroy(b);
        return fail(ctx, SCVX_ERR_NOMEM, "device allocation failed for the batch (" + std::to_string(nB * b->work_stride * 8 >> 20) + " MiB of solver workspace)");
    }
    *out = b;
    return SCVX_OK;
}

void scvx_batch_destroy(scvx_batch* b) {
    if (!b) return;
    (void)hipSetDevice(b->device);
    for (hipEvent_t e : b->events) (void)hipEventDestroy(e);
    for (hipEvent_t e : b->ev_nlive) if (e) (void)hipEventDestroy(e);
    if (b->h_nlive) (void)hipHostFree(b->h_nlive);
    void* ptrs[] = {b->traj0, b->traj, b->cand, b->sol, b->x, b->u, b->sigma, b->cx, b->cu, b->csigma, b->endpoint, b->deriv, b->xprop,
                    b->nu, b->rk, b->cost, b->ic, b->info, b->out, b->work, b->iter, b->status, b->active, b->live, b->ttr, b->deriv_f, b->acc, b->d_nlive, b->k1skip};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    delete b;
}

int scvx_batch_set_solver(scvx_batch* b, const scvx_solver_opts* o) {
    if (!b || !o) return SCVX_ERR_ARG;
    if (o->max_iter < 1 || o->refine < 0 || !(o->tol > 0) || !(o->accept_tol >= o->tol) || o->retries < 0 || o->retries > 7)
        return fail(b->ctx, SCVX_ERR_ARG, "bad solver options (max_iter >= 1, refine >= 0, 0 < tol <= accept_tol, 0 <= retries <= 7)");
    // a caller built against the struct before `retries` / `reserved0` were added passes whatever follows its own fields here
    if (o->reserved0 != 0)
        return fail(b->ctx, SCVX_ERR_ARG, "scvx_solver_opts.reserved0 must be 0 (start from scvx_solver_opts_default; an older struct layout is not accepted)");
    b->opts = *o;
    b->C.max_iter = o->max_iter;
    b->C.refine = o->refine;
    b->C.tol = o->tol;
    b->C.accept = o->accept_tol;
    b->C.pad = o->reuse_inactive_tr ? 1 : 0;
    b->C.warm = o->warm_start ? 1 : 0;
    b->C.retries = o->retries;
    return SCVX_OK;
}

int scvx_batch_init(scvx_batch* b, const double* ic) {
    int rc = check_batch(b, false);
    if (rc) return rc;
    scvx_ctx* ctx = b->ctx;
    const scvx_problem& p = ctx->prob;
    const int K = b->K, B = b->B, nrec = b->nrec;
    std::vector<double> rec((size_t)B * nrec), hic((size_t)B * 6), hrk(B, 100.0), hcost(B, INFINITY);
    std::vector<int> hiter(B, 0), hstat(B, SCVX_ST_RUNNING), hact(B, 1);
    for (int t = 0; t < B; t++) {
        double* c = &hic[(size_t)t * 6];
        for (int i = 0; i < 3; i++) { c[i] = ic ? ic[(size_t)t * 6 + i] : p.rIi[i]; c[3 + i] = ic ? ic[(size_t)t * 6 + 3 + i] : p.vIi[i]; }
        double* r = &rec[(size_t)t * nrec];
        double* X = r;
        double* U = r + (size_t)(K + 1) * 14;
        // FirstRound.linear_points, initial_solve.jl:113-129
        for (int k = 0; k <= K; k++) {
            const double a = (double)(K - k) / K, bb = (double)k / K;
            double* xk = X + 14 * k;
            const double mk = a * p.mwet + bb * p.mdry;
            xk[0] = mk;
            double nv[3];
            for (int i = 0; i < 3; i++) {
                xk[1 + i] = a * c[i] + bb * p.rIf[i];
                xk[4 + i] = a * c[3 + i] + bb * p.vIf[i];
                nv[i] = -xk[4 + i];
                xk[11 + i] = 0.0;
            }
            rotation_between_e1(nv, xk + 7);
            for (int c = 0; c < b->NU; c++) U[b->NU * k + c] = 0.0;   // fin controls (if any) start at zero
            U[b->NU * k] = mk * p.g;
        }
        r[nrec - 1] = p.tf_guess;
    }
    hipStream_t st = ctx->stream;
    SCVX_HIP(ctx, hipMemcpyAsync(b->traj, rec.data(), rec.size() * 8, hipMemcpyHostToDevice, st));
    SCVX_HIP(ctx, hipMemcpyAsync(b->traj0, b->traj, rec.size() * 8, hipMemcpyDeviceToDevice, st));
    SCVX_HIP(ctx, hipMemcpyAsync(b->ic, hic.data(), hic.size() * 8, hipMemcpyHostToDevice, st));
    SCVX_HIP(ctx, hipMemcpyAsync(b->rk, hrk.data(), (size_t)B * 8, hipMemcpyHostToDevice, st));
    SCVX_HIP(ctx, hipMemcpyAsync(b->cost, hcost.data(), (size_t)B * 8, hipMemcpyHostToDevice, st));
    SCVX_HIP(ctx, hipMemcpyAsync(b->iter, hiter.data(), (size_t)B * 4, hipMemcpyHostToDevice, st));
    SCVX_HIP(ctx, hipMemcpyAsync(b->status, hstat.data(), (size_t)B * 4, hipMemcpyHostToDevice, st));
    SCVX_HIP(ctx, hipMemcpyAsync(b->active, hact.data(), (size_t)B * 4, hipMemcpyHostToDevice, st));
    SCVX_HIP(ctx, hipMemcpyAsync(b->live, hact.data(), (size_t)B * 4, hipMemcpyHostToDevice, st));
    SCVX_HIP(ctx, hipMemsetAsync(b->out, 0, (size_t)B * 16, st));
    SCVX_HIP(ctx, hipMemsetAsync(b->info, 0, (size_t)B * 32, st));
    SCVX_HIP(ctx, hipMemsetAsync(b->ttr, 0x7f, (size_t)B * 8, st));   // a huge finite value: nothing to reuse yet
    rc = split_views(b, b->traj, b->x, b->u, b->sigma);
    if (rc) return rc;
    SCVX_HIP(ctx, relinearize(b, nullptr));
    SCVX_HIP(ctx, hipStreamSynchronize(st));  // host staging buffers go out of scope
    b->initialised = true;
    b->nactive_host = -1;
    return SCVX_OK;
}

int scvx_batch_init_threedof(scvx_batch* b, const double* ic, const scvx_threedof_opts* opts, int32_t* status3) {
    int rc = scvx_batch_init(b, ic);
    if (rc) return rc;
    scvx_ctx* ctx = b->ctx;
    const int B = b->B, K = b->K;
    hipStream_t st = ctx->stream;
    const size_t no = (size_t)scvx_threedof_record_doubles(K);
    double *d_sol = nullptr, *d_info = nullptr;
    hipError_t e = hipMalloc((void**)&d_sol, (size_t)B * no * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&d_info, (size_t)B * 6 * 8);
    std::vector<double> hinfo((size_t)B * 6);
    if (e == hipSuccess) {
        rc = scvx::threedof_solve_dev(ctx, B, b->ic, opts, d_sol, d_info);
        if (rc == SCVX_OK) rc = scvx::threedof_to_record(ctx, B, K, d_sol, d_info, b->traj, opts ? opts->attitude : 0);
        if (rc == SCVX_OK) {
            e = hipMemcpyAsync(b->traj0, b->traj, (size_t)B * b->nrec * 8, hipMemcpyDeviceToDevice, st);
            if (e == hipSuccess) e = hipMemcpyAsync(hinfo.data(), d_info, hinfo.size() * 8, hipMemcpyDeviceToHost, st);
        }
        if (rc == SCVX_OK && e == hipSuccess) rc = split_views(b, b->traj, b->x, b->u, b->sigma);
        if (rc == SCVX_OK && e == hipSuccess)
            e = relinearize(b, nullptr);
    }
    const hipError_t es = hipStreamSynchronize(st);
    if (d_sol) (void)hipFree(d_sol);
    if (d_info) (void)hipFree(d_info);
    if (rc) return rc;
    if (e != hipSuccess || es != hipSuccess)
        return fail(ctx, SCVX_ERR_HIP, std::string("scvx_batch_init_threedof: ") + hipGetErrorString(e != hipSuccess ? e : es));
    if (status3) for (int t = 0; t < B; t++) status3[t] = (int32_t)hinfo[(size_t)t * 6];
    return SCVX_OK;
}

int scvx_batch_reset(scvx_batch* b) {
    int rc = check_batch(b, true);
    if (rc) return rc;
    scvx_ctx* ctx = b->ctx;
    hipStream_t st = ctx->stream;
    SCVX_HIP(ctx, hipMemcpyAsync(b->traj, b->traj0, (size_t)b->B * b->nrec * 8, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(scvx::reset_scalars_kernel, dim3((unsigned)((b->B + 255) / 256)), dim3(256), 0, st, b->B, b->rk, b->cost,
                       b->iter, b->status, b->active, b->live, b->out, b->info);
    SCVX_HIP(ctx, hipGetLastError());
    rc = split_views(b, b->traj, b->x, b->u, b->sigma);
    if (rc) return rc;
    SCVX_HIP(ctx, relinearize(b, nullptr));
    b->nactive_host = -1;
    return SCVX_OK;
}

int scvx_solve_step_async(scvx_batch* b) {
    int rc = check_batch(b, true);
    if (rc) return rc;
    b->nlive_hint = b->nactive_host >= 0 && b->nactive_host < b->B ? b->nactive_host : -1;
    rc = enqueue_step(b, b->active);
    b->nlive_hint = -1;
    return rc;
}

static int read_step_outputs(scvx_batch* b, int32_t* status, double* nu_norm, double* dJ) {
    scvx_ctx* ctx = b->ctx;
    const int B = b->B;
    std::vector<double> out((size_t)B * 2);
    SCVX_HIP(ctx, hipMemcpyAsync(out.data(), b->out, out.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (status) SCVX_HIP(ctx, hipMemcpyAsync(status, b->status, (size_t)B * 4, hipMemcpyDeviceToHost, ctx->stream));
    SCVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int t = 0; t < B; t++) {
        if (nu_norm) nu_norm[t] = out[2 * t];
        if (dJ) dJ[t] = out[2 * t + 1];
    }
    return SCVX_OK;
}

int scvx_solve_step(scvx_batch* b, int32_t* status, double* nu_norm, double* dJ) {
    int rc = scvx_solve_step_async(b);
    if (rc) return rc;
    return read_step_outputs(b, status, nu_norm, dJ);
}

int scvx_solve(scvx_batch* b, int32_t* status, int32_t* iters, double* nu_norm, double* dJ) {
    int rc = check_batch(b, true);
    if (rc) return rc;
    scvx_ctx* ctx = b->ctx;
    const int B = b->B;
    // Rocketland.solve_problem (rocketland.jl:432-443): cnu = cdel = Inf, iter = 1, loop while not converged and
    // iter < imax.  Per trajectory: `live` starts as `active`, tr_update clears it on convergence or failure, and only
    // live trajectories are stepped.
    SCVX_HIP(ctx, hipMemsetAsync(b->d_nlive, 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL(scvx::copy_flags_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, ctx->stream, B, b->active, b->live, b->d_nlive);
    SCVX_HIP(ctx, hipGetLastError());
    // The count of live trajectories is kept ON THE DEVICE (tr_update decrements it) and mirrored into pinned host memory by an
    // asynchronous 4-byte copy after every step.  Before step `it` is enqueued the host waits only for the copy made after step
    // it - 2: step it - 1 is still queued or running, so the device never idles on the host, and the host never runs more than
    // two steps ahead of a count it has not seen (the count picks the conic solver's executor for the tail and ends the loop
    // when nothing is live).  There is no read-back of the per-trajectory flags inside the loop.
    b->nlive_hint = -1;
    bool pending[2] = {false, false};
    for (int it = 1; it < ctx->prob.imax; it++) {
        const int q = it & 1;
        if (pending[q]) {
            SCVX_HIP(ctx, hipEventSynchronize(b->ev_nlive[q]));
            pending[q] = false;
            const int nlive = b->h_nlive[q];
            if (nlive == 0) break;
            b->nlive_hint = nlive;
        }
        rc = enqueue_step(b, b->live);
        if (rc) { b->nlive_hint = -1; return rc; }
        SCVX_HIP(ctx, hipMemcpyAsync(&b->h_nlive[q], b->d_nlive, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        SCVX_HIP(ctx, hipEventRecord(b->ev_nlive[q], ctx->stream));
        pending[q] = true;
    }
    b->nlive_hint = -1;
    if (iters) {
        SCVX_HIP(ctx, hipMemcpyAsync(iters, b->iter, (size_t)B * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    return read_step_outputs(b, status, nu_norm, dJ);
}

int scvx_batch_get_trajectory(scvx_batch* b, double* traj) {
    int rc = check_batch(b, true);
    if (rc) return rc;
    if (!traj) return fail(b->ctx, SCVX_ERR_ARG, "null buffer");
    SCVX_HIP(b->ctx, hipMemcpyAsync(traj, b->traj, (size_t)b->B * b->nrec * 8, hipMemcpyDeviceToHost, b->ctx->stream));
    SCVX_HIP(b->ctx, hipStreamSynchronize(b->ctx->stream));
    return SCVX_OK;
}

int scvx_batch_set_trajectory(scvx_batch* b, const double* traj) {
    int rc = check_batch(b, true);
    if (rc) return rc;
    if (!traj) return fail(b->ctx, SCVX_ERR_ARG, "null buffer");
    scvx_ctx* ctx = b->ctx;
    SCVX_HIP(ctx, hipMemcpyAsync(b->traj, traj, (size_t)b->B * b->nrec * 8, hipMemcpyHostToDevice, ctx->stream));
    SCVX_HIP(ctx, hipMemsetAsync(b->ttr, 0x7f, (size_t)b->B * 8, ctx->stream));   // a new iterate: no optimum to reuse
    rc = split_views(b, b->traj, b->x, b->u, b->sigma);
    if (rc) return rc;
    SCVX_HIP(ctx, relinearize(b, nullptr));
    SCVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SCVX_OK;
}

int scvx_batch_trajectory_dev(scvx_batch* b, double** traj_dev, int64_t* n_doubles) {
    if (!b || !traj_dev || !n_doubles) return SCVX_ERR_ARG;
    *traj_dev = b->traj;
    *n_doubles = (int64_t)b->B * b->nrec;
    return SCVX_OK;
}

int scvx_batch_get_linearization(scvx_batch* b, double* endpoint, double* deriv) {
    int rc = check_batch(b, true);
    if (rc) return rc;
    scvx_ctx* ctx = b->ctx;
    if (endpoint) SCVX_HIP(ctx, hipMemcpyAsync(endpoint, b->endpoint, (size_t)b->B * b->K * 14 * 8, hipMemcpyDeviceToHost, ctx->stream));
    const size_t nd = (size_t)b->B * b->K * b->dsz;
    if (deriv && b->deriv_f) {   // float tiles: widened on the host (the values the conic solve reads)
        std::vector<float> tmp(nd);
        SCVX_HIP(ctx, hipMemcpyAsync(tmp.data(), b->deriv_f, nd * 4, hipMemcpyDeviceToHost, ctx->stream));
        SCVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (size_t i = 0; i < nd; i++) deriv[i] = (double)tmp[i];
    } else if (deriv) {
        SCVX_HIP(ctx, hipMemcpyAsync(deriv, b->deriv, nd * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    SCVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SCVX_OK;
}

// Keep the derivative tiles [B][K][21][14] in float: the discretisation still integrates in double and rounds each entry
// once, at the store; the conic solve widens on load and does all of its arithmetic, its workspace and its pivots in double.
// Halves the bytes of the one input the solve re-reads in every pass (E, E', the factorisation).  The endpoint stays double.
// Switching re-linearises an initialised batch at once, so the buffer the next solve reads is always current.
int scvx_batch_set_linearization_f32(scvx_batch* b, int on) {
    if (!b || !b->ctx) return SCVX_ERR_ARG;
    scvx_ctx* ctx = b->ctx;
    int rc = SCVX_OK;
    SCVX_HIP(ctx, hipSetDevice(ctx->device));
    if ((on != 0) == (b->deriv_f != nullptr)) return SCVX_OK;
    SCVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const size_t nd = (size_t)b->B * b->K * b->dsz;
    if (on) {
        if ((rc = dmalloc(ctx, &b->deriv_f, nd))) return rc;
        (void)hipFree(b->deriv);
        b->deriv = nullptr;
    } else {
        if ((rc = dmalloc(ctx, &b->deriv, nd))) return rc;
        (void)hipFree(b->deriv_f);
        b->deriv_f = nullptr;
    }
    if (b->initialised) {
        SCVX_HIP(ctx, relinearize(b, nullptr));
        SCVX_HIP(ctx, hipMemsetAsync(b->ttr, 0x7f, (size_t)b->B * 8, ctx->stream));   // the data changed: no optimum to reuse, no warm start
        SCVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return SCVX_OK;
}

int scvx_batch_get_scalars(scvx_batch* b, double* rk, double* cost, int32_t* iter) {
    int rc = check_batch(b, true);
    if (rc) return rc;
    scvx_ctx* ctx = b->ctx;
    if (rk) SCVX_HIP(ctx, hipMemcpyAsync(rk, b->rk, (size_t)b->B * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (cost) SCVX_HIP(ctx, hipMemcpyAsync(cost, b->cost, (size_t)b->B * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (iter) SCVX_HIP(ctx, hipMemcpyAsync(iter, b->iter, (size_t)b->B * 4, hipMemcpyDeviceToHost, ctx->stream));
    SCVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SCVX_OK;
}

int scvx_batch_set_scalars(scvx_batch* b, const double* rk, const double* cost, const int32_t* iter) {
    int rc = check_batch(b, true);
    if (rc) return rc;
    scvx_ctx* ctx = b->ctx;
    if (rk) SCVX_HIP(ctx, hipMemcpyAsync(b->rk, rk, (size_t)b->B * 8, hipMemcpyHostToDevice, ctx->stream));
    if (cost) SCVX_HIP(ctx, hipMemcpyAsync(b->cost, cost, (size_t)b->B * 8, hipMemcpyHostToDevice, ctx->stream));
    if (iter) SCVX_HIP(ctx, hipMemcpyAsync(b->iter, iter, (size_t)b->B * 4, hipMemcpyHostToDevice, ctx->stream));
    // an edited / restored batch solves its next subproblem cold: the kept optimum (reuse_inactive_tr) and the warm-start
    // iterate in the work slab belong to the state before the edit (ttr >= 1e300 disables both)
    SCVX_HIP(ctx, hipMemsetAsync(b->ttr, 0x7f, (size_t)b->B * 8, ctx->stream));
    SCVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SCVX_OK;
}

int scvx_batch_get_solver_stats(scvx_batch* b, int32_t* status, int32_t* iters, double* merit, double* pobj) {
    int rc = check_batch(b, true);
    if (rc) return rc;
    scvx_ctx* ctx = b->ctx;
    std::vector<double> info((size_t)b->B * 4);
    SCVX_HIP(ctx, hipMemcpyAsync(info.data(), b->info, info.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    SCVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int t = 0; t < b->B; t++) {
        if (status) status[t] = (int32_t)info[4 * t];
        if (iters) iters[t] = (int32_t)info[4 * t + 1];
        if (merit) merit[t] = info[4 * t + 2];
        if (pobj) pobj[t] = info[4 * t + 3];
    }
    return SCVX_OK;
}

// Diagnostic export (not part of include/scvx.h): the compile-time switches of the conic kernel this library was built with, so that a
// measurement can name the build it was taken on -- the hash of the sources alone does not see a -D on the command line.
int scvx_debug_build_switches(char* out, int n) {
    return std::snprintf(out, (size_t)(n > 0 ? n : 0),
                         "CHOL_DPP=%d CHOL_ORDER=%d RESID_UPDATE=%d RESID_FRESH_FROM=%g TWISTED_TSPACE=%d FUSED_RES=%d CARRY_BIGSUMS=%d REFINE_FUSED=%d "
                         "K4_OCC=%d K4_BLOCK_OCC=%d K4_PIPELINE=%d K4_TWISTED=%d STREAM_U=%d CHAIN_R=%d FACTOR_T=%d PROF=%d",
                         SCVX_CHOL_DPP, SCVX_CHOL_ORDER, SCVX_RESID_UPDATE, (double)SCVX_RESID_FRESH_FROM, SCVX_TWISTED_TSPACE, SCVX_FUSED_RES,
                         SCVX_CARRY_BIGSUMS, SCVX_REFINE_FUSED, SCVX_K4_OCC, SCVX_K4_BLOCK_OCC, SCVX_K4_PIPELINE, SCVX_K4_TWISTED, SCVX_STREAM_U,
                         SCVX_CHAIN_R, (int)sizeof(SCVX_FACTOR_T),
#if defined(SCVX_IPM_PROF)
                         1
#else
                         0
#endif
                         );
}

#if defined(SCVX_IPM_PROF)
int scvx_debug_ipm_prof(scvx_batch* b, double* out32) {
    (void)hipStreamSynchronize(b->ctx->stream);
    return hipMemcpy(out32, b->work, 128 * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;   // out32: 128 doubles (wavefronts 0..3)
}
#endif

int scvx_batch_get_step_stats(scvx_batch* b, double* out8, int reset) {
    int rc = check_batch(b, false);
    if (rc) return rc;
    if (!out8) return fail(b->ctx, SCVX_ERR_ARG, "null buffer");
    scvx_ctx* ctx = b->ctx;
    static_assert(scvx::ACC_N == 8, "scvx_batch_get_step_stats documents eight totals");
    SCVX_HIP(ctx, hipMemcpyAsync(out8, b->acc, sizeof(double) * scvx::ACC_N, hipMemcpyDeviceToHost, ctx->stream));
    if (reset) SCVX_HIP(ctx, hipMemsetAsync(b->acc, 0, sizeof(double) * scvx::ACC_N, ctx->stream));
    SCVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SCVX_OK;
}

int scvx_batch_set_profiling(scvx_batch* b, int enable) {
    if (!b) return SCVX_ERR_ARG;
    b->nmarks = 0;
    b->profiling = enable != 0;
    return SCVX_OK;
}

int scvx_batch_get_profile(scvx_batch* b, double* ms, int64_t* steps) {
    int rc = check_batch(b, true);
    if (rc) return rc;
    if (!ms || !steps) return fail(b->ctx, SCVX_ERR_ARG, "null buffer");
    scvx_ctx* ctx = b->ctx;
    SCVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 5; i++) ms[i] = 0.0;
    const size_t n = b->nmarks / 7;
    // marks per step: 0 | socp | 1 | candidate+unpack | 2 | propagate | 3 | tr_update | 4 | unpack | 5 | linearize | 6
    static const int slot[6] = {0, 4, 1, 2, 4, 3};
    for (size_t s = 0; s < n; s++)
        for (int i = 0; i < 6; i++) {
            float t = 0.f;
            SCVX_HIP(ctx, hipEventElapsedTime(&t, b->events[7 * s + i], b->events[7 * s + i + 1]));
            ms[slot[i]] += (double)t;
        }
    *steps = (int64_t)n;
    b->nmarks = 0;   // the events stay in the pool
    return SCVX_OK;
}

int scvx_batch_get_flags(scvx_batch* b, int32_t* status, int32_t* active, int32_t* live) {
    int rc = check_batch(b, true);
    if (rc) return rc;
    scvx_ctx* ctx = b->ctx;
    if (status) SCVX_HIP(ctx, hipMemcpyAsync(status, b->status, (size_t)b->B * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (active) SCVX_HIP(ctx, hipMemcpyAsync(active, b->active, (size_t)b->B * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (live) SCVX_HIP(ctx, hipMemcpyAsync(live, b->live, (size_t)b->B * 4, hipMemcpyDeviceToHost, ctx->stream));
    SCVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SCVX_OK;
}

int scvx_batch_set_flags(scvx_batch* b, const int32_t* status, const int32_t* active, const int32_t* live) {
    int rc = check_batch(b, true);
    if (rc) return rc;
    scvx_ctx* ctx = b->ctx;
    if (status) SCVX_HIP(ctx, hipMemcpyAsync(b->status, status, (size_t)b->B * 4, hipMemcpyHostToDevice, ctx->stream));
    if (active) {
        SCVX_HIP(ctx, hipMemcpyAsync(b->active, active, (size_t)b->B * 4, hipMemcpyHostToDevice, ctx->stream));
        int n = 0;
        for (int t = 0; t < b->B; t++) n += active[t] != 0;
        b->nactive_host = n;   // a mostly masked batch picks the conic solver's few-trajectories executors (as scvx_solve's tail does)
    }
    if (live) SCVX_HIP(ctx, hipMemcpyAsync(b->live, live, (size_t)b->B * 4, hipMemcpyHostToDevice, ctx->stream));
    SCVX_HIP(ctx, hipMemsetAsync(b->ttr, 0x7f, (size_t)b->B * 8, ctx->stream));   // see scvx_batch_set_scalars
    SCVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SCVX_OK;
}

int scvx_allgather_trajectories(scvx_batch* b, double* out_dev) {
    int rc = check_batch(b, true);
    if (rc) return rc;
    return scvx_allgather_f64(b->ctx, b->traj, out_dev, (int64_t)b->B * b->nrec);
}

int scvx_allgather_status(scvx_batch* b, int32_t* status_out_dev, int32_t* iters_out_dev) {
    int rc = check_batch(b, true);
    if (rc) return rc;
    if (status_out_dev && (rc = scvx_allgather_i32(b->ctx, b->status, status_out_dev, b->B))) return rc;
    if (iters_out_dev && (rc = scvx_allgather_i32(b->ctx, b->iter, iters_out_dev, b->B))) return rc;
    return SCVX_OK;
}

int scvx_socp_solve(scvx_batch* b, double* sol, double* nu) {
    int rc = check_batch(b, true);
    if (rc) return rc;
    scvx_ctx* ctx = b->ctx;
    rc = enqueue_socp(b, b->active);
    if (rc) return rc;
    const size_t n = (size_t)b->B * b->nrec;
    hipLaunchKernelGGL(scvx::candidate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, b->B, b->nrec,
                       b->traj, b->sol, b->cand);
    SCVX_HIP(ctx, hipGetLastError());
    if (sol) SCVX_HIP(ctx, hipMemcpyAsync(sol, b->cand, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (nu) SCVX_HIP(ctx, hipMemcpyAsync(nu, b->nu, (size_t)b->B * b->K * 14 * 8, hipMemcpyDeviceToHost, ctx->stream));
    SCVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SCVX_OK;
}

}  // extern "C"
