// K0: FirstRound.solve_initial (initial_solve.jl:17-110) -- the 3-DoF lossless-convexification landing SOCP -- for a
// batch of initial conditions, one wavefront per trajectory (algorithm and data layout: scvx_threedof_core.hpp).
//
// Launch shape: persistent single-wavefront workgroups, `grid = min(B, 8 per CU)`, each striding over trajectories, so the
// 253 KB solver slab exists once per workgroup in flight (<= 0.5 GB) instead of once per trajectory.  The constant
// tables of the problem (203 KB at K = 30: equality part in ELL and band form, cone rows, E') are built on the host
// when the problem or the options change, cached in the context and read by every trajectory through L2.
#include <cstring>
#include <new>
#include <vector>
#include "scvx_internal.hpp"
#include "scvx_threedof_core.hpp"

namespace scvx {

struct TdCache {
    td::Problem3 P{};
    scvx_threedof_opts o{};
    td::Tables t{};           // device pointers
    void* blob = nullptr;     // one device allocation behind them
    double* work = nullptr;   // [grid][stride]
    double* prof = nullptr;   // diagnostic builds (SCVX_TD_PROF): section cycles of trajectory 0
    size_t work_doubles = 0;
};

namespace {

struct WaveEx3 {
    td::lptr lds;
    __device__ __forceinline__ int lane() const { return (int)threadIdx.x; }
    __device__ __forceinline__ int nlanes() const { return 64; }
    __device__ __forceinline__ void sync() { __syncthreads(); }
    // LDS-only ordering inside the single wavefront of the block (global traffic stays in flight)
    __device__ __forceinline__ void sync_lds() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    }
    __device__ __forceinline__ double sum(double x) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
        return x;
    }
    __device__ __forceinline__ double min(double x) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x = fmin(x, __shfl_xor(x, o, 64));
        return x;
    }
    __device__ __forceinline__ td::lptr fast() { return lds; }
    static constexpr int kLanes = 64;
    static constexpr bool kRegisterSweep = true;

    // value of x in lane `src` (wave-uniform, a constant after unrolling) in every lane: two v_readlane_b32
    static __device__ __forceinline__ double bcast(double x, int src) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(x), src);
        const int hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
        return __hiloint2double(hi, lo);
    }

    // The two triangular sweeps of xs <- (L D L')^-1 xs with the running part of the vector in REGISTERS: lane l holds
    // x[J + l] (forward; x[J - l] backward) for a chunk of CH columns; step t broadcasts the finished x[J + t] with
    // v_readlane and every lane within the bandwidth does one FMA -- no LDS round trip inside the chunk (the vector
    // visits LDS once per chunk).  The L entries of the NEXT chunk are in flight while this one computes.
    static constexpr int CH = 16;   // CH + BW <= 64
    // The chunk's L entries as they lie in memory, no masking: entry of step t for lane l is m[(J +- t) * BS + (l - t)] -- one base
    // address, constant stride.  L is zero-padded by LPAD on both sides (chunk starts are clamped to [-1, nb]), columns past the
    // matrix hold zeros (the window is fed zeros there) and the never-written corner of the row copy (row r, d > r) is zeroed
    // once per solve, so the only entries that must not be used are the ones outside the band, d = l - t not in [1, BW]: a
    // predicate of (lane, step) alone, which the compiler keeps in scalar registers across the whole sweep (run_chunk).
    template <bool FWD>
    __device__ __forceinline__ void load_chunk(double (&c)[CH], td::cgptr m, int J, int nb) const {
        const int l = (int)threadIdx.x;
        const int Jc = FWD ? (J < nb ? J : nb) : (J > -1 ? J : -1);
        td::cgptr base = m + ((long)Jc * td::BS + l);
#pragma unroll
        for (int t = 0; t < CH; t++) c[t] = base[FWD ? t * (td::BS - 1) : -t * (td::BS + 1)];
    }
    template <bool FWD>
    __device__ __forceinline__ void run_chunk(const double (&c)[CH], td::lptr xs, int J, int nb) {
        if (FWD ? J >= nb : J < 0) return;   // wave-uniform: a prefetched chunk past the end
        const int l = (int)threadIdx.x;
        const int idx = FWD ? J + l : J - l;
        const bool in = idx >= 0 && idx < nb;
        double xw = in ? xs[idx] : 0.0;
#pragma unroll
        for (int t = 0; t < CH; t++) {
            const double b = bcast(xw, t);
            if ((unsigned)(l - t - 1) < (unsigned)td::BW) xw = fma(-c[t], b, xw);   // 1 <= l - t <= BW: inside the band
        }
        if (in) xs[idx] = xw;
        sync_lds();
    }
    // Four chunks in flight: under load an L entry comes from HBM (a slab per wavefront in flight, 0.5 GB in all), a chunk computes
    // in ~0.2 us and a request takes ten times that -- with one chunk of look-ahead the sweeps ran at memory latency.
    __device__ __forceinline__ void band_sweeps(td::lptr xs, td::cgptr lb, td::cgptr ut, int nb) {
        double c0[CH], c1[CH], c2[CH], c3[CH];
        load_chunk<true>(c0, lb, 0, nb);
        load_chunk<true>(c1, lb, CH, nb);
        load_chunk<true>(c2, lb, 2 * CH, nb);
        for (int J = 0; J < nb; J += 4 * CH) {
            load_chunk<true>(c3, lb, J + 3 * CH, nb);
            run_chunk<true>(c0, xs, J, nb);
            load_chunk<true>(c0, lb, J + 4 * CH, nb);
            run_chunk<true>(c1, xs, J + CH, nb);
            load_chunk<true>(c1, lb, J + 5 * CH, nb);
            run_chunk<true>(c2, xs, J + 2 * CH, nb);
            load_chunk<true>(c2, lb, J + 6 * CH, nb);
            run_chunk<true>(c3, xs, J + 3 * CH, nb);
        }
        for (int j = (int)threadIdx.x; j < nb; j += 64) xs[j] *= lb[(size_t)j * td::BS];
        sync_lds();
        load_chunk<false>(c0, ut, nb - 1, nb);
        load_chunk<false>(c1, ut, nb - 1 - CH, nb);
        load_chunk<false>(c2, ut, nb - 1 - 2 * CH, nb);
        for (int J = nb - 1; J >= 0; J -= 4 * CH) {
            load_chunk<false>(c3, ut, J - 3 * CH, nb);
            run_chunk<false>(c0, xs, J, nb);
            load_chunk<false>(c0, ut, J - 4 * CH, nb);
            run_chunk<false>(c1, xs, J - CH, nb);
            load_chunk<false>(c1, ut, J - 5 * CH, nb);
            run_chunk<false>(c2, xs, J - 2 * CH, nb);
            load_chunk<false>(c2, ut, J - 6 * CH, nb);
            run_chunk<false>(c3, xs, J - 3 * CH, nb);
        }
    }
};

// sol [B][(K+1)*15+1]; info [B][6] = status, iters, pobj, gap, pres, dres
#ifndef SCVX_K0_WAVES
#define SCVX_K0_WAVES 2   // wavefronts per SIMD the kernel is compiled for (and the launch is sized for): profiles/r03_k0_occupancy.md
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SCVX_K0_WAVES, SCVX_K0_WAVES))) void threedof_kernel(td::Tables Tin, int B, const double* __restrict__ ic, double* work,
                                                      size_t stride, double* sol, double* info, double* prof_out) {
    extern __shared__ double td_lds[];
    // tables, executor and solver object in LDS, one copy for the wavefront: as automatic objects they sat in private memory (one
    // copy per LANE) and every non-inlined routine opened with loads of that memory (profiles/r04_k4_lds_frame.md, the same change in K4)
    struct Frame {
        td::Tables T; WaveEx3 ex; td::Solver<WaveEx3> S;
        __device__ Frame(const td::Tables& t, td::lptr lds, double* w) : T(t), ex{lds}, S(ex, T, w) {}
    };
    __shared__ __attribute__((aligned(16))) unsigned char frame_mem[(sizeof(Frame) + 15) & ~(size_t)15];
    const int no = td::out_doubles(Tin.N);
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        Frame* const F = new (frame_mem) Frame(Tin, (td::lptr)td_lds, work + (size_t)blockIdx.x * stride);
        td::Solver<WaveEx3>& S = F->S;
        const td::Result r = S.solve(ic + (size_t)b * 6, sol + (size_t)b * no);
        if (threadIdx.x == 0) {
            double* o = info + (size_t)b * 6;
            o[0] = r.status; o[1] = r.iters; o[2] = r.pobj; o[3] = r.gap; o[4] = r.pres; o[5] = r.dres;
        }
        __syncthreads();
#if defined(SCVX_TD_PROF)
        if (b == 0 && threadIdx.x == 0) for (int i = 0; i < 16; i++) prof_out[i] = S.prof[i];
#endif
    }
}

__device__ void rotation_between_e1_dev(const double* b, double* q) {
    // Rotations.rotation_between([1,0,0], b) as [w,x,y,z] (initial_solve.jl:98-99), same branches as scvx_batch_init's
    const double nb = sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
    double w = nb + b[0];
    double v0 = 0.0, v1 = -b[2], v2 = b[1];   // e1 x b
    if (fabs(w) < 100 * 2.220446049250313e-16) { v0 = 0; v1 = 0; v2 = 1; w = 0; }
    const double n = sqrt(w * w + v0 * v0 + v1 * v1 + v2 * v2);
    q[0] = w / n; q[1] = v0 / n; q[2] = v1 / n; q[3] = v2 / n;
}

// initial_solve.jl:90-105: LinPoints from the 3-DoF optimum -- state (ma, r, v, rotation_between(e1, -T), 0), control
// (|T|, 0, 0) -- written over the trajectory record of every trajectory whose 3-DoF solve is optimal
__global__ void threedof_to_record_kernel(int B, int K, int NU, const double* __restrict__ sol, const double* __restrict__ info,
                                          double sigma, double tsign, double* __restrict__ rec) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * (K + 1)) return;
    const int b = i / (K + 1), k = i - b * (K + 1);
    if (info[(size_t)b * 6] != 0.0) return;
    const int nrec = (K + 1) * (14 + NU) + 1;
    const double* z = sol + (size_t)b * td::out_doubles(K) + (size_t)k * td::NV;
    double* x = rec + (size_t)b * nrec + 14 * k;
    double* u = rec + (size_t)b * nrec + (size_t)(K + 1) * 14 + NU * k;
    for (int c = 3; c < NU; c++) u[c] = 0.0;   // fin controls start at zero
    x[0] = z[6];
    for (int j = 0; j < 3; j++) { x[1 + j] = z[j]; x[4 + j] = z[3 + j]; x[11 + j] = 0.0; }
    const double nT[3] = {tsign * z[7], tsign * z[8], tsign * z[9]};
    rotation_between_e1_dev(nT, x + 7);
    u[0] = sqrt(z[7] * z[7] + z[8] * z[8] + z[9] * z[9]); u[1] = 0.0; u[2] = 0.0;
    if (k == 0) rec[(size_t)b * nrec + nrec - 1] = sigma;
}

bool same_setup(const TdCache* c, const td::Problem3& P, const scvx_threedof_opts& o) {
    // the tables depend on the problem and on the solver options, not on the attitude convention of the 6-DoF start
    return c && c->blob && std::memcmp(&c->P, &P, sizeof P) == 0 && c->o.max_iter == o.max_iter && c->o.refine == o.refine &&
           c->o.tol == o.tol && c->o.delta == o.delta;
}

}  // namespace

void td_cache_free(scvx_ctx* ctx) {
    if (!ctx || !ctx->td) return;
    if (ctx->td->blob) (void)hipFree(ctx->td->blob);
    if (ctx->td->work) (void)hipFree(ctx->td->work);
    if (ctx->td->prof) (void)hipFree(ctx->td->prof);
    delete ctx->td;
    ctx->td = nullptr;
}

// tables of ctx->prob with options o on the device (cached)
static int td_setup(scvx_ctx* ctx, const scvx_threedof_opts& o) {
    const scvx_problem& p = ctx->prob;
    td::Problem3 P{};
    P.K = p.K; P.alpha = p.alpha; P.tf_guess = p.tf_guess; P.mwet = p.mwet; P.mdry = p.mdry; P.g = p.g;
    P.Tmin = p.Tmin; P.Tmax = p.Tmax; P.thetaMax = p.thetaMax; P.gammaGs = p.gammaGs;
    if (same_setup(ctx->td, P, o)) return SCVX_OK;
    if (!(o.tol > 0.0) || o.max_iter < 1 || o.refine < 0 || !(o.delta > 0.0) || o.attitude < 0 || o.attitude > 1)
        return fail(ctx, SCVX_ERR_ARG, "threedof options: tol > 0, max_iter >= 1, refine >= 0, delta > 0, attitude 0 or 1 required");
    td::HostTables H;
    if (const char* e = td::build_tables(P, o.tol, o.max_iter, o.refine, o.delta, H)) return fail(ctx, SCVX_ERR_ARG, e);
    if (td::fast_doubles(P.K) * 8 > 64 * 1024) return fail(ctx, SCVX_ERR_ARG, "K too large for the 3-DoF solver's LDS window");
    td_cache_free(ctx);
    TdCache* c = new (std::nothrow) TdCache();
    if (!c) return SCVX_ERR_NOMEM;
    ctx->td = c;
    c->P = P; c->o = o;
    // one blob: the double tables, then the int tables
    const std::vector<double>* dv[] = {&H.a_val, &H.kc, &H.e_c0, &H.e_c1, &H.e_h, &H.t_coef, &H.q};
    const std::vector<int>* iv[] = {&H.a_col, &H.e_v0, &H.e_v1, &H.t_row};
    size_t nd = 0, ni = 0;
    for (auto v : dv) nd += (v->size() + 7) & ~(size_t)7;
    for (auto v : iv) ni += (v->size() + 7) & ~(size_t)7;
    std::vector<char> host(nd * 8 + ni * 4);
    SCVX_HIP(ctx, hipMalloc(&c->blob, host.size()));
    const double* dptr[7];
    const int* iptr[4];
    size_t off = 0;
    for (int k = 0; k < 7; k++) {
        std::memcpy(host.data() + off, dv[k]->data(), dv[k]->size() * 8);
        dptr[k] = (const double*)((char*)c->blob + off);
        off += ((dv[k]->size() + 7) & ~(size_t)7) * 8;
    }
    for (int k = 0; k < 4; k++) {
        std::memcpy(host.data() + off, iv[k]->data(), iv[k]->size() * 4);
        iptr[k] = (const int*)((char*)c->blob + off);
        off += ((iv[k]->size() + 7) & ~(size_t)7) * 4;
    }
    SCVX_HIP(ctx, hipMemcpy(c->blob, host.data(), host.size(), hipMemcpyHostToDevice));
    c->t = H.t;
    c->t.a_val = dptr[0]; c->t.kc = dptr[1]; c->t.e_c0 = dptr[2]; c->t.e_c1 = dptr[3]; c->t.e_h = dptr[4];
    c->t.t_coef = dptr[5]; c->t.q = dptr[6];
    c->t.a_col = iptr[0]; c->t.e_v0 = iptr[1]; c->t.e_v1 = iptr[2]; c->t.t_row = iptr[3];
    return SCVX_OK;
}

int threedof_solve_dev(scvx_ctx* ctx, int B, const double* ic_dev, const scvx_threedof_opts* opts, double* sol_dev,
                       double* info_dev) {
    scvx_threedof_opts o;
    if (opts) o = *opts; else scvx_threedof_default_opts(&o);
    int rc = td_setup(ctx, o);
    if (rc) return rc;
    TdCache* c = ctx->td;
    const int cap = (ctx->num_cus > 0 ? ctx->num_cus : 256) * 4 * SCVX_K0_WAVES;   // SCVX_K0_WAVES wavefronts per SIMD
    const int grid = B < cap ? B : cap;
    td::Layout L;
    L.init(c->P.K);
    const size_t need = (size_t)grid * L.total;
    if (c->work_doubles < need) {
        SCVX_HIP(ctx, hipStreamSynchronize(ctx->stream));   // an earlier launch may still be reading the old slab
        if (c->work) (void)hipFree(c->work);
        c->work = nullptr; c->work_doubles = 0;
        SCVX_HIP(ctx, hipMalloc((void**)&c->work, need * 8));
        c->work_doubles = need;
    }
    const size_t lds = td::fast_doubles(c->P.K) * 8;
#if defined(SCVX_TD_PROF)
    if (!c->prof) SCVX_HIP(ctx, hipMalloc((void**)&c->prof, 16 * 8));
#endif
    hipLaunchKernelGGL(threedof_kernel, dim3(grid), dim3(64), lds, ctx->stream, c->t, B, ic_dev, c->work, L.total, sol_dev,
                       info_dev, c->prof);
    SCVX_HIP(ctx, hipGetLastError());
    return SCVX_OK;
}

int threedof_to_record(scvx_ctx* ctx, int B, int K, const double* sol_dev, const double* info_dev, double* rec_dev, int attitude) {
    const int n = B * (K + 1);
    hipLaunchKernelGGL(threedof_to_record_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, B, K, scvx_control_dim(ctx), sol_dev, info_dev,
                       ctx->prob.tf_guess, attitude == 1 ? 1.0 : -1.0, rec_dev);
    SCVX_HIP(ctx, hipGetLastError());
    return SCVX_OK;
}

}  // namespace scvx

extern "C" {

#if defined(SCVX_TD_PROF)
int scvx_debug_td_prof(scvx_ctx* ctx, double* out16) {
    if (!ctx || !ctx->td || !ctx->td->prof) return SCVX_ERR_STATE;
    (void)hipStreamSynchronize(ctx->stream);
    return hipMemcpy(out16, ctx->td->prof, 16 * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : SCVX_ERR_HIP;
}
#endif

int scvx_threedof_default_opts(scvx_threedof_opts* o) {
    if (!o) return SCVX_ERR_ARG;
    o->max_iter = 60;
    o->refine = 1;
    o->tol = 1e-9;
    o->delta = 1e-9;
    o->attitude = 0;
    o->reserved = 0;
    return SCVX_OK;
}

int32_t scvx_threedof_record_doubles(int K) { return K < 1 ? 0 : scvx::td::out_doubles(K); }

int scvx_threedof_solve_dev(scvx_ctx* ctx, int B, const double* ic_dev, const scvx_threedof_opts* opts, double* sol_dev,
                            double* info_dev) {
    if (!ctx || !ic_dev || !sol_dev || !info_dev) return SCVX_ERR_ARG;
    if (B < 1) return scvx::fail(ctx, SCVX_ERR_ARG, "B >= 1 required");
    SCVX_HIP(ctx, hipSetDevice(ctx->device));
    return scvx::threedof_solve_dev(ctx, B, ic_dev, opts, sol_dev, info_dev);
}

int scvx_threedof_solve(scvx_ctx* ctx, int B, const double* ic, const scvx_threedof_opts* opts, double* sol, int32_t* status,
                        double* info) {
    if (!ctx || !sol) return SCVX_ERR_ARG;
    if (B < 1) return scvx::fail(ctx, SCVX_ERR_ARG, "B >= 1 required");
    SCVX_HIP(ctx, hipSetDevice(ctx->device));
    const scvx_problem& p = ctx->prob;
    std::vector<double> hic((size_t)B * 6);
    for (int t = 0; t < B; t++)
        for (int i = 0; i < 3; i++) {
            hic[(size_t)t * 6 + i] = ic ? ic[(size_t)t * 6 + i] : p.rIi[i];
            hic[(size_t)t * 6 + 3 + i] = ic ? ic[(size_t)t * 6 + 3 + i] : p.vIi[i];
        }
    const size_t no = (size_t)scvx::td::out_doubles(p.K);
    double *d_ic = nullptr, *d_sol = nullptr, *d_info = nullptr;
    int rc = SCVX_OK;
    auto release = [&]() {
        if (d_ic) (void)hipFree(d_ic);
        if (d_sol) (void)hipFree(d_sol);
        if (d_info) (void)hipFree(d_info);
    };
    hipError_t e = hipMalloc((void**)&d_ic, hic.size() * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&d_sol, (size_t)B * no * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&d_info, (size_t)B * 6 * 8);
    if (e == hipSuccess) e = hipMemcpyAsync(d_ic, hic.data(), hic.size() * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) { release(); return scvx::fail(ctx, SCVX_ERR_HIP, std::string("scvx_threedof_solve: ") + hipGetErrorString(e)); }
    rc = scvx::threedof_solve_dev(ctx, B, d_ic, opts, d_sol, d_info);
    std::vector<double> hinfo((size_t)B * 6);
    if (rc == SCVX_OK) {
        e = hipMemcpyAsync(sol, d_sol, (size_t)B * no * 8, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(hinfo.data(), d_info, hinfo.size() * 8, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = scvx::fail(ctx, SCVX_ERR_HIP, std::string("scvx_threedof_solve: ") + hipGetErrorString(e));
    } else {
        (void)hipStreamSynchronize(ctx->stream);
    }
    release();
    if (rc) return rc;
    for (int t = 0; t < B; t++) {
        if (status) status[t] = (int32_t)hinfo[(size_t)t * 6];
        if (info) for (int q = 0; q < 5; q++) info[(size_t)t * 5 + q] = hinfo[(size_t)t * 6 + 1 + q];
    }
    return SCVX_OK;
}

}  // extern "C"
