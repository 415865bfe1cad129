// C ABI of libscvx_hip.so: context management and the discretisation entry points (include/scvx.h).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "scvx_internal.hpp"

using scvx::fail;

namespace {

void invert3(const double* M /*row-major*/, double* Mi) {
    const double a = M[0], b = M[1], c = M[2], d = M[3], e = M[4], f = M[5], g = M[6], h = M[7], i = M[8];
    const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    const double id = 1.0 / det;
    Mi[0] = (e * i - f * h) * id;
    Mi[1] = (c * h - b * i) * id;
    Mi[2] = (b * f - c * e) * id;
    Mi[3] = (f * g - d * i) * id;
    Mi[4] = (a * i - c * g) * id;
    Mi[5] = (c * d - a * f) * id;
    Mi[6] = (d * h - e * g) * id;
    Mi[7] = (b * g - a * h) * id;
    Mi[8] = (a * e - b * d) * id;
}

// ProbInfo(from::DescentProblem)  master.jl:82
void fill_dyn(const scvx_problem& p, scvx::DynParams& d) {
    d.alpha = p.alpha;
    d.g0 = p.g;
    d.sos = p.sos;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) d.J[3 * r + c] = p.jB[3 * c + r];  // column-major -> row-major
    invert3(d.J, d.Jinv);
    for (int i = 0; i < 3; i++) d.rTB[i] = p.rTB[i];
    const double* r = p.rTB;
    const double rx[9] = {0, -r[2], r[1], r[2], 0, -r[0], -r[1], r[0], 0};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += d.Jinv[3 * i + k] * rx[3 * k + j];
            d.JrT[3 * i + j] = s;
        }
    // fin extension (SCVX_MODEL_FINS): torque arm of the fin force, Jinv [rFB]x
    d.fin = (p.model_flags & SCVX_MODEL_FINS) ? 1 : 0;
    {
        const double* f = p.rFB;
        const double fx[9] = {0, -f[2], f[1], f[2], 0, -f[0], -f[1], f[0], 0};
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                double s = 0;
                for (int k = 0; k < 3; k++) s += d.Jinv[3 * i + k] * fx[3 * k + j];
                d.JrF[3 * i + j] = s;
            }
    }
    d.aero = 0;  // becomes 1 only once tables are uploaded (scvx_set_aero_table)
    d.n_aoa = d.n_mach = 0;
    d.aoa0 = d.inv_daoa = d.mach0 = d.inv_dmach = 0.0;
    d.force_scalar = p.force_scalar;
    d.cdrag = d.clift = nullptr;
}

// Cubic(Line(OnGrid())) prefilter along one axis of length n (aerodynamics.jl:19-21): solves
//   c[-1] - 2 c[0] + c[1] = 0,  (c[i-1] + 4 c[i] + c[i+1]) / 6 = f[i],  c[n-2] - 2 c[n-1] + c[n] = 0
// for the n+2 coefficients, by dense Gaussian elimination with partial pivoting (host, one-off).
void prefilter_axis(int n, int count, int stride_in, int step_in, const double* in, int stride_out, int step_out,
                    double* out) {
    const int N = n + 2;
    std::vector<double> M((size_t)N * N, 0.0), piv(N);
    M[0] = 1; M[1] = -2; M[2] = 1;
    for (int i = 1; i <= n; i++) {
        M[(size_t)i * N + i - 1] = 1.0 / 6.0;
        M[(size_t)i * N + i] = 4.0 / 6.0;
        M[(size_t)i * N + i + 1] = 1.0 / 6.0;
    }
    M[(size_t)(N - 1) * N + N - 3] = 1; M[(size_t)(N - 1) * N + N - 2] = -2; M[(size_t)(N - 1) * N + N - 1] = 1;
    // LU with partial pivoting
    std::vector<int> perm(N);
    for (int i = 0; i < N; i++) perm[i] = i;
    for (int k = 0; k < N; k++) {
        int pr = k;
        for (int i = k + 1; i < N; i++)
            if (std::fabs(M[(size_t)i * N + k]) > std::fabs(M[(size_t)pr * N + k])) pr = i;
        if (pr != k) {
            for (int j = 0; j < N; j++) std::swap(M[(size_t)k * N + j], M[(size_t)pr * N + j]);
            std::swap(perm[k], perm[pr]);
        }
        for (int i = k + 1; i < N; i++) {
            const double f = M[(size_t)i * N + k] / M[(size_t)k * N + k];
            if (f == 0.0) continue;
            M[(size_t)i * N + k] = f;
            for (int j = k + 1; j < N; j++) M[(size_t)i * N + j] -= f * M[(size_t)k * N + j];
        }
    }
    std::vector<double> rhs(N), y(N);
    for (int l = 0; l < count; l++) {
        rhs[0] = 0;
        rhs[N - 1] = 0;
        for (int i = 0; i < n; i++) rhs[i + 1] = in[(size_t)l * stride_in + (size_t)i * step_in];
        for (int i = 0; i < N; i++) y[i] = rhs[perm[i]];
        for (int i = 0; i < N; i++)
            for (int j = 0; j < i; j++) y[i] -= M[(size_t)i * N + j] * y[j];
        for (int i = N - 1; i >= 0; i--) {
            for (int j = i + 1; j < N; j++) y[i] -= M[(size_t)i * N + j] * y[j];
            y[i] /= M[(size_t)i * N + i];
        }
        for (int i = 0; i < N; i++) out[(size_t)l * stride_out + (size_t)i * step_out] = y[i];
    }
}

std::vector<double> prefilter_table(const double* tab, int na, int nm) {
    // tab[j*na + i] (aoa fastest) -> coef[(nm+2)][(na+2)]
    std::vector<double> ca((size_t)nm * (na + 2));
    prefilter_axis(na, nm, na, 1, tab, na + 2, 1, ca.data());
    std::vector<double> c((size_t)(nm + 2) * (na + 2));
    prefilter_axis(nm, na + 2, 1, na + 2, ca.data(), 1, na + 2, c.data());
    return c;
}

}  // namespace

extern "C" {

int scvx_ctx_create(const scvx_problem* p, int device, scvx_ctx** out) {
    if (!p || !out) return SCVX_ERR_ARG;
    *out = nullptr;
    if (p->K < 1) return SCVX_ERR_ARG;
    if ((p->model_flags & SCVX_MODEL_FINS) && !(p->finmxf > 0.0)) return SCVX_ERR_ARG;   // the fin cone needs its bound
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0 || device < 0 || device >= ndev) return SCVX_ERR_HIP;
    scvx_ctx* ctx = new (std::nothrow) scvx_ctx();
    if (!ctx) return SCVX_ERR_NOMEM;
    ctx->device = device;
    ctx->prob = *p;
    fill_dyn(*p, ctx->dyn);
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return SCVX_ERR_HIP;
    }
    ctx->stream = ctx->own_stream;
    if (hipDeviceGetAttribute(&ctx->num_cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) ctx->num_cus = 0;
    if (const char* v = std::getenv("SCVX_K1_VARIANT")) ctx->k1_variant = std::atoi(v) ? 1 : 0;
    if (const char* v = std::getenv("SCVX_K1_PERSIST")) ctx->k1_persist = std::atoi(v) ? 1 : 0;
    if (const char* v = std::getenv("SCVX_K1_SG")) ctx->k1_sg = std::atoi(v) ? 1 : 0;
    *out = ctx;
    return SCVX_OK;
}

void scvx_ctx_destroy(scvx_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)scvx_comm_destroy(ctx);
    scvx::td_cache_free(ctx);
    if (ctx->d_cdrag) (void)hipFree(ctx->d_cdrag);
    if (ctx->d_clift) (void)hipFree(ctx->d_clift);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int scvx_abi_version(void) { return SCVX_ABI_VERSION; }

int scvx_abi_struct_sizes(int32_t out[3]) {
    if (!out) return SCVX_ERR_ARG;
    out[0] = (int32_t)sizeof(scvx_problem); out[1] = (int32_t)sizeof(scvx_solver_opts); out[2] = (int32_t)sizeof(scvx_threedof_opts);
    return SCVX_OK;
}

const char* scvx_last_error(const scvx_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int scvx_control_dim(const scvx_ctx* ctx) { return ctx && (ctx->prob.model_flags & SCVX_MODEL_FINS) ? 5 : 3; }

int scvx_set_stream(scvx_ctx* ctx, void* hip_stream) {
    if (!ctx) return SCVX_ERR_ARG;
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return SCVX_OK;
}

int scvx_use_null_stream(scvx_ctx* ctx) {
    if (!ctx) return SCVX_ERR_ARG;
    ctx->stream = nullptr;   // HIP's legacy default stream
    return SCVX_OK;
}

int scvx_get_stream(const scvx_ctx* ctx, void** hip_stream) {
    if (!ctx || !hip_stream) return SCVX_ERR_ARG;
    *hip_stream = (void*)ctx->stream;
    return SCVX_OK;
}

int scvx_synchronize(scvx_ctx* ctx) {
    if (!ctx) return SCVX_ERR_ARG;
    SCVX_HIP(ctx, hipSetDevice(ctx->device));
    SCVX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SCVX_OK;
}

int scvx_set_nsub(scvx_ctx* ctx, int nsub) {
    if (!ctx || nsub < 1 || nsub > 1000) return fail(ctx, SCVX_ERR_ARG, "nsub must be in [1,1000]");
    ctx->nsub = nsub;
    return SCVX_OK;
}

int scvx_get_nsub(const scvx_ctx* ctx) { return ctx ? ctx->nsub : SCVX_ERR_ARG; }

int scvx_set_aero_table(scvx_ctx* ctx, const double* drag, const double* lift, const double* trq, int n_aoa,
                        int n_mach, double aoa0, double daoa, double mach0, double dmach) {
    (void)trq;  // the torque table is loaded by the reference and never reaches the dynamics (dynamics.jl:69)
    if (!ctx || !drag || !lift || n_aoa < 4 || n_mach < 4 || !(daoa > 0) || !(dmach > 0))
        return fail(ctx, SCVX_ERR_ARG, "bad aero table");
    SCVX_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<double> cd = prefilter_table(drag, n_aoa, n_mach);
    std::vector<double> cl = prefilter_table(lift, n_aoa, n_mach);
    const size_t bytes = cd.size() * sizeof(double);
    if (ctx->d_cdrag) (void)hipFree(ctx->d_cdrag);
    if (ctx->d_clift) (void)hipFree(ctx->d_clift);
    ctx->d_cdrag = ctx->d_clift = nullptr;
    SCVX_HIP(ctx, hipMalloc(&ctx->d_cdrag, bytes));
    SCVX_HIP(ctx, hipMalloc(&ctx->d_clift, bytes));
    SCVX_HIP(ctx, hipMemcpy(ctx->d_cdrag, cd.data(), bytes, hipMemcpyHostToDevice));
    SCVX_HIP(ctx, hipMemcpy(ctx->d_clift, cl.data(), bytes, hipMemcpyHostToDevice));
    scvx::DynParams& d = ctx->dyn;
    d.aero = 1;
    d.n_aoa = n_aoa;
    d.n_mach = n_mach;
    d.aoa0 = aoa0;
    d.inv_daoa = 1.0 / daoa;
    d.mach0 = mach0;
    d.inv_dmach = 1.0 / dmach;
    d.force_scalar = ctx->prob.force_scalar;
    d.cdrag = ctx->d_cdrag;
    d.clift = ctx->d_clift;
    return SCVX_OK;
}

static int check_disc(scvx_ctx* ctx, int B, int K, const void* a, const void* b, const void* c, const void* d) {
    if (!ctx) return SCVX_ERR_ARG;
    if (B < 0 || K < 1) return fail(ctx, SCVX_ERR_ARG, "B >= 0 and K >= 1 required");
    if (B > 0 && (!a || !b || !c || !d)) return fail(ctx, SCVX_ERR_ARG, "null buffer");
    if (ctx->prob.aero_kind == 1 && !ctx->dyn.aero)
        return fail(ctx, SCVX_ERR_STATE, "AtmosphericData problem: call scvx_set_aero_table first");
    return SCVX_OK;
}

int scvx_linearize_f64(scvx_ctx* ctx, int B, int K, const double* x_dev, const double* u_dev, const double* sigma_dev,
                       double dt, double* endpoint_dev, double* deriv_dev) {
    int rc = check_disc(ctx, B, K, x_dev, u_dev, sigma_dev, endpoint_dev);
    if (rc) return rc;
    if (B > 0 && !deriv_dev) return fail(ctx, SCVX_ERR_ARG, "null buffer");
    SCVX_HIP(ctx, hipSetDevice(ctx->device));
    SCVX_HIP(ctx, scvx::launch_linearize(ctx, B, K, x_dev, u_dev, sigma_dev, dt, endpoint_dev, deriv_dev, ctx->stream));
    return SCVX_OK;
}

int scvx_propagate_f64(scvx_ctx* ctx, int B, int K, const double* x_dev, const double* u_dev, const double* sigma_dev,
                       double dt, double* xnext_dev) {
    int rc = check_disc(ctx, B, K, x_dev, u_dev, sigma_dev, xnext_dev);
    if (rc) return rc;
    SCVX_HIP(ctx, hipSetDevice(ctx->device));
    SCVX_HIP(ctx, scvx::launch_propagate(ctx, B, K, x_dev, u_dev, sigma_dev, dt, xnext_dev, ctx->stream));
    return SCVX_OK;
}

int scvx_linearize_f32(scvx_ctx* ctx, int B, int K, const float* x_dev, const float* u_dev, const float* sigma_dev, float dt,
                       float* endpoint_dev, float* deriv_dev) {
    int rc = check_disc(ctx, B, K, x_dev, u_dev, sigma_dev, endpoint_dev);
    if (rc) return rc;
    if (B > 0 && !deriv_dev) return fail(ctx, SCVX_ERR_ARG, "null buffer");
    SCVX_HIP(ctx, hipSetDevice(ctx->device));
    SCVX_HIP(ctx, scvx::launch_linearize_f32(ctx, B, K, x_dev, u_dev, sigma_dev, dt, endpoint_dev, deriv_dev, ctx->stream));
    return SCVX_OK;
}

int scvx_propagate_f32(scvx_ctx* ctx, int B, int K, const float* x_dev, const float* u_dev, const float* sigma_dev, float dt,
                       float* xnext_dev) {
    int rc = check_disc(ctx, B, K, x_dev, u_dev, sigma_dev, xnext_dev);
    if (rc) return rc;
    SCVX_HIP(ctx, hipSetDevice(ctx->device));
    SCVX_HIP(ctx, scvx::launch_propagate_f32(ctx, B, K, x_dev, u_dev, sigma_dev, dt, xnext_dev, ctx->stream));
    return SCVX_OK;
}

namespace {
struct DevBufF {
    float* p = nullptr;
    ~DevBufF() {
        if (p) (void)hipFree(p);
    }
};
}  // namespace

static int disc_host_f32(scvx_ctx* ctx, int B, int K, const float* x, const float* u, const float* sigma, float dt,
                         float* endpoint, float* deriv, bool with_deriv) {
    int rc = check_disc(ctx, B, K, x, u, sigma, endpoint);
    if (rc) return rc;
    if (B == 0) return SCVX_OK;
    if (with_deriv && !deriv) return fail(ctx, SCVX_ERR_ARG, "null buffer");
    SCVX_HIP(ctx, hipSetDevice(ctx->device));
    const int NU = scvx_control_dim(ctx);
    const size_t nx = (size_t)B * (K + 1) * 14, nu = (size_t)B * (K + 1) * NU, ne = (size_t)B * K * 14,
                 nd = (size_t)B * K * 14 * (14 + 2 * NU + 1);
    DevBufF dx, du, ds, de, dd;
    SCVX_HIP(ctx, hipMalloc(&dx.p, nx * 4));
    SCVX_HIP(ctx, hipMalloc(&du.p, nu * 4));
    SCVX_HIP(ctx, hipMalloc(&ds.p, (size_t)B * 4));
    SCVX_HIP(ctx, hipMalloc(&de.p, ne * 4));
    if (with_deriv) SCVX_HIP(ctx, hipMalloc(&dd.p, nd * 4));
    hipStream_t st = ctx->stream;
    SCVX_HIP(ctx, hipMemcpyAsync(dx.p, x, nx * 4, hipMemcpyHostToDevice, st));
    SCVX_HIP(ctx, hipMemcpyAsync(du.p, u, nu * 4, hipMemcpyHostToDevice, st));
    SCVX_HIP(ctx, hipMemcpyAsync(ds.p, sigma, (size_t)B * 4, hipMemcpyHostToDevice, st));
    if (with_deriv)
        SCVX_HIP(ctx, scvx::launch_linearize_f32(ctx, B, K, dx.p, du.p, ds.p, dt, de.p, dd.p, st));
    else
        SCVX_HIP(ctx, scvx::launch_propagate_f32(ctx, B, K, dx.p, du.p, ds.p, dt, de.p, st));
    SCVX_HIP(ctx, hipMemcpyAsync(endpoint, de.p, ne * 4, hipMemcpyDeviceToHost, st));
    if (with_deriv) SCVX_HIP(ctx, hipMemcpyAsync(deriv, dd.p, nd * 4, hipMemcpyDeviceToHost, st));
    SCVX_HIP(ctx, hipStreamSynchronize(st));
    return SCVX_OK;
}

int scvx_linearize_f32_host(scvx_ctx* ctx, int B, int K, const float* x, const float* u, const float* sigma, float dt,
                            float* endpoint, float* deriv) {
    return disc_host_f32(ctx, B, K, x, u, sigma, dt, endpoint, deriv, true);
}

int scvx_propagate_f32_host(scvx_ctx* ctx, int B, int K, const float* x, const float* u, const float* sigma, float dt,
                            float* xnext) {
    return disc_host_f32(ctx, B, K, x, u, sigma, dt, xnext, nullptr, false);
}

namespace {
struct DevBuf {
    double* p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
};
}  // namespace

static int disc_host(scvx_ctx* ctx, int B, int K, const double* x, const double* u, const double* sigma, double dt,
                     double* endpoint, double* deriv, bool with_deriv) {
    int rc = check_disc(ctx, B, K, x, u, sigma, endpoint);
    if (rc) return rc;
    if (B == 0) return SCVX_OK;
    if (with_deriv && !deriv) return fail(ctx, SCVX_ERR_ARG, "null buffer");
    SCVX_HIP(ctx, hipSetDevice(ctx->device));
    const int NU = scvx_control_dim(ctx);
    const size_t nx = (size_t)B * (K + 1) * 14, nu = (size_t)B * (K + 1) * NU, ne = (size_t)B * K * 14,
                 nd = (size_t)B * K * 14 * (14 + 2 * NU + 1);
    DevBuf dx, du, ds, de, dd;
    SCVX_HIP(ctx, hipMalloc(&dx.p, nx * 8));
    SCVX_HIP(ctx, hipMalloc(&du.p, nu * 8));
    SCVX_HIP(ctx, hipMalloc(&ds.p, (size_t)B * 8));
    SCVX_HIP(ctx, hipMalloc(&de.p, ne * 8));
    if (with_deriv) SCVX_HIP(ctx, hipMalloc(&dd.p, nd * 8));
    hipStream_t st = ctx->stream;
    SCVX_HIP(ctx, hipMemcpyAsync(dx.p, x, nx * 8, hipMemcpyHostToDevice, st));
    SCVX_HIP(ctx, hipMemcpyAsync(du.p, u, nu * 8, hipMemcpyHostToDevice, st));
    SCVX_HIP(ctx, hipMemcpyAsync(ds.p, sigma, (size_t)B * 8, hipMemcpyHostToDevice, st));
    if (with_deriv)
        SCVX_HIP(ctx, scvx::launch_linearize(ctx, B, K, dx.p, du.p, ds.p, dt, de.p, dd.p, st));
    else
        SCVX_HIP(ctx, scvx::launch_propagate(ctx, B, K, dx.p, du.p, ds.p, dt, de.p, st));
    SCVX_HIP(ctx, hipMemcpyAsync(endpoint, de.p, ne * 8, hipMemcpyDeviceToHost, st));
    if (with_deriv) SCVX_HIP(ctx, hipMemcpyAsync(deriv, dd.p, nd * 8, hipMemcpyDeviceToHost, st));
    SCVX_HIP(ctx, hipStreamSynchronize(st));
    return SCVX_OK;
}

int scvx_linearize_f64_host(scvx_ctx* ctx, int B, int K, const double* x, const double* u, const double* sigma,
                            double dt, double* endpoint, double* deriv) {
    return disc_host(ctx, B, K, x, u, sigma, dt, endpoint, deriv, true);
}

int scvx_propagate_f64_host(scvx_ctx* ctx, int B, int K, const double* x, const double* u, const double* sigma,
                            double dt, double* xnext) {
    return disc_host(ctx, B, K, x, u, sigma, dt, xnext, nullptr, false);
}

}  // extern "C"
