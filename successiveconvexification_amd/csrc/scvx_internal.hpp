// Internal declarations shared by the translation units of libscvx_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include "scvx.h"
#include "scvx_dyn.hpp"

namespace scvx { struct TdCache; }   // device tables + workspace of the 3-DoF initialiser (scvx_threedof.hip)

struct scvx_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    scvx_problem prob{};
    scvx::DynParams dyn{};
    int nsub = 10;
    int k1_persist = -1; // persistent blocks in K1's producer/consumer kernel: -1 auto (by npts), 0 / 1 forced (SCVX_K1_PERSIST)
    int num_cus = 0;     // compute units of the device (persistent-block launch shapes)
    int k1_variant = 1;  // 0: one-lane-per-column kernel, 1: producer/consumer kernel (SCVX_K1_VARIANT overrides)
    int k1_sg = 1;       // producer/consumer pipeline per RK stage (1, default) or per substep (0); SCVX_K1_SG overrides
    double* d_cdrag = nullptr;
    double* d_clift = nullptr;
    void* comm = nullptr;   // ncclComm_t of scvx_comm_create (RCCL, bound at run time: csrc/scvx_comm.hip)
    int comm_rank = 0, comm_world = 0;
    scvx::TdCache* td = nullptr;
    std::string err;
};

namespace scvx {

struct NcclId { char internal[SCVX_COMM_ID_BYTES]; };   // layout of ncclUniqueId (rccl.h), passed by value to RCCL

// K1: endpoint[B*K][14], deriv[B*K][14+2NU+1][14] from x[B][K+1][14], u[B][K+1][NU], sigma[B]  (NU = scvx_control_dim).
hipError_t launch_linearize(const scvx_ctx* ctx, int B, int K, const double* x, const double* u, const double* sigma,
                            double dt, double* endpoint, double* deriv, hipStream_t st, const int* skip = nullptr);
// K1 in double arithmetic with the derivative tiles stored as float (scvx_batch_set_linearization_f32)
hipError_t launch_linearize_store_f32(const scvx_ctx* ctx, int B, int K, const double* x, const double* u, const double* sigma,
                                      double dt, double* endpoint, float* deriv, hipStream_t st, const int* skip = nullptr);
// K2: xnext[B*K][14].
hipError_t launch_propagate(const scvx_ctx* ctx, int B, int K, const double* x, const double* u, const double* sigma,
                            double dt, double* xnext, hipStream_t st);

// fp32 forms (column-per-lane kernel / one thread per segment, float arithmetic): scvx_linearize_f32, scvx_propagate_f32
hipError_t launch_linearize_f32(const scvx_ctx* ctx, int B, int K, const float* x, const float* u, const float* sigma,
                                float dt, float* endpoint, float* deriv, hipStream_t st);
hipError_t launch_propagate_f32(const scvx_ctx* ctx, int B, int K, const float* x, const float* u, const float* sigma,
                                float dt, float* xnext, hipStream_t st);

// K0 (scvx_threedof.hip): the batched 3-DoF landing SOCP on device arrays, enqueued on ctx->stream; sol [B][(K+1)*15+1],
// info [B][6] = status, iters, pobj, gap, pres, dres.  threedof_to_record overwrites the trajectory records [B][(K+1)*(14+NU)+1]
// of the trajectories whose solve is optimal with the LinPoints of initial_solve.jl:90-105.
int threedof_solve_dev(scvx_ctx* ctx, int B, const double* ic_dev, const scvx_threedof_opts* opts, double* sol_dev,
                       double* info_dev);
int threedof_to_record(scvx_ctx* ctx, int B, int K, const double* sol_dev, const double* info_dev, double* rec_dev, int attitude);
void td_cache_free(scvx_ctx* ctx);

inline int fail(scvx_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    return code;
}

}  // namespace scvx

#define SCVX_HIP(ctx, call)                                                                        \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return scvx::fail((ctx), SCVX_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)
