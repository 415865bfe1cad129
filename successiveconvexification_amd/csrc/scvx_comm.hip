// The one exchange step of the path: an all-gather of the final trajectory records over RCCL / xGMI,
// owned by the context so that a host without torch (the Julia shim) can gather.  Nothing inside the SCvx iteration
// communicates: trajectories are independent (rocketland.jl:226-321 reads no other problem's data).
//
// RCCL is bound at run time (dlopen of librccl.so.1), not at link time: a process that already carries a copy of the
// library -- torch ships its own under the same soname -- gets THAT copy, so there are never two RCCL runtimes in one
// address space, and libscvx_hip.so loads on a single-GPU box whether or not RCCL is installed.
#include <dlfcn.h>
#include <cstdlib>
#include <cstring>
#include <string>
#include "scvx_internal.hpp"

namespace {

// the five entry points used, with the types of rccl.h (ncclResult_t and ncclDataType_t are ints on the wire)
struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, scvx::NcclId, int) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string err;
};

Rccl& rccl() {
    static Rccl r;
    return r;
}

bool load_rccl() {
    Rccl& r = rccl();
    if (r.handle) return true;
    // SCVX_RCCL_LIB names the one library to try (a site with its own build; tests point it at a missing file to see
    // the failure path), otherwise the usual sonames
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    const char* forced = std::getenv("SCVX_RCCL_LIB");
    const char* why = nullptr;
    if (forced && *forced) {
        r.handle = dlopen(forced, RTLD_NOW | RTLD_GLOBAL);
        if (!r.handle) why = dlerror();
    } else {
        for (const char* n : names) {
            r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
            if (const char* e = dlerror()) why = e;   // dlerror() clears the message: one call per failure
        }
    }
    if (!r.handle) {
        r.err = std::string("RCCL not found: ") + (why ? why : "dlopen failed");
        return false;
    }
    r.GetUniqueId = (int (*)(void*))dlsym(r.handle, "ncclGetUniqueId");
    r.CommInitRank = (int (*)(void**, int, scvx::NcclId, int))dlsym(r.handle, "ncclCommInitRank");
    r.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(r.handle, "ncclAllGather");
    r.CommDestroy = (int (*)(void*))dlsym(r.handle, "ncclCommDestroy");
    r.GetErrorString = (const char* (*)(int))dlsym(r.handle, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy) {
        r.err = "RCCL library lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy";
        dlclose(r.handle);
        r.handle = nullptr;
        return false;
    }
    return true;
}

int nccl_fail(scvx_ctx* ctx, const char* what, int rc) {
    const char* s = rccl().GetErrorString ? rccl().GetErrorString(rc) : "";
    return scvx::fail(ctx, SCVX_ERR_COMM, std::string(what) + " failed: " + (s ? s : "") + " (" + std::to_string(rc) + ")");
}

constexpr int kNcclInt32 = 2, kNcclFloat64 = 8;   // ncclDataType_t values of rccl.h

}  // namespace

extern "C" {

int scvx_comm_probe(void) { return load_rccl() ? SCVX_OK : SCVX_ERR_COMM; }

int scvx_comm_unique_id(void* id_out) {
    if (!id_out) return SCVX_ERR_ARG;
    if (!load_rccl()) return SCVX_ERR_COMM;
    scvx::NcclId id;
    std::memset(&id, 0, sizeof id);
    const int rc = rccl().GetUniqueId(&id);
    if (rc != 0) return SCVX_ERR_COMM;
    std::memcpy(id_out, &id, SCVX_COMM_ID_BYTES);
    return SCVX_OK;
}

int scvx_comm_create(scvx_ctx* ctx, const void* unique_id, int rank, int world) {
    if (!ctx || !unique_id) return SCVX_ERR_ARG;
    if (world < 1 || rank < 0 || rank >= world) return scvx::fail(ctx, SCVX_ERR_ARG, "need 0 <= rank < world");
    if (ctx->comm) return scvx::fail(ctx, SCVX_ERR_STATE, "the context already owns a communicator");
    if (!load_rccl()) return scvx::fail(ctx, SCVX_ERR_COMM, rccl().err);
    SCVX_HIP(ctx, hipSetDevice(ctx->device));
    scvx::NcclId id;
    std::memcpy(&id, unique_id, SCVX_COMM_ID_BYTES);
    void* comm = nullptr;
    const int rc = rccl().CommInitRank(&comm, world, id, rank);
    if (rc != 0) return nccl_fail(ctx, "ncclCommInitRank", rc);
    ctx->comm = comm;
    ctx->comm_rank = rank;
    ctx->comm_world = world;
    return SCVX_OK;
}

int scvx_comm_destroy(scvx_ctx* ctx) {
    if (!ctx) return SCVX_ERR_ARG;
    if (ctx->comm) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        rccl().CommDestroy(ctx->comm);
        ctx->comm = nullptr;
    }
    ctx->comm_world = 0;
    return SCVX_OK;
}

int scvx_comm_info(const scvx_ctx* ctx, int* rank, int* world) {
    if (!ctx) return SCVX_ERR_ARG;
    if (rank) *rank = ctx->comm ? ctx->comm_rank : 0;
    if (world) *world = ctx->comm ? ctx->comm_world : 0;
    return SCVX_OK;
}

int scvx_allgather_f64(scvx_ctx* ctx, const double* send_dev, double* recv_dev, int64_t count) {
    if (!ctx || !send_dev || !recv_dev || count < 0) return SCVX_ERR_ARG;
    if (!ctx->comm) return scvx::fail(ctx, SCVX_ERR_STATE, "no communicator: call scvx_comm_create first");
    SCVX_HIP(ctx, hipSetDevice(ctx->device));
    const int rc = rccl().AllGather(send_dev, recv_dev, (size_t)count, kNcclFloat64, ctx->comm, ctx->stream);
    if (rc != 0) return nccl_fail(ctx, "ncclAllGather", rc);
    return SCVX_OK;
}

int scvx_allgather_i32(scvx_ctx* ctx, const int32_t* send_dev, int32_t* recv_dev, int64_t count) {
    if (!ctx || !send_dev || !recv_dev || count < 0) return SCVX_ERR_ARG;
    if (!ctx->comm) return scvx::fail(ctx, SCVX_ERR_STATE, "no communicator: call scvx_comm_create first");
    SCVX_HIP(ctx, hipSetDevice(ctx->device));
    const int rc = rccl().AllGather(send_dev, recv_dev, (size_t)count, kNcclInt32, ctx->comm, ctx->stream);
    if (rc != 0) return nccl_fail(ctx, "ncclAllGather", rc);
    return SCVX_OK;
}

}  // extern "C"
