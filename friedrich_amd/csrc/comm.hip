// comm.hip -- RCCL plumbing (one process per GPU; collectives over xGMI).  librccl is loaded lazily with
// dlopen so single-GPU users never pay for it; there is no other transport.
#include <dlfcn.h>
#include <cstdlib>
#include <rccl/rccl.h>

#include "fr_internal.hpp"

namespace fr {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static RcclApi g_rccl;

static int rccl_load(fr_ctx* ctx)
{
    if (g_rccl.handle) return FR_OK;
    // The RCCL build must match the process's HIP runtime (torch bundles its own pair): the host names it.
    void* h = nullptr;
    if (const char* p = getenv("FRIEDRICH_AMD_RCCL_PATH")) h = dlopen(p, RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);  // already loaded (e.g. by torch)
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return set_err(ctx, FR_RCCL_ERROR, "cannot load librccl: %s", dlerror());
#define LOAD(name)                                                               \
    g_rccl.name = (decltype(g_rccl.name))dlsym(h, "nccl" #name);                 \
    if (!g_rccl.name) return set_err(ctx, FR_RCCL_ERROR, "librccl lacks nccl" #name)
    LOAD(GetUniqueId);
    LOAD(CommInitRank);
    LOAD(CommDestroy);
    LOAD(Broadcast);
    LOAD(AllGather);
    LOAD(AllReduce);
    LOAD(GetErrorString);
#undef LOAD
    g_rccl.handle = h;
    return FR_OK;
}

#define FR_NCCL(ctx, call)                                                                                   \
    do {                                                                                                     \
        ncclResult_t r__ = (call);                                                                           \
        if (r__ != ncclSuccess)                                                                              \
            return set_err((ctx), FR_RCCL_ERROR, "%s failed: %s", #call, g_rccl.GetErrorString(r__));       \
    } while (0)

int comm_bcast(fr_ctx* ctx, double* buf, size_t count, int root)
{
    if (ctx->world <= 1) return FR_OK;
    ProfScope ps(ctx, FR_PROF_COMM, 0.0, 8.0 * (double)count);
    FR_NCCL(ctx, g_rccl.Broadcast(buf, buf, count, ncclDouble, root, (ncclComm_t)ctx->comm, ctx->ls));
    return FR_OK;
}

int comm_bcast_i64(fr_ctx* ctx, int64_t* buf, size_t count, int root)
{
    if (ctx->world <= 1) return FR_OK;
    FR_NCCL(ctx, g_rccl.Broadcast(buf, buf, count, ncclInt64, root, (ncclComm_t)ctx->comm, ctx->ls));
    return FR_OK;
}

int comm_allgather(fr_ctx* ctx, const double* send, double* recv, size_t count_per_rank)
{
    if (ctx->world <= 1) {
        if (send != recv) FR_HIP(ctx, hipMemcpyAsync(recv, send, 8 * count_per_rank, hipMemcpyDeviceToDevice, ctx->stream));
        return FR_OK;
    }
    ProfScope ps(ctx, FR_PROF_COMM, 0.0, 8.0 * (double)count_per_rank * ctx->world);
    FR_NCCL(ctx, g_rccl.AllGather(send, recv, count_per_rank, ncclDouble, (ncclComm_t)ctx->comm, ctx->ls));
    return FR_OK;
}

}  // namespace fr

using namespace fr;

extern "C" {

void fr_comm_destroy_internal(fr_ctx* ctx)
{
    if (ctx->comm && g_rccl.CommDestroy) {
        g_rccl.CommDestroy((ncclComm_t)ctx->comm);
        ctx->comm = nullptr;
    }
}

int fr_comm_unique_id(void* out_id)
{
    if (!out_id) return FR_INVALID_ARGUMENT;
    static_assert(sizeof(ncclUniqueId) == FR_COMM_ID_BYTES, "ncclUniqueId size");
    if (rccl_load(nullptr) != FR_OK) return FR_RCCL_ERROR;
    ncclUniqueId id;
    if (g_rccl.GetUniqueId(&id) != ncclSuccess) return FR_RCCL_ERROR;
    memcpy(out_id, &id, sizeof(id));
    return FR_OK;
}

int fr_ctx_comm_init(fr_ctx* ctx, int rank, int world_size, const void* unique_id)
{
    if (!ctx || world_size < 1 || rank < 0 || rank >= world_size) return FR_INVALID_ARGUMENT;
    FR_HIP(ctx, hipSetDevice(ctx->device));
    ctx->rank = rank;
    ctx->world = world_size;
    if (world_size == 1) return FR_OK;
    if (!unique_id) return FR_INVALID_ARGUMENT;
    FR_TRY(rccl_load(ctx));
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclComm_t comm = nullptr;
    FR_NCCL(ctx, g_rccl.CommInitRank(&comm, world_size, id, rank));
    ctx->comm = comm;
    return FR_OK;
}

int fr_ctx_comm_info(const fr_ctx* ctx, int* rank, int* world_size)
{
    if (!ctx) return FR_INVALID_ARGUMENT;
    if (rank) *rank = ctx->rank;
    if (world_size) *world_size = ctx->world;
    return FR_OK;
}

}  // extern "C"
