// comm.hip -- collectives of the sharded Cholesky.
//
// Transport 1 (production): RCCL over xGMI, one process per GPU.  librccl is loaded lazily with dlopen so
// single-GPU users never pay for it; the RCCL build has to match the process's HIP runtime (torch bundles its
// own pair), so the host may name it through FRIEDRICH_AMD_RCCL_PATH.
// Transport 2 ("local"): the ranks are host threads of ONE process that share a device; a broadcast is a
// device-to-device copy between the ranks' buffers, synchronised with a host barrier.  It exists so the sharded
// code path (ownership maps, panel pack/broadcast/unpack, stream ordering, info merge) can be exercised on a
// 1-GPU box; it is not a performance path.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <map>
#include <mutex>

#include "fr_internal.hpp"
#include <vector>

namespace fr {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static RcclApi g_rccl;

static int rccl_load(fr_ctx* ctx)
{
    if (g_rccl.handle) return FR_OK;
    void* h = nullptr;
    if (const char* p = getenv("FRIEDRICH_AMD_RCCL_PATH")) h = dlopen(p, RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);  // already loaded (e.g. by torch)
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return set_err(ctx, FR_RCCL_ERROR, "cannot load librccl: %s", dlerror());
#define LOAD(name)                                               \
    g_rccl.name = (decltype(g_rccl.name))dlsym(h, "nccl" #name); \
    if (!g_rccl.name) return set_err(ctx, FR_RCCL_ERROR, "librccl lacks nccl" #name)
    LOAD(GetUniqueId);
    LOAD(CommInitRank);
    LOAD(CommDestroy);
    LOAD(CommAbort);
    LOAD(Broadcast);
    LOAD(AllGather);
    LOAD(Send);
    LOAD(Recv);
    LOAD(GroupStart);
    LOAD(GroupEnd);
    LOAD(GetErrorString);
#undef LOAD
    g_rccl.handle = h;
    return FR_OK;
}

#define FR_NCCL(ctx, call)                                                                            \
    do {                                                                                              \
        ncclResult_t r__ = (call);                                                                    \
        if (r__ != ncclSuccess)                                                                       \
            return set_err((ctx), FR_RCCL_ERROR, "%s failed: %s", #call, g_rccl.GetErrorString(r__)); \
    } while (0)

// ---- local transport ---------------------------------------------------------------------------------
struct LocalGroup {
    int world = 0;
    int attached = 0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t gen = 0;
    const void* ptrs[2][64] = {{nullptr}};  // double-buffered by barrier generation parity
    bool broken = false;
};
struct LocalComm {
    LocalGroup* g;
    int group_id;
};

static std::mutex g_local_mutex;
static std::map<int, LocalGroup*> g_local_groups;

// host barrier; every rank publishes one pointer first and gets a snapshot of all of them (taken from the slot of
// THIS barrier generation: a fast rank entering the next barrier writes the other slot).  false on timeout.
static bool local_barrier(LocalGroup* g, int rank, const void* publish, const void** snapshot = nullptr)
{
    std::unique_lock<std::mutex> lk(g->m);
    if (g->broken) return false;
    const uint64_t my_gen = g->gen;
    const void** slot = g->ptrs[my_gen & 1];
    slot[rank] = publish;
    struct Snap {
        const void** dst;
        const void** src;
        int n;
        ~Snap()
        {
            if (dst)
                for (int i = 0; i < n; ++i) dst[i] = src[i];
        }
    } snap{snapshot, slot, g->world};
    if (++g->arrived == g->world) {
        g->arrived = 0;
        ++g->gen;
        g->cv.notify_all();
        return true;
    }
    const bool ok = g->cv.wait_for(lk, std::chrono::seconds(120), [&] { return g->gen != my_gen || g->broken; });
    if (!ok || g->broken) {
        g->broken = true;
        g->cv.notify_all();
        return false;
    }
    return true;
}

static int local_bcast(fr_ctx* ctx, void* buf, size_t bytes, int root)
{
    LocalGroup* g = ((LocalComm*)ctx->local)->g;
    FR_HIP(ctx, hipStreamSynchronize(ctx->ls));
    const void* all[64];
    if (!local_barrier(g, ctx->rank, buf, all)) return set_err(ctx, FR_RCCL_ERROR, "local broadcast: barrier timed out");
    const void* src = all[root];
    if (ctx->rank != root && bytes > 0) {
        FR_HIP(ctx, hipMemcpyAsync(buf, src, bytes, hipMemcpyDeviceToDevice, ctx->ls));
        FR_HIP(ctx, hipStreamSynchronize(ctx->ls));
    }
    if (!local_barrier(g, ctx->rank, nullptr)) return set_err(ctx, FR_RCCL_ERROR, "local broadcast: barrier timed out");
    return FR_OK;
}

static int local_allgather(fr_ctx* ctx, const void* send, void* recv, size_t bytes_per_rank)
{
    LocalGroup* g = ((LocalComm*)ctx->local)->g;
    FR_HIP(ctx, hipStreamSynchronize(ctx->ls));
    const void* srcs[64];
    if (!local_barrier(g, ctx->rank, send, srcs)) return set_err(ctx, FR_RCCL_ERROR, "local allgather: barrier timed out");
    for (int r = 0; r < g->world; ++r)
        if (bytes_per_rank > 0 && (const char*)srcs[r] != (const char*)recv + (size_t)r * bytes_per_rank)  // (in place: own slice already there)
            FR_HIP(ctx, hipMemcpyAsync((char*)recv + (size_t)r * bytes_per_rank, srcs[r], bytes_per_rank,
                                       hipMemcpyDeviceToDevice, ctx->ls));
    FR_HIP(ctx, hipStreamSynchronize(ctx->ls));
    if (!local_barrier(g, ctx->rank, nullptr)) return set_err(ctx, FR_RCCL_ERROR, "local allgather: barrier timed out");
    return FR_OK;
}

static int local_scatter(fr_ctx* ctx, char* buf, size_t bytes_per_rank, int root)
{
    LocalGroup* g = ((LocalComm*)ctx->local)->g;
    FR_HIP(ctx, hipStreamSynchronize(ctx->ls));
    const void* all[64];
    if (!local_barrier(g, ctx->rank, buf, all)) return set_err(ctx, FR_RCCL_ERROR, "local scatter: barrier timed out");
    if (ctx->rank != root && bytes_per_rank > 0) {
        const size_t off = (size_t)ctx->rank * bytes_per_rank;
        FR_HIP(ctx, hipMemcpyAsync(buf + off, (const char*)all[root] + off, bytes_per_rank, hipMemcpyDeviceToDevice, ctx->ls));
        FR_HIP(ctx, hipStreamSynchronize(ctx->ls));
    }
    if (!local_barrier(g, ctx->rank, nullptr)) return set_err(ctx, FR_RCCL_ERROR, "local scatter: barrier timed out");
    return FR_OK;
}

// ---- dispatch ----------------------------------------------------------------------------------------
// which = 0: the context's first communicator (panel stream: everything the schedules 0 / 1 send, the diagonal chain of
// schedule 2); which = 1: the second one (bulk stream of schedule 2: scatter / all-gather of the rows below the chain).  Two
// communicators because operations of ONE communicator execute in issue order whatever stream they are on: the 4.5 MB of
// the diagonal chain would queue behind the 100 MB bulk transfers of the previous panel.  Every rank issues the operations
// of BOTH communicators in the same host order (chol.hip, potrf_dist_chain), which is what keeps two concurrently used
// communicators deadlock-free: the earliest unfinished operation in that order has all its dependencies complete and is at
// the head of its queue on every rank.
static inline ncclComm_t pick_comm(fr_ctx* ctx, int which) { return (ncclComm_t)((which == 1 && ctx->comm2) ? ctx->comm2 : ctx->comm); }

// Scatter: slice r (count doubles at buf + r * count) of the ROOT's buffer lands in the same place of rank r's buffer.
// RCCL: one grouped set of point-to-point sends from the root, each over its own xGMI link.
int comm_scatter(fr_ctx* ctx, double* buf, size_t count_per_rank, int root, int which)
{
    if (ctx->world <= 1 || count_per_rank == 0) return FR_OK;
    ProfScope ps(ctx, FR_PROF_COMM, 0.0, 8.0 * (double)count_per_rank * (ctx->world - 1));
    if (ctx->local) return local_scatter(ctx, (char*)buf, 8 * count_per_rank, root);
    ncclComm_t comm = pick_comm(ctx, which);
    FR_NCCL(ctx, g_rccl.GroupStart());
    if (ctx->rank == root) {
        for (int r = 0; r < ctx->world; ++r)
            if (r != root)
                FR_NCCL(ctx, g_rccl.Send(buf + (size_t)r * count_per_rank, count_per_rank, ncclDouble, r, comm, ctx->ls));
    } else {
        FR_NCCL(ctx, g_rccl.Recv(buf + (size_t)ctx->rank * count_per_rank, count_per_rank, ncclDouble, root, comm, ctx->ls));
    }
    FR_NCCL(ctx, g_rccl.GroupEnd());
    return FR_OK;
}

// Fan-out: the root's buffer to every rank as ONE grouped set of point-to-point sends, each over its own xGMI link (the
// node is fully connected: 7 links per GPU).  For the few MB of a diagonal block this is the latency-optimal broadcast -- one
// hop, every link busy at once -- where a ring / tree broadcast forwards through intermediate ranks.
int comm_fanout(fr_ctx* ctx, double* buf, size_t count, int root, int which)
{
    if (ctx->world <= 1 || count == 0) return FR_OK;
    ProfScope ps(ctx, FR_PROF_COMM, 0.0, 8.0 * (double)count * (ctx->world - 1));
    if (ctx->local) return local_bcast(ctx, buf, 8 * count, root);
    ncclComm_t comm = pick_comm(ctx, which);
    FR_NCCL(ctx, g_rccl.GroupStart());
    if (ctx->rank == root) {
        for (int r = 0; r < ctx->world; ++r)
            if (r != root) FR_NCCL(ctx, g_rccl.Send(buf, count, ncclDouble, r, comm, ctx->ls));
    } else {
        FR_NCCL(ctx, g_rccl.Recv(buf, count, ncclDouble, root, comm, ctx->ls));
    }
    FR_NCCL(ctx, g_rccl.GroupEnd());
    return FR_OK;
}

int comm_bcast(fr_ctx* ctx, double* buf, size_t count, int root)
{
    if (ctx->world <= 1) return FR_OK;
    ProfScope ps(ctx, FR_PROF_COMM, 0.0, 8.0 * (double)count);
    if (ctx->local) return local_bcast(ctx, buf, 8 * count, root);
    FR_NCCL(ctx, g_rccl.Broadcast(buf, buf, count, ncclDouble, root, (ncclComm_t)ctx->comm, ctx->ls));
    return FR_OK;
}

int comm_allgather_i64(fr_ctx* ctx, const int64_t* send, int64_t* recv, size_t count_per_rank)
{
    if (ctx->world <= 1) {
        if (send != recv)
            FR_HIP(ctx, hipMemcpyAsync(recv, send, 8 * count_per_rank, hipMemcpyDeviceToDevice, ctx->ls));
        return FR_OK;
    }
    if (ctx->local) return local_allgather(ctx, send, recv, 8 * count_per_rank);
    FR_NCCL(ctx, g_rccl.AllGather(send, recv, count_per_rank, ncclInt64, (ncclComm_t)ctx->comm, ctx->ls));
    return FR_OK;
}

int comm_allgather(fr_ctx* ctx, const double* send, double* recv, size_t count_per_rank, int which)
{
    if (ctx->world <= 1) {
        if (send != recv)
            FR_HIP(ctx, hipMemcpyAsync(recv, send, 8 * count_per_rank, hipMemcpyDeviceToDevice, ctx->ls));
        return FR_OK;
    }
    ProfScope ps(ctx, FR_PROF_COMM, 0.0, 8.0 * (double)count_per_rank * ctx->world);
    if (ctx->local) return local_allgather(ctx, send, recv, 8 * count_per_rank);
    FR_NCCL(ctx, g_rccl.AllGather(send, recv, count_per_rank, ncclDouble, pick_comm(ctx, which), ctx->ls));
    return FR_OK;
}

// Every rank contributes ok (1) / failed (0); all learn whether EVERY rank is ok.  Used before the first panel exchange of a
// sharded factorisation: a rank that could not allocate its buffers must not leave its peers waiting inside a broadcast
// (RCCL collectives have no timeout).  Synchronises the launch stream.
int comm_agree(fr_ctx* ctx, bool ok, bool* all_ok)
{
    *all_ok = ok;
    if (ctx->world <= 1) return FR_OK;
    if (!ctx->agree_buf) FR_HIP(ctx, hipMalloc((void**)&ctx->agree_buf, sizeof(int64_t) * 65));
    int64_t mine = ok ? 1 : 0;
    FR_HIP(ctx, hipMemcpyAsync(ctx->agree_buf, &mine, sizeof(int64_t), hipMemcpyHostToDevice, ctx->ls));
    FR_TRY(comm_allgather_i64(ctx, ctx->agree_buf, ctx->agree_buf + 1, 1));
    std::vector<int64_t> h((size_t)ctx->world);
    FR_HIP(ctx, hipMemcpyAsync(h.data(), ctx->agree_buf + 1, sizeof(int64_t) * h.size(), hipMemcpyDeviceToHost, ctx->ls));
    FR_HIP(ctx, hipStreamSynchronize(ctx->ls));
    for (int64_t v : h)
        if (v != 1) *all_ok = false;
    return FR_OK;
}

// A rank that fails on the host side in the middle of a sharded factorisation tears its communicator down instead of
// leaving through a normal return: its peers' pending collectives then end in an error (RCCL) / a broken barrier (local
// transport) rather than waiting forever.  The context cannot take part in collectives afterwards.
void comm_abort(fr_ctx* ctx)
{
    if (ctx->comm2 && g_rccl.CommAbort) {
        (void)g_rccl.CommAbort((ncclComm_t)ctx->comm2);
        ctx->comm2 = nullptr;
    }
    if (ctx->comm && g_rccl.CommAbort) {
        (void)g_rccl.CommAbort((ncclComm_t)ctx->comm);
        ctx->comm = nullptr;
        ctx->world = 1;
        ctx->rank = 0;
    }
    if (ctx->local) {
        LocalGroup* g = ((LocalComm*)ctx->local)->g;
        std::lock_guard<std::mutex> lk(g->m);
        g->broken = true;
        g->cv.notify_all();
    }
}

}  // namespace fr

using namespace fr;

extern "C" {

void fr_comm_destroy_internal(fr_ctx* ctx)
{
    if (ctx->comm2 && g_rccl.CommDestroy) {
        g_rccl.CommDestroy((ncclComm_t)ctx->comm2);
        ctx->comm2 = nullptr;
    }
    if (ctx->comm && g_rccl.CommDestroy) {
        g_rccl.CommDestroy((ncclComm_t)ctx->comm);
        ctx->comm = nullptr;
    }
    if (ctx->agree_buf) {
        (void)hipFree(ctx->agree_buf);
        ctx->agree_buf = nullptr;
    }
    if (ctx->local) {
        LocalComm* lc = (LocalComm*)ctx->local;
        {
            std::lock_guard<std::mutex> lk(g_local_mutex);
            if (--lc->g->attached == 0) {
                g_local_groups.erase(lc->group_id);
                delete lc->g;
            }
        }
        delete lc;
        ctx->local = nullptr;
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
    FR_LOCK(ctx);
    FR_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->comm || ctx->local) return set_err(ctx, FR_INVALID_ARGUMENT, "communicator already initialised");
    if (world_size == 1 && !unique_id) {
        ctx->rank = 0;
        ctx->world = 1;
        return FR_OK;
    }
    if (!unique_id) return FR_INVALID_ARGUMENT;
    FR_TRY(rccl_load(ctx));
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclComm_t comm = nullptr;
    FR_NCCL(ctx, g_rccl.CommInitRank(&comm, world_size, id, rank));
    ctx->comm = comm;
    ctx->rank = rank;
    ctx->world = world_size;
    if (world_size > 1) {
        // the second communicator (bulk stream of the chain-first schedule): rank 0 draws its id, the first communicator
        // carries it to the others
        ncclUniqueId id2;
        memset(&id2, 0, sizeof(id2));
        if (rank == 0) FR_NCCL(ctx, g_rccl.GetUniqueId(&id2));
        void* d = nullptr;
        FR_HIP(ctx, hipMalloc(&d, sizeof(id2)));
        hipError_t e = hipMemcpyAsync(d, &id2, sizeof(id2), hipMemcpyHostToDevice, ctx->stream);
        ncclResult_t r = ncclSuccess;
        if (e == hipSuccess) r = g_rccl.Broadcast(d, d, sizeof(id2), ncclChar, 0, comm, ctx->stream);
        if (e == hipSuccess && r == ncclSuccess) e = hipMemcpyAsync(&id2, d, sizeof(id2), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && r == ncclSuccess) e = hipStreamSynchronize(ctx->stream);
        (void)hipFree(d);
        if (e != hipSuccess) return set_err(ctx, FR_HIP_ERROR, "handing over the second communicator's id failed: %s", hipGetErrorString(e));
        if (r != ncclSuccess) return set_err(ctx, FR_RCCL_ERROR, "broadcast of the second communicator's id failed: %s", g_rccl.GetErrorString(r));
        ncclComm_t comm2 = nullptr;
        FR_NCCL(ctx, g_rccl.CommInitRank(&comm2, world_size, id2, rank));
        ctx->comm2 = comm2;
    }
    return FR_OK;
}

int fr_ctx_comm_init_local(fr_ctx* ctx, int group_id, int rank, int world_size)
{
    if (!ctx || world_size < 1 || world_size > 64 || rank < 0 || rank >= world_size) return FR_INVALID_ARGUMENT;
    FR_LOCK(ctx);
    if (ctx->comm || ctx->local) return set_err(ctx, FR_INVALID_ARGUMENT, "communicator already initialised");
    std::lock_guard<std::mutex> lk(g_local_mutex);
    LocalGroup*& g = g_local_groups[group_id];
    if (!g) {
        g = new LocalGroup();
        g->world = world_size;
    }
    if (g->world != world_size) return set_err(ctx, FR_INVALID_ARGUMENT, "local group %d has world size %d", group_id, g->world);
    g->attached += 1;
    LocalComm* lc = new LocalComm();
    lc->g = g;
    lc->group_id = group_id;
    ctx->local = lc;
    ctx->rank = rank;
    ctx->world = world_size;
    return FR_OK;
}

int fr_ctx_comm_selftest(fr_ctx* ctx)
{
    if (!ctx) return FR_INVALID_ARGUMENT;
    FR_LOCK(ctx);
    if (!ctx->comm && !ctx->local) return FR_OK;
    FR_HIP(ctx, hipSetDevice(ctx->device));
    const int W = ctx->world, R = ctx->rank;
    const size_t cnt = 256;
    std::vector<double> h(cnt * (size_t)(2 + W));
    for (size_t i = 0; i < cnt; ++i) {
        h[i] = 1000.0 * R + (double)i;        // broadcast buffer (rank 0's content must win)
        h[cnt + i] = 2000.0 * R + (double)i;  // all-gather contribution
    }
    double* d = nullptr;
    FR_HIP(ctx, hipMalloc(&d, sizeof(double) * h.size()));
    int st = FR_OK;
    hipStream_t saved = ctx->ls;
    ctx->ls = ctx->stream2 ? ctx->stream2 : ctx->stream;  // the panel stream carries the broadcasts of the factorisation
    do {
        if (hipMemcpyAsync(d, h.data(), sizeof(double) * 2 * cnt, hipMemcpyHostToDevice, ctx->ls) != hipSuccess) {
            st = set_err(ctx, FR_HIP_ERROR, "selftest: upload failed");
            break;
        }
        const int world_saved = ctx->world;
        if (W == 1 && ctx->comm) ctx->world = 2;  // a 1-rank RCCL communicator: still issue the real collectives
        st = comm_bcast(ctx, d, cnt, 0);
        if (st == FR_OK) st = comm_allgather(ctx, d + cnt, d + 2 * cnt, cnt, 0);
        // the chain-first schedule's calls: a fan-out on the first communicator, an (in-place) all-gather on the second
        if (st == FR_OK && W > 1) st = comm_fanout(ctx, d, cnt, 0, 0);  // (point-to-point: needs a real peer)
        if (st == FR_OK && ctx->stream3) {
            if (hipStreamSynchronize(ctx->ls) != hipSuccess) st = FR_HIP_ERROR;
            ctx->ls = ctx->stream3;
            if (st == FR_OK) st = comm_allgather(ctx, d + (2 + (size_t)ctx->rank) * cnt, d + 2 * cnt, cnt, 1);
            if (st == FR_OK && hipStreamSynchronize(ctx->ls) != hipSuccess) st = FR_HIP_ERROR;
        }
        ctx->world = world_saved;
        if (st != FR_OK) break;
        if (hipStreamSynchronize(ctx->ls) != hipSuccess ||
            hipMemcpy(h.data(), d, sizeof(double) * h.size(), hipMemcpyDeviceToHost) != hipSuccess) {
            st = set_err(ctx, FR_HIP_ERROR, "selftest: download failed");
            break;
        }
        for (size_t i = 0; i < cnt && st == FR_OK; ++i)
            if (h[i] != (double)i) st = set_err(ctx, FR_RCCL_ERROR, "selftest: broadcast payload mismatch at %zu", i);
        for (int r = 0; r < W && st == FR_OK; ++r)
            for (size_t i = 0; i < cnt; ++i)
                if (h[(2 + (size_t)r) * cnt + i] != 2000.0 * r + (double)i) {
                    st = set_err(ctx, FR_RCCL_ERROR, "selftest: all-gather payload mismatch (rank %d, %zu)", r, i);
                    break;
                }
    } while (0);
    ctx->ls = saved;
    (void)hipFree(d);
    return st;
}

int fr_ctx_comm_info(const fr_ctx* ctx, int* rank, int* world_size)
{
    if (!ctx) return FR_INVALID_ARGUMENT;
    if (rank) *rank = ctx->rank;
    if (world_size) *world_size = ctx->world;
    return FR_OK;
}

}  // extern "C"
