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

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <thread>

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
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;  // optional
};

static RcclApi g_rccl;

static int rccl_load(fr_ctx* ctx)
{
    static std::mutex load_mutex;  // (contexts of several host threads may attach at the same time)
    std::lock_guard<std::mutex> lk(load_mutex);
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
    g_rccl.CommGetAsyncError = (decltype(g_rccl.CommGetAsyncError))dlsym(h, "ncclCommGetAsyncError");
    g_rccl.handle = h;
    return FR_OK;
}

#define FR_NCCL(ctx, call)                                                                            \
    do {                                                                                              \
        ncclResult_t r__ = (call);                                                                    \
        if (r__ != ncclSuccess)                                                                       \
            return set_err((ctx), FR_RCCL_ERROR, "%s failed: %s", #call, g_rccl.GetErrorString(r__)); \
    } while (0)

// inside a CallGuard named `guard`: ownership of the handles passes to the RCCL call and back PER CALL (enter / leave are
// compare-and-swaps on the watchdog's state word), so the watchdog can only abort while this thread is inside RCCL and this
// thread never enters RCCL on handles the watchdog has taken
#define FR_NCCL_G(ctx, call)                                                                                   \
    do {                                                                                                       \
        if (!guard.enter()) return set_err((ctx), FR_RCCL_ERROR, "communicators aborted by the watchdog");     \
        ncclResult_t r__ = (call);                                                                             \
        if (!guard.leave()) return set_err((ctx), FR_RCCL_ERROR, "communicators aborted by the watchdog");     \
        if (r__ != ncclSuccess)                                                                                \
            return set_err((ctx), FR_RCCL_ERROR, "%s failed: %s", #call, g_rccl.GetErrorString(r__));          \
    } while (0)

static inline int64_t now_ms()
{
    return std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---- watchdog ------------------------------------------------------------------------------------------
// RCCL collectives have no timeout.  Two ways a sharded factorisation can wait forever, two guards:
//   * on the DEVICE: a collective kernel spins for a peer that never arrives -> the stream never drains.  The host thread
//     never waits for such a stream with hipStreamSynchronize; it polls (comm_stream_sync) and, past the deadline
//     (option "comm_timeout_ms"), aborts both communicators itself -- ncclCommAbort makes the pending kernels exit -- and
//     returns FR_RCCL_ERROR.
//   * on the HOST: a call into RCCL blocks (connection set-up with a peer that never calls).  The thread that is stuck
//     cannot help itself, so every context with a communicator has a watchdog thread.  The state word `call` says who owns
//     the handles:
//         0                 no guarded sequence
//         2 * seq           the host thread is inside a guarded sequence, BETWEEN two RCCL calls: it runs, the watchdog keeps off
//         2 * seq + 1       the host thread is INSIDE an RCCL call (since call_t0); seq is bumped at EVERY enter(), so the word
//                           names one call, not one sequence
//         kWatchAborting    the watchdog has taken the handles
//     The host thread moves 2 seq -> 2 seq + 1 (CallGuard::enter) and back (leave) with compare-and-swaps around EVERY single
//     RCCL call; the watchdog moves 2 seq + 1 -> kWatchAborting, and only that: it aborts a call that has been in progress
//     for longer than the deadline -- the documented use of ncclCommAbort from a second thread -- and nothing else, so a
//     check-then-call window does not exist (round-4 advisor finding: the former single check in front of a call was not atomic
//     with it).  The watchdog only READS ctx->comm / comm2 (they cannot change while the host thread is inside a call); the
//     host thread, once its call returns and its leave() fails, waits for `aborted` and nulls them itself.
// After either, the context is "lost" (every collective fails at once) until the host calls fr_ctx_comm_finalize and attaches
// a new communicator; bench.py then falls back to a more conservative schedule (friedrich_amd/sharding.py: guarded_schedule).
constexpr uint64_t kWatchAborting = ~uint64_t(0);

struct CommWatch {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    bool stop = false;
    std::atomic<uint64_t> call{0};  // see above
    std::atomic<int64_t> call_t0{0};
    std::atomic<int> aborted{0};    // the watchdog has torn the communicators down
    uint64_t seq = 0;
};

// host thread only (never the watchdog): abort and forget both handles
static void abort_handles(fr_ctx* ctx)
{
    if (ctx->comm2 && g_rccl.CommAbort) (void)g_rccl.CommAbort((ncclComm_t)ctx->comm2);
    if (ctx->comm && g_rccl.CommAbort) (void)g_rccl.CommAbort((ncclComm_t)ctx->comm);
    ctx->comm2 = nullptr;
    ctx->comm = nullptr;
}

static void watch_main(fr_ctx* ctx, CommWatch* w)
{
    std::unique_lock<std::mutex> lk(w->m);
    while (!w->stop) {
        w->cv.wait_for(lk, std::chrono::milliseconds(25));
        if (w->stop) break;
        const uint64_t c = w->call.load();
        const int64_t T = ctx->comm_timeout_ms;
        if (c == 0 || c == kWatchAborting || (c & 1) == 0 || T <= 0) continue;  // (only a call IN PROGRESS can be stuck)
        if (now_ms() - w->call_t0.load() < T) continue;
        uint64_t expect = c;
        if (!w->call.compare_exchange_strong(expect, kWatchAborting)) continue;  // the call returned in the meantime
        // the host thread is inside RCCL and, when it comes back, will find its leave() refused: the handles are ours to abort
        // (read only: the host thread nulls them after `aborted`)
        void* c2 = ctx->comm2;
        void* c1 = ctx->comm;
        if (c2 && g_rccl.CommAbort) (void)g_rccl.CommAbort((ncclComm_t)c2);
        if (c1 && g_rccl.CommAbort) (void)g_rccl.CommAbort((ncclComm_t)c1);
        __atomic_fetch_add(&ctx->comm_timeouts, (int64_t)1, __ATOMIC_RELAXED);
        w->aborted.store(1);
    }
}

static void watch_start(fr_ctx* ctx)
{
    if (ctx->watch) return;
    CommWatch* w = new CommWatch();
    ctx->watch = w;
    w->th = std::thread(watch_main, ctx, w);
}

static void watch_stop(fr_ctx* ctx)
{
    CommWatch* w = (CommWatch*)ctx->watch;
    if (!w) return;
    {
        std::lock_guard<std::mutex> lk(w->m);
        w->stop = true;
    }
    w->cv.notify_all();
    if (w->th.joinable()) w->th.join();
    delete w;
    ctx->watch = nullptr;
}

// Brackets the RCCL host calls of one collective (a grouped send / receive is several calls).
struct CallGuard {
    fr_ctx* ctx;
    CommWatch* w;
    uint64_t mine = 0;  // 2 * seq
    bool lost = false;
    explicit CallGuard(fr_ctx* c) : ctx(c), w((CommWatch*)c->watch)
    {
        if (!w) return;
        mine = 2 * ++w->seq;
        w->call.store(mine);
    }
    // in front of one RCCL call; false: the watchdog has the handles, do not call
    bool enter()
    {
        if (!w) return true;
        if (lost) return false;
        w->call_t0.store(now_ms());
        // a FRESH sequence value per call (advisor finding, round 5: with one value per guarded sequence the watchdog could load
        // "call A in progress", see A past its deadline, and have its compare-and-swap succeed on call B's identical word -- an
        // ABA that aborted a call which had only just started)
        const uint64_t fresh = 2 * ++w->seq;
        uint64_t expect = mine;
        if (w->call.compare_exchange_strong(expect, fresh | 1)) {
            mine = fresh;
            return true;
        }
        take_note();
        return false;
    }
    // behind it; false: the watchdog took the handles over while the call was in progress
    bool leave()
    {
        if (!w) return true;
        uint64_t expect = mine | 1;
        if (w->call.compare_exchange_strong(expect, mine)) return true;
        take_note();
        return false;
    }
    bool taken_over() const { return lost; }
    // the watchdog decided a call was stuck and is tearing the communicators down (or has): wait for it, then forget the
    // handles (this thread is the only one that writes them) and mark the context
    void take_note()
    {
        if (lost) return;
        lost = true;
        while (!w->aborted.load()) std::this_thread::sleep_for(std::chrono::milliseconds(1));
        ctx->comm = ctx->comm2 = nullptr;
        ctx->comm_lost = true;
        set_err(ctx, FR_RCCL_ERROR, "an RCCL call did not return within %lld ms (dist_schedule %lld): communicators aborted by the watchdog",
                (long long)ctx->comm_timeout_ms, (long long)ctx->dist_schedule);
    }
    ~CallGuard()
    {
        if (!w || lost) return;
        uint64_t expect = mine;
        if (!w->call.compare_exchange_strong(expect, 0)) take_note();  // (cannot happen: the watchdog only takes 2 seq + 1)
    }
};

// An RCCL group opened on this thread is closed on EVERY way out of the sequence (an error of one of the grouped calls used to
// leave the thread's group depth above zero: the communicator created by the recovery path would then have been initialised
// inside an open group).  After a take-over the handles are gone and ncclGroupEnd would walk operations queued on a freed
// communicator, so it is skipped -- that needs a send / receive *enqueue* to block for the whole deadline first.
struct GroupGuard {
    CallGuard& guard;
    bool open = false;
    explicit GroupGuard(CallGuard& g) : guard(g) {}
    ~GroupGuard()
    {
        if (!open || guard.taken_over()) return;
        if (!guard.enter()) return;
        (void)g_rccl.GroupEnd();
        (void)guard.leave();
    }
};

// ncclCommInitRank with a deadline.  The call is a rendezvous of all ranks and blocks until the last one arrives; the
// communicator it is about to create does not exist yet, so no ncclCommAbort can reach it (round-4 advisor finding: the
// watchdog could only abort the OTHER, healthy communicator).  It therefore runs on a helper thread and this thread waits for
// it with the deadline (option "comm_timeout_ms"; 0: for ever).  Past the deadline the job is abandoned: the caller gets
// ncclSystemError at once, and the helper, if RCCL ever lets it go, aborts the communicator it was handed and frees the job.
// What an abandoned job costs (advisor finding, round 5): a detached thread that may sit inside ncclCommInitRank for ever, holding
// the device and the g_rccl entry points.  So (i) librccl is never dlclosed (rccl_load keeps the handle for the life of the
// process), (ii) the abandoned jobs are counted and a process that has kMaxAbandonedInits of them outstanding refuses further
// attempts at once (FR_RCCL_ERROR) instead of leaking a thread per fall-back of the schedule ladder, (iii) peers whose init DID
// complete hold a communicator with a rank that never arrived -- their first collective runs into comm_stream_sync's deadline and
// the context is lost like after any other missing peer.
constexpr int kMaxAbandonedInits = 8;
static std::atomic<int> g_abandoned_inits{0};

struct InitJob {
    std::mutex m;
    std::condition_variable cv;
    bool done = false, abandoned = false;
    ncclComm_t comm = nullptr;
    ncclResult_t res = ncclSystemError;
};

static ncclResult_t init_rank_bounded(fr_ctx* ctx, ncclComm_t* out, int world, const ncclUniqueId& id, int rank, bool* timed_out)
{
    *out = nullptr;
    *timed_out = false;
    if (g_abandoned_inits.load() >= kMaxAbandonedInits) return ncclSystemError;  // (the caller reports FR_RCCL_ERROR)
    auto job = std::make_shared<InitJob>();
    const int device = ctx->device;
    std::thread([job, device, world, id, rank] {
        ncclComm_t c = nullptr;
        ncclResult_t r = hipSetDevice(device) == hipSuccess ? g_rccl.CommInitRank(&c, world, id, rank) : ncclUnhandledCudaError;
        std::lock_guard<std::mutex> lk(job->m);
        if (job->abandoned) {
            if (r == ncclSuccess && c && g_rccl.CommAbort) (void)g_rccl.CommAbort(c);
            g_abandoned_inits.fetch_sub(1);  // RCCL let the helper go after all
            return;
        }
        job->comm = r == ncclSuccess ? c : nullptr;
        job->res = r;
        job->done = true;
        job->cv.notify_all();
    }).detach();
    std::unique_lock<std::mutex> lk(job->m);
    const int64_t T = ctx->comm_timeout_ms;
    if (T > 0) {
        if (!job->cv.wait_for(lk, std::chrono::milliseconds(T), [&] { return job->done; })) {
            job->abandoned = true;
            g_abandoned_inits.fetch_add(1);
            *timed_out = true;
            __atomic_fetch_add(&ctx->comm_timeouts, (int64_t)1, __ATOMIC_RELAXED);
            return ncclSystemError;
        }
    } else {
        job->cv.wait(lk, [&] { return job->done; });
    }
    *out = job->comm;
    return job->res;
}

// ---- test hook: one rank goes missing ------------------------------------------------------------------------
// FRIEDRICH_AMD_TEST_COMM_HANG = "schedule,rank,nth": while the context runs dist_schedule `schedule`, rank `rank` does not take
// part in its nth collective (counted from 1 over the context's life) -- it stalls instead, the way a crashed or wedged peer
// would -- so that the tests can watch the time-outs and the schedule fall-back work.  Returns true when the call is the one.
static bool test_hang_hit(fr_ctx* ctx)
{
    if (ctx->test_hang_nth <= 0 || ctx->dist_schedule != ctx->test_hang_schedule || ctx->rank != ctx->test_hang_rank) return false;
    return ++ctx->test_comm_calls == ctx->test_hang_nth;
}

static int test_hang_stall(fr_ctx* ctx)
{
    const int64_t T = ctx->comm_timeout_ms > 0 ? ctx->comm_timeout_ms : 2000;
    std::this_thread::sleep_for(std::chrono::milliseconds(T + T / 2 + 200));
    comm_abort(ctx);
    return set_err(ctx, FR_RCCL_ERROR, "injected hang (FRIEDRICH_AMD_TEST_COMM_HANG): this rank skipped a collective");
}

// ---- local transport ---------------------------------------------------------------------------------
// Default: ASYNCHRONOUS -- like RCCL, a collective is work enqueued on the calling rank's stream and ordered against the
// peers' streams by events, never by waiting for a stream on the host: a rank records "my buffer is ready" on its stream,
// the ranks exchange {pointer, events} through a host rendezvous (host threads stay in step per collective; the DEVICE
// work does not), every receiver makes its stream wait for the sender's event, copies device-to-device on its own stream
// and records "I have read it", and after a second rendezvous the sender's stream waits for those.  A missing event
// dependency BETWEEN THE STREAMS OF ONE RANK (chain / bulk / main) therefore shows up here as a wrong factor, exactly as it
// would over RCCL -- round 3's transport synchronised the stream around every collective and could not expose one.
// FRIEDRICH_AMD_LOCAL_SYNC=1 restores that behaviour (debugging aid).
struct LocalPub {
    const void* ptr = nullptr;
    hipEvent_t ready = nullptr, done = nullptr;
};
struct LocalGroup {
    int world = 0;
    int attached = 0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t gen = 0;
    LocalPub pubs[2][64];  // double-buffered by barrier generation parity
    bool broken = false;
    bool async = true;
    // The ranks' ready / done events are destroyed with the GROUP, by the last rank to detach: a peer that is still on its way
    // from the second rendezvous of an exchange to its hipStreamWaitEvent calls holds the bare handles (LocalPub), and the rank
    // they belong to may be through its teardown by then (a rank that was descheduled for the milliseconds its peer needs to
    // finish -- the one crash of the thread-rank tests in round 5's suite runs is attributed to this window).
    std::vector<hipEvent_t> retired;
};
struct LocalComm {
    LocalGroup* g;
    int group_id;
    hipEvent_t ready = nullptr, done = nullptr;
};

static std::mutex g_local_mutex;
static std::map<int, LocalGroup*> g_local_groups;

// host rendezvous; every rank publishes {pointer, events} first and gets a snapshot of all of them (taken from the slot of
// THIS generation: a fast rank entering the next rendezvous writes the other slot).  false on timeout / broken group.
static bool local_barrier(fr_ctx* ctx, const void* publish, LocalPub* snapshot = nullptr)
{
    LocalComm* lc = (LocalComm*)ctx->local;
    LocalGroup* g = lc->g;
    const int rank = ctx->rank;
    std::unique_lock<std::mutex> lk(g->m);
    if (g->broken) return false;
    const uint64_t my_gen = g->gen;
    LocalPub* slot = g->pubs[my_gen & 1];
    slot[rank].ptr = publish;
    slot[rank].ready = lc->ready;
    slot[rank].done = lc->done;
    struct Snap {
        LocalPub* dst;
        LocalPub* src;
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
    const int64_t T = ctx->comm_timeout_ms > 0 ? ctx->comm_timeout_ms : 120000;
    const bool ok = g->cv.wait_for(lk, std::chrono::milliseconds(T), [&] { return g->gen != my_gen || g->broken; });
    if (!ok || g->broken) {
        if (!ok) __atomic_fetch_add(&ctx->comm_timeouts, (int64_t)1, __ATOMIC_RELAXED);
        g->broken = true;
        g->cv.notify_all();
        return false;
    }
    return true;
}

static int local_fail(fr_ctx* ctx, const char* what)
{
    ctx->comm_lost = true;
    return set_err(ctx, FR_RCCL_ERROR, "local %s: a peer did not arrive within %lld ms (dist_schedule %lld)", what,
                   (long long)(ctx->comm_timeout_ms > 0 ? ctx->comm_timeout_ms : 120000), (long long)ctx->dist_schedule);
}

// one transfer pattern for broadcast / fan-out (slices = 0: the whole buffer from the root), scatter (slice `rank` of the
// root's buffer to the same place of rank's) and all-gather (root < 0: slice r of rank r's `send` to recv + r * bytes)
static int local_exchange(fr_ctx* ctx, const char* what, const void* send, void* recv, size_t bytes, int root, bool sliced)
{
    LocalComm* lc = (LocalComm*)ctx->local;
    LocalGroup* g = lc->g;
    const int W = g->world, me = ctx->rank;
    const bool async = g->async;
    hipStream_t s = ctx->ls;
    if (async)
        FR_HIP(ctx, hipEventRecord(lc->ready, s));  // everything this rank contributes is ahead of this point on its stream
    else
        FR_HIP(ctx, hipStreamSynchronize(s));
    LocalPub all[64];
    if (!local_barrier(ctx, send, all)) return local_fail(ctx, what);
    auto pull = [&](int from, const char* src, char* dst, size_t nbytes) -> int {
        if (nbytes == 0 || src == dst) return FR_OK;  // (in place: the own slice is already there)
        if (async && from != me) FR_HIP(ctx, hipStreamWaitEvent(s, all[from].ready, 0));
        FR_HIP(ctx, hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToDevice, s));
        return FR_OK;
    };
    if (root >= 0) {
        if (me != root) {
            const size_t off = sliced ? (size_t)me * bytes : 0;
            FR_TRY(pull(root, (const char*)all[root].ptr + off, (char*)recv + off, bytes));
        }
    } else {
        for (int r = 0; r < W; ++r) FR_TRY(pull(r, (const char*)all[r].ptr, (char*)recv + (size_t)r * bytes, bytes));
    }
    if (async)
        FR_HIP(ctx, hipEventRecord(lc->done, s));  // this rank has read what it wanted from its peers
    else
        FR_HIP(ctx, hipStreamSynchronize(s));
    if (!local_barrier(ctx, nullptr, nullptr)) return local_fail(ctx, what);
    if (async) {
        // a sender's buffer may be overwritten by its later stream work only after every reader is done with it
        if (root >= 0) {
            if (me == root)
                for (int r = 0; r < W; ++r)
                    if (r != me) FR_HIP(ctx, hipStreamWaitEvent(s, all[r].done, 0));
        } else {
            for (int r = 0; r < W; ++r)
                if (r != me) FR_HIP(ctx, hipStreamWaitEvent(s, all[r].done, 0));
        }
    }
    return FR_OK;
}

static int local_bcast(fr_ctx* ctx, void* buf, size_t bytes, int root) { return local_exchange(ctx, "broadcast", buf, buf, bytes, root, false); }
static int local_allgather(fr_ctx* ctx, const void* send, void* recv, size_t bytes_per_rank)
{
    return local_exchange(ctx, "all-gather", send, recv, bytes_per_rank, -1, true);
}
static int local_scatter(fr_ctx* ctx, char* buf, size_t bytes_per_rank, int root)
{
    return local_exchange(ctx, "scatter", buf, buf, bytes_per_rank, root, true);
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

// common prologue of every collective: a lost communicator fails at once (never a silent single-rank no-op), the test hook
#define COMM_ENTER(ctx)                                                                                                   \
    do {                                                                                                                  \
        if ((ctx)->comm_lost || (!(ctx)->comm && !(ctx)->local))                                                          \
            return set_err((ctx), FR_RCCL_ERROR, "the communicator of this context was aborted (time-out or failed peer): " \
                                                 "call fr_ctx_comm_finalize and attach a new one");                      \
        if (test_hang_hit(ctx)) return test_hang_stall(ctx);                                                              \
    } while (0)

// Scatter: slice r (count doubles at buf + r * count) of the ROOT's buffer lands in the same place of rank r's buffer.
// RCCL: one grouped set of point-to-point sends from the root, each over its own xGMI link.
int comm_scatter(fr_ctx* ctx, double* buf, size_t count_per_rank, int root, int which)
{
    if (ctx->world <= 1 || count_per_rank == 0) return FR_OK;
    COMM_ENTER(ctx);
    ProfScope ps(ctx, FR_PROF_COMM, 0.0, 8.0 * (double)count_per_rank * (ctx->world - 1));
    if (ctx->local) return local_scatter(ctx, (char*)buf, 8 * count_per_rank, root);
    ncclComm_t comm = pick_comm(ctx, which);
    CallGuard guard(ctx);
    GroupGuard group(guard);
    FR_NCCL_G(ctx, g_rccl.GroupStart());
    group.open = true;
    if (ctx->rank == root) {
        for (int r = 0; r < ctx->world; ++r)
            if (r != root)
                FR_NCCL_G(ctx, g_rccl.Send(buf + (size_t)r * count_per_rank, count_per_rank, ncclDouble, r, comm, ctx->ls));
    } else {
        FR_NCCL_G(ctx, g_rccl.Recv(buf + (size_t)ctx->rank * count_per_rank, count_per_rank, ncclDouble, root, comm, ctx->ls));
    }
    group.open = false;  // (ncclGroupEnd closes the group whatever it returns)
    FR_NCCL_G(ctx, g_rccl.GroupEnd());
    return FR_OK;
}

// Fan-out: the root's buffer to every rank as ONE grouped set of point-to-point sends, each over its own xGMI link (the
// node is fully connected: 7 links per GPU).  For the few MB of a diagonal block this is the latency-optimal broadcast -- one
// hop, every link busy at once -- where a ring / tree broadcast forwards through intermediate ranks.
int comm_fanout(fr_ctx* ctx, double* buf, size_t count, int root, int which)
{
    if (ctx->world <= 1 || count == 0) return FR_OK;
    COMM_ENTER(ctx);
    ProfScope ps(ctx, FR_PROF_COMM, 0.0, 8.0 * (double)count * (ctx->world - 1));
    if (ctx->local) return local_bcast(ctx, buf, 8 * count, root);
    ncclComm_t comm = pick_comm(ctx, which);
    CallGuard guard(ctx);
    GroupGuard group(guard);
    FR_NCCL_G(ctx, g_rccl.GroupStart());
    group.open = true;
    if (ctx->rank == root) {
        for (int r = 0; r < ctx->world; ++r)
            if (r != root) FR_NCCL_G(ctx, g_rccl.Send(buf, count, ncclDouble, r, comm, ctx->ls));
    } else {
        FR_NCCL_G(ctx, g_rccl.Recv(buf, count, ncclDouble, root, comm, ctx->ls));
    }
    group.open = false;  // (ncclGroupEnd closes the group whatever it returns)
    FR_NCCL_G(ctx, g_rccl.GroupEnd());
    return FR_OK;
}

int comm_bcast(fr_ctx* ctx, double* buf, size_t count, int root)
{
    if (ctx->world <= 1) return FR_OK;
    COMM_ENTER(ctx);
    ProfScope ps(ctx, FR_PROF_COMM, 0.0, 8.0 * (double)count);
    if (ctx->local) return local_bcast(ctx, buf, 8 * count, root);
    CallGuard guard(ctx);
    FR_NCCL_G(ctx, g_rccl.Broadcast(buf, buf, count, ncclDouble, root, (ncclComm_t)ctx->comm, ctx->ls));
    return FR_OK;
}

int comm_allgather_i64(fr_ctx* ctx, const int64_t* send, int64_t* recv, size_t count_per_rank)
{
    if (ctx->world <= 1) {
        if (send != recv)
            FR_HIP(ctx, hipMemcpyAsync(recv, send, 8 * count_per_rank, hipMemcpyDeviceToDevice, ctx->ls));
        return FR_OK;
    }
    COMM_ENTER(ctx);
    if (ctx->local) return local_allgather(ctx, send, recv, 8 * count_per_rank);
    CallGuard guard(ctx);
    FR_NCCL_G(ctx, g_rccl.AllGather(send, recv, count_per_rank, ncclInt64, (ncclComm_t)ctx->comm, ctx->ls));
    return FR_OK;
}

int comm_allgather(fr_ctx* ctx, const double* send, double* recv, size_t count_per_rank, int which)
{
    if (ctx->world <= 1) {
        if (send != recv)
            FR_HIP(ctx, hipMemcpyAsync(recv, send, 8 * count_per_rank, hipMemcpyDeviceToDevice, ctx->ls));
        return FR_OK;
    }
    COMM_ENTER(ctx);
    ProfScope ps(ctx, FR_PROF_COMM, 0.0, 8.0 * (double)count_per_rank * ctx->world);
    if (ctx->local) return local_allgather(ctx, send, recv, 8 * count_per_rank);
    CallGuard guard(ctx);
    ncclComm_t comm = pick_comm(ctx, which);
    FR_NCCL_G(ctx, g_rccl.AllGather(send, recv, count_per_rank, ncclDouble, comm, ctx->ls));
    return FR_OK;
}

// After an abort the collectives in flight end (RCCL: their kernels see the abort flag) and the library's own device-side
// waits are bounded, so the streams empty; should they not (a wedged device), the host does not wait for ever either.
void comm_drain(fr_ctx* ctx)
{
    const int64_t t0 = now_ms();
    for (hipStream_t q : {ctx->stream2, ctx->stream3, ctx->stream}) {
        while (q && hipStreamQuery(q) == hipErrorNotReady && now_ms() - t0 < 20000) std::this_thread::sleep_for(std::chrono::milliseconds(1));
        (void)hipGetLastError();
    }
}

// Wait for a stream that may hold collectives -- never with hipStreamSynchronize alone: a collective whose peer never arrives
// would keep the host there for ever.  Polls; past the deadline (option "comm_timeout_ms", 0 = wait for ever) or on an
// asynchronous RCCL error the communicators are aborted (the pending collective kernels then exit), the streams are given a
// bounded time to drain, and the caller gets FR_RCCL_ERROR.
int comm_stream_sync(fr_ctx* ctx, hipStream_t s, const char* what)
{
    if (ctx->world <= 1 || ctx->comm_timeout_ms <= 0 || ctx->local || !ctx->comm) {
        FR_HIP(ctx, hipStreamSynchronize(s));
        return FR_OK;
    }
    const int64_t t0 = now_ms();
    int spins = 0;
    const char* why = "timed out";
    for (;;) {
        const hipError_t e = hipStreamQuery(s);
        if (e == hipSuccess) return FR_OK;
        if (e != hipErrorNotReady) return set_err(ctx, FR_HIP_ERROR, "hipStreamQuery failed while waiting for %s: %s", what, hipGetErrorString(e));
        (void)hipGetLastError();
        if (++spins > 2000) {  // (the first ~ms by yielding: the common case is a wait of a few hundred microseconds)
            std::this_thread::sleep_for(std::chrono::microseconds(50));
            if ((spins & 1023) == 0 && g_rccl.CommGetAsyncError && ctx->comm) {
                ncclResult_t ar = ncclSuccess;
                if (g_rccl.CommGetAsyncError((ncclComm_t)ctx->comm, &ar) == ncclSuccess && ar != ncclSuccess && ar != ncclInProgress) {
                    why = "reported an asynchronous RCCL error";
                    break;
                }
            }
        } else {
            std::this_thread::yield();
        }
        if (now_ms() - t0 > ctx->comm_timeout_ms) break;
    }
    __atomic_fetch_add(&ctx->comm_timeouts, (int64_t)1, __ATOMIC_RELAXED);
    const int64_t waited = now_ms() - t0;
    comm_abort(ctx);
    comm_drain(ctx);
    return set_err(ctx, FR_RCCL_ERROR, "%s %s after %lld ms (dist_schedule %lld, rank %d of %d): communicators aborted", what, why,
                   (long long)waited, (long long)ctx->dist_schedule, ctx->rank, ctx->world);
}

int comm_d2h(fr_ctx* ctx, void* dst, const void* src, size_t bytes, hipStream_t s, const char* what)
{
    if (ctx->world > 1 && ctx->comm && !ctx->local && ctx->comm_timeout_ms > 0) FR_TRY(comm_stream_sync(ctx, s, what));
    FR_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s));
    return FR_OK;
}

// Every rank contributes ok (1) / failed (0); all learn whether EVERY rank is ok.  Used before the first panel exchange of a
// sharded factorisation: a rank that could not allocate its buffers must not leave its peers waiting inside a broadcast.
// Synchronises the launch stream (with the deadline).
int comm_agree(fr_ctx* ctx, bool ok, bool* all_ok)
{
    *all_ok = ok;
    if (ctx->world <= 1) return FR_OK;
    if (!ctx->agree_buf) FR_HIP(ctx, hipMalloc((void**)&ctx->agree_buf, sizeof(int64_t) * 65));
    int64_t mine = ok ? 1 : 0;
    FR_HIP(ctx, hipMemcpyAsync(ctx->agree_buf, &mine, sizeof(int64_t), hipMemcpyHostToDevice, ctx->ls));
    FR_TRY(comm_allgather_i64(ctx, ctx->agree_buf, ctx->agree_buf + 1, 1));
    std::vector<int64_t> h((size_t)ctx->world);
    FR_TRY(comm_d2h(ctx, h.data(), ctx->agree_buf + 1, sizeof(int64_t) * h.size(), ctx->ls, "status agreement of the ranks"));
    FR_TRY(comm_stream_sync(ctx, ctx->ls, "status agreement of the ranks"));
    for (int64_t v : h)
        if (v != 1) *all_ok = false;
    return FR_OK;
}

// A rank that fails on the host side in the middle of a sharded factorisation -- or whose wait ran out -- tears its
// communicators down instead of leaving through a normal return: its peers' pending collectives then end in an error (RCCL)
// / a broken rendezvous (local transport) rather than waiting forever.  The context keeps its rank and world size but is
// "lost": every collective fails until fr_ctx_comm_finalize.
void comm_abort(fr_ctx* ctx)
{
    CommWatch* w = (CommWatch*)ctx->watch;
    if (ctx->comm || ctx->comm2) {
        if (w && w->aborted.load()) {
            ctx->comm = ctx->comm2 = nullptr;  // (the watchdog has done it)
        } else {
            abort_handles(ctx);
        }
    }
    if (ctx->local) {
        LocalGroup* g = ((LocalComm*)ctx->local)->g;
        std::lock_guard<std::mutex> lk(g->m);
        g->broken = true;
        g->cv.notify_all();
    }
    ctx->comm_lost = true;
}

// Second communicator over the same ranks (bulk stream of the chain-first schedule), created on first use: its id is drawn
// by rank 0 and travels over the first communicator.  Collective: every rank of the first communicator calls it at the same
// point of its host program (the start of a schedule-2 factorisation / the self-test).  A zeroed id tells the peers that rank
// 0 could not draw one; a rank whose hand-over fails locally (RCCL error, time-out: its context is lost and its first
// communicator aborted) leaves, and its peers' ncclCommInitRank -- which has a deadline of its own, init_rank_bounded -- or
// their status agreement on the first communicator ends in FR_RCCL_ERROR instead of waiting for it.  No way out of this
// function leaves a half-made communicator behind.
int ensure_comm2(fr_ctx* ctx)
{
    if (ctx->world <= 1 || ctx->local || ctx->comm2) return FR_OK;
    COMM_ENTER(ctx);
    if (!ctx->agree_buf) FR_HIP(ctx, hipMalloc((void**)&ctx->agree_buf, sizeof(int64_t) * 65));
    static_assert(sizeof(ncclUniqueId) <= sizeof(int64_t) * 64, "id fits the agreement buffer");
    ncclUniqueId id2;
    memset(&id2, 0, sizeof(id2));
    if (ctx->rank == 0 && g_rccl.GetUniqueId(&id2) != ncclSuccess) memset(&id2, 0, sizeof(id2));
    void* d = ctx->agree_buf + 1;
    hipStream_t s = ctx->stream;
    FR_HIP(ctx, hipMemcpyAsync(d, &id2, sizeof(id2), hipMemcpyHostToDevice, s));
    {
        CallGuard guard(ctx);
        FR_NCCL_G(ctx, g_rccl.Broadcast(d, d, sizeof(id2), ncclChar, 0, (ncclComm_t)ctx->comm, s));
    }
    FR_TRY(comm_d2h(ctx, &id2, d, sizeof(id2), s, "hand-over of the second communicator's id"));
    FR_TRY(comm_stream_sync(ctx, s, "hand-over of the second communicator's id"));
    bool id_ok = false;
    for (size_t i = 0; i < sizeof(id2); ++i) id_ok = id_ok || ((const char*)&id2)[i] != 0;
    ncclComm_t comm2 = nullptr;
    bool ok = id_ok, timed_out = false;
    if (id_ok) ok = init_rank_bounded(ctx, &comm2, ctx->world, id2, ctx->rank, &timed_out) == ncclSuccess;
    if (timed_out) {
        // a peer never arrived: the first communicator may be healthy, but the ranks no longer agree on what comes next
        comm_abort(ctx);
        return set_err(ctx, FR_RCCL_ERROR, "creation of the second communicator did not finish within %lld ms (rank %d of %d): communicators aborted",
                       (long long)ctx->comm_timeout_ms, ctx->rank, ctx->world);
    }
    bool all_ok = false;
    hipStream_t saved = ctx->ls;
    ctx->ls = s;
    const int st = comm_agree(ctx, ok, &all_ok);
    ctx->ls = saved;
    if (st != FR_OK || !all_ok || ctx->comm_lost) {
        if (comm2) (void)g_rccl.CommAbort(comm2);
        if (st != FR_OK) return st;
        if (ctx->comm_lost) return FR_RCCL_ERROR;
        return set_err(ctx, FR_RCCL_ERROR, id_ok ? "a rank could not create the second communicator" : "rank 0 could not draw an id for the second communicator");
    }
    ctx->comm2 = comm2;
    return FR_OK;
}

}  // namespace fr

using namespace fr;

extern "C" {

static void comm_release(fr_ctx* ctx, bool abort)
{
    watch_stop(ctx);  // first: nobody else touches the handles from here on
    if (ctx->comm2) {
        if (abort && g_rccl.CommAbort) (void)g_rccl.CommAbort((ncclComm_t)ctx->comm2);
        else if (g_rccl.CommDestroy) (void)g_rccl.CommDestroy((ncclComm_t)ctx->comm2);
        ctx->comm2 = nullptr;
    }
    if (ctx->comm) {
        if (abort && g_rccl.CommAbort) (void)g_rccl.CommAbort((ncclComm_t)ctx->comm);
        else if (g_rccl.CommDestroy) (void)g_rccl.CommDestroy((ncclComm_t)ctx->comm);
        ctx->comm = nullptr;
    }
    if (ctx->local) {
        LocalComm* lc = (LocalComm*)ctx->local;
        {
            std::lock_guard<std::mutex> lk(g_local_mutex);
            if (abort) {
                std::lock_guard<std::mutex> lk2(lc->g->m);
                lc->g->broken = true;
                lc->g->cv.notify_all();
            }
            if (lc->ready) lc->g->retired.push_back(lc->ready);
            if (lc->done) lc->g->retired.push_back(lc->done);
            if (--lc->g->attached == 0) {
                for (hipEvent_t e : lc->g->retired) (void)hipEventDestroy(e);
                g_local_groups.erase(lc->group_id);
                delete lc->g;
            }
        }
        delete lc;
        ctx->local = nullptr;
    }
    ctx->comm_lost = false;
    ctx->rank = 0;
    ctx->world = 1;
}

}  // extern "C"

void fr::comm_destroy_internal(fr_ctx* ctx)
{
    comm_release(ctx, ctx->comm_lost);
    if (ctx->agree_buf) {
        (void)hipFree(ctx->agree_buf);
        ctx->agree_buf = nullptr;
    }
}

extern "C" {

int fr_ctx_comm_finalize(fr_ctx* ctx, int abort)
{
    if (!ctx) return FR_INVALID_ARGUMENT;
    FR_LOCK(ctx);
    (void)hipSetDevice(ctx->device);
    // whatever the communicator still had in flight is either complete or, with `abort`, about to be cut short
    if (abort || ctx->comm_lost) comm_abort(ctx);
    comm_drain(ctx);
    comm_release(ctx, abort != 0 || ctx->comm_lost);
    return FR_OK;
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
    if (ctx->comm || ctx->local || ctx->comm_lost) return set_err(ctx, FR_INVALID_ARGUMENT, "communicator already initialised (fr_ctx_comm_finalize first)");
    if (world_size == 1 && !unique_id) {
        ctx->rank = 0;
        ctx->world = 1;
        return FR_OK;
    }
    if (!unique_id) return FR_INVALID_ARGUMENT;
    FR_TRY(rccl_load(ctx));
    // (the one device buffer the communicator's own hand-shakes need, BEFORE any rank can be left waiting for this one)
    if (!ctx->agree_buf) FR_HIP(ctx, hipMalloc((void**)&ctx->agree_buf, sizeof(int64_t) * 65));
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclComm_t comm = nullptr;
    bool timed_out = false;
    const ncclResult_t ir = init_rank_bounded(ctx, &comm, world_size, id, rank, &timed_out);
    if (timed_out)
        return set_err(ctx, FR_RCCL_ERROR, "ncclCommInitRank did not finish within %lld ms (rank %d of %d): a peer never arrived", (long long)ctx->comm_timeout_ms,
                       rank, world_size);
    if (ir != ncclSuccess) return set_err(ctx, FR_RCCL_ERROR, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(ir));
    ctx->comm = comm;
    ctx->rank = rank;
    ctx->world = world_size;
    if (world_size > 1) watch_start(ctx);
    // the second communicator (chain-first schedule only) is created on first use: ensure_comm2
    return FR_OK;
}

int fr_ctx_comm_init_local(fr_ctx* ctx, int group_id, int rank, int world_size)
{
    if (!ctx || world_size < 1 || world_size > 64 || rank < 0 || rank >= world_size) return FR_INVALID_ARGUMENT;
    FR_LOCK(ctx);
    FR_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->comm || ctx->local || ctx->comm_lost) return set_err(ctx, FR_INVALID_ARGUMENT, "communicator already initialised (fr_ctx_comm_finalize first)");
    hipEvent_t ready = nullptr, done = nullptr;
    FR_HIP(ctx, hipEventCreateWithFlags(&ready, hipEventDisableTiming));
    if (hipEventCreateWithFlags(&done, hipEventDisableTiming) != hipSuccess) {
        (void)hipEventDestroy(ready);
        return set_err(ctx, FR_HIP_ERROR, "hipEventCreate failed");
    }
    std::lock_guard<std::mutex> lk(g_local_mutex);
    LocalGroup*& g = g_local_groups[group_id];
    if (!g) {
        g = new LocalGroup();
        g->world = world_size;
        const char* e = getenv("FRIEDRICH_AMD_LOCAL_SYNC");
        g->async = !(e && e[0] == '1');
    }
    if (g->world != world_size) {
        (void)hipEventDestroy(ready);
        (void)hipEventDestroy(done);
        return set_err(ctx, FR_INVALID_ARGUMENT, "local group %d has world size %d", group_id, g->world);
    }
    g->attached += 1;
    LocalComm* lc = new LocalComm();
    lc->g = g;
    lc->group_id = group_id;
    lc->ready = ready;
    lc->done = done;
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
    if (ctx->dist_schedule == 2) FR_TRY(ensure_comm2(ctx));  // (the chain-first schedule's second communicator: collective)
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
            st = comm_stream_sync(ctx, ctx->ls, "communicator self-test");
            ctx->ls = ctx->stream3;
            if (st == FR_OK) st = comm_allgather(ctx, d + (2 + (size_t)ctx->rank) * cnt, d + 2 * cnt, cnt, 1);
            if (st == FR_OK) st = comm_stream_sync(ctx, ctx->ls, "communicator self-test");
        }
        ctx->world = world_saved;
        if (st != FR_OK) break;
        if (comm_stream_sync(ctx, ctx->ls, "communicator self-test") != FR_OK ||
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
