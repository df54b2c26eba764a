// rccl_mock.hip -- TEST-ONLY stand-in for librccl (never shipped, never loaded unless FRIEDRICH_AMD_RCCL_PATH names it).
//
// RCCL refuses two ranks on one device ("Duplicate GPU detected", RCCL 2.26.6), and the builder's boxes have one GPU, so in five
// rounds the RCCL branch of friedrich_amd/csrc/comm.hip -- CallGuard / GroupGuard / the watchdog / init_rank_bounded / ensure_comm2 --
// had only ever met a 1-rank communicator.  This library implements the dozen nccl* entry points that comm.hip resolves with dlsym
// for ranks that are HOST THREADS of one process sharing one device, so that those code paths execute with peers:
//
//   ncclGetUniqueId, ncclCommInitRank (a rendezvous of all ranks, like the real one), ncclCommDestroy, ncclCommAbort (wakes this
//   rank's blocked calls, ends its device-side waits), ncclBroadcast, ncclAllGather, ncclSend / ncclRecv inside ncclGroupStart /
//   ncclGroupEnd (the group end is a rendezvous of the whole communicator: comm.hip's scatter and fan-out are called by every
//   rank), ncclCommGetAsyncError, ncclGetErrorString.
//
// Semantics kept from the real thing, because the library's guards are written against them:
//   * a collective is WORK ON THE CALLER'S STREAM, ordered against the peers' streams by events (ready -> copy -> done), never by
//     synchronising a stream on the host;
//   * a call blocks the HOST while a peer has not made the matching call (connection set-up in the real library) -- for ever,
//     unless this rank's communicator is aborted from another thread (the watchdog's use of ncclCommAbort);
//   * with RCCL_MOCK_RENDEZVOUS_MS=<t> a call gives up waiting on the host after t ms and puts a DEVICE-side wait on the stream
//     instead (a one-wave kernel that polls the communicator's abort word, capped at 60 s): the stream then never drains until
//     ncclCommAbort -- a collective kernel spinning for a peer that never arrives, the case comm_stream_sync polls for;
//   * RCCL_MOCK_INIT_ABSENT_RANK=<r>: rank r's ncclCommInitRank never completes (bounded at 30 s): the rendezvous no abort can reach.
// Data movement: device-to-device copies on the receiver's stream (all ranks share the device).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

using clk = std::chrono::steady_clock;

size_t type_bytes(ncclDataType_t t)
{
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        default: return 8;  // int64, uint64, float64
    }
}

struct P2P {
    bool send;
    void* buf;
    size_t bytes;
    int peer;
};

struct Pub {  // what a rank shows its peers for one operation
    const void* send = nullptr;
    void* recv = nullptr;
    size_t bytes = 0;
    int root = 0;
    int kind = 0;  // 1 broadcast, 2 all-gather, 3 p2p group
    std::vector<P2P> ops;
    hipEvent_t ready = nullptr, done = nullptr;
};

struct Group;

struct Comm {
    std::shared_ptr<Group> g;
    int rank = 0;
    std::atomic<int> aborted{0};
    unsigned* abort_word = nullptr;  // host-mapped: device-side waits of this rank poll it
    hipEvent_t ready[4] = {}, done[4] = {};
    uint64_t seq = 0;
    bool broken = false;  // a rendezvous gave up: the stream holds a device-side wait
};

struct Group {
    int world = 0;
    std::mutex m;
    std::condition_variable cv;
    int init_arrived = 0;
    // one rendezvous at a time (every rank issues the communicator's operations in the same order)
    uint64_t gen = 0;
    int arrived = 0;
    std::vector<Pub> pubs[2];
    std::vector<Comm*> comms;
};

std::mutex g_reg_m;
std::map<std::string, std::weak_ptr<Group>> g_reg;
std::atomic<uint64_t> g_id_counter{1};

thread_local int tl_group_depth = 0;
thread_local std::vector<std::pair<Comm*, std::pair<P2P, hipStream_t>>> tl_group_ops;

long env_ms(const char* name, long dflt)
{
    const char* v = getenv(name);
    return v ? atol(v) : dflt;
}

__global__ void mock_wait_kernel(volatile unsigned* abort_word)
{
    // a collective kernel waiting for a peer that never arrives: ends on ncclCommAbort (or after 60 s: never hang a test box)
    const unsigned long long t0 = wall_clock64();
    while (*abort_word == 0u && wall_clock64() - t0 < 6000000000ull) __builtin_amdgcn_s_sleep(64);
}

// Barrier of the group's ranks for operation generation `gen`.  Returns 0 when everybody is here, 1 when this rank's communicator
// was aborted while waiting, 2 when the host-side patience (RCCL_MOCK_RENDEZVOUS_MS) ran out.
int rendezvous(Comm* c, std::unique_lock<std::mutex>& lk, uint64_t my_gen)
{
    Group* g = c->g.get();
    if (++g->arrived == g->world) {
        g->arrived = 0;
        ++g->gen;
        g->cv.notify_all();
        return 0;
    }
    const long patience = env_ms("RCCL_MOCK_RENDEZVOUS_MS", -1);
    const auto t0 = clk::now();
    while (g->gen == my_gen) {
        g->cv.wait_for(lk, std::chrono::milliseconds(5));
        if (g->gen != my_gen) break;
        if (c->aborted.load()) {
            --g->arrived;
            return 1;
        }
        if (patience >= 0 && std::chrono::duration_cast<std::chrono::milliseconds>(clk::now() - t0).count() > patience) {
            --g->arrived;
            return 2;
        }
    }
    return 0;
}

ncclResult_t give_up_on_device(Comm* c, hipStream_t s)
{
    c->broken = true;
    hipLaunchKernelGGL(mock_wait_kernel, dim3(1), dim3(64), 0, s, (volatile unsigned*)c->abort_word);
    return hipGetLastError() == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

// One operation of the communicator: publish, meet, pull what is mine from the peers' buffers, meet again, wait for my readers.
ncclResult_t run_op(Comm* c, Pub mine, hipStream_t s)
{
    if (c->aborted.load()) return ncclInvalidUsage;
    if (c->broken) return ncclSuccess;  // (queued behind the device-side wait: it will never run)
    Group* g = c->g.get();
    const int slot = (int)(c->seq & 3);
    const int par = (int)(c->seq & 1);
    ++c->seq;
    mine.ready = c->ready[slot];
    mine.done = c->done[slot];
    if (hipEventRecord(mine.ready, s) != hipSuccess) return ncclUnhandledCudaError;
    std::vector<Pub> peers;
    {
        std::unique_lock<std::mutex> lk(g->m);
        g->pubs[par][c->rank] = mine;
        const uint64_t gen = g->gen;
        const int r = rendezvous(c, lk, gen);
        if (r == 1) return ncclSystemError;
        if (r == 2) {
            lk.unlock();
            return give_up_on_device(c, s);
        }
        peers = g->pubs[par];
    }
    const int W = g->world, me = c->rank;
    bool ok = true;
    auto pull = [&](int from, const void* src, void* dst, size_t bytes) {
        if (bytes == 0) return;
        if (from != me) ok = ok && hipStreamWaitEvent(s, peers[from].ready, 0) == hipSuccess;
        if (src != dst) ok = ok && hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s) == hipSuccess;
    };
    if (mine.kind == 1) {
        if (me != mine.root) pull(mine.root, peers[mine.root].send, mine.recv, mine.bytes);
        else if (mine.send != mine.recv) pull(me, mine.send, mine.recv, mine.bytes);
    } else if (mine.kind == 2) {
        for (int r = 0; r < W; ++r) pull(r, peers[r].send, (char*)mine.recv + (size_t)r * mine.bytes, mine.bytes);
    } else {
        // my k-th receive from peer p matches p's k-th send to me
        std::vector<int> next(W, 0);
        for (const P2P& op : mine.ops) {
            if (op.send) continue;
            const std::vector<P2P>& theirs = peers[op.peer].ops;
            int seen = 0;
            const P2P* match = nullptr;
            for (const P2P& t : theirs)
                if (t.send && t.peer == me && seen++ == next[op.peer]) {
                    match = &t;
                    break;
                }
            ++next[op.peer];
            if (!match || match->bytes != op.bytes) return ncclInvalidUsage;
            pull(op.peer, match->buf, op.buf, op.bytes);
        }
    }
    ok = ok && hipEventRecord(mine.done, s) == hipSuccess;
    {
        std::unique_lock<std::mutex> lk(g->m);
        const uint64_t gen = g->gen;
        const int r = rendezvous(c, lk, gen);
        if (r == 1) return ncclSystemError;
        if (r == 2) {
            lk.unlock();
            return give_up_on_device(c, s);
        }
    }
    // my buffer may be overwritten by what I enqueue next: not before its readers are through
    for (int r = 0; r < W; ++r)
        if (r != me) ok = ok && hipStreamWaitEvent(s, peers[r].done, 0) == hipSuccess;
    return ok ? ncclSuccess : ncclUnhandledCudaError;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "rccl-mock-%llu-%llu", (unsigned long long)g_id_counter.fetch_add(1),
             (unsigned long long)clk::now().time_since_epoch().count());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int world, ncclUniqueId id, int rank)
{
    if (!out || world < 1 || rank < 0 || rank >= world) return ncclInvalidArgument;
    const std::string key(id.internal, strnlen(id.internal, sizeof(id.internal)));
    std::shared_ptr<Group> g;
    {
        std::lock_guard<std::mutex> lk(g_reg_m);
        g = g_reg[key].lock();
        if (!g) {
            g = std::make_shared<Group>();
            g->world = world;
            g->pubs[0].resize(world);
            g->pubs[1].resize(world);
            g->comms.assign(world, nullptr);
            g_reg[key] = g;
        }
    }
    if (g->world != world) return ncclInvalidArgument;
    const long absent = env_ms("RCCL_MOCK_INIT_ABSENT_RANK", -1);
    if (absent == rank) {
        std::this_thread::sleep_for(std::chrono::seconds(30));
        return ncclSystemError;
    }
    Comm* c = new Comm();
    c->g = g;
    c->rank = rank;
    void* h = nullptr;
    if (hipHostMalloc(&h, 64, hipHostMallocMapped) != hipSuccess) {
        delete c;
        return ncclUnhandledCudaError;
    }
    memset(h, 0, 64);
    c->abort_word = (unsigned*)h;
    for (int i = 0; i < 4; ++i)
        if (hipEventCreateWithFlags(&c->ready[i], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->done[i], hipEventDisableTiming) != hipSuccess)
            return ncclUnhandledCudaError;
    // the rendezvous of the real call: nobody returns before everybody is here (bounded at 60 s: a test must end)
    {
        std::unique_lock<std::mutex> lk(g->m);
        g->comms[rank] = c;
        ++g->init_arrived;
        g->cv.notify_all();
        const auto t0 = clk::now();
        while (g->init_arrived < world) {
            g->cv.wait_for(lk, std::chrono::milliseconds(5));
            if (std::chrono::duration_cast<std::chrono::seconds>(clk::now() - t0).count() > 60) return ncclSystemError;
        }
    }
    *out = (ncclComm_t)c;
    return ncclSuccess;
}

static void release_comm(Comm* c)
{
    // (events and the abort word are left to the process: a peer may still hold the bare handles on its way out of a rendezvous)
    c->aborted.store(1);
    if (c->abort_word) *(volatile unsigned*)c->abort_word = 1u;
    c->g->cv.notify_all();
}

ncclResult_t ncclCommAbort(ncclComm_t comm)
{
    if (!comm) return ncclInvalidArgument;
    release_comm((Comm*)comm);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    if (!comm) return ncclInvalidArgument;
    release_comm((Comm*)comm);
    return ncclSuccess;
}

ncclResult_t ncclCommGetAsyncError(ncclComm_t comm, ncclResult_t* err)
{
    if (!comm || !err) return ncclInvalidArgument;
    *err = ncclSuccess;  // (a peer that never arrives is not an error RCCL reports: the caller's deadline has to find it)
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
        case ncclSuccess: return "no error (mock)";
        case ncclUnhandledCudaError: return "unhandled HIP error (mock)";
        case ncclSystemError: return "unhandled system error: communicator aborted or rendezvous failed (mock)";
        case ncclInvalidArgument: return "invalid argument (mock)";
        case ncclInvalidUsage: return "invalid usage (mock)";
        default: return "error (mock)";
    }
}

ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t t, int root, ncclComm_t comm, hipStream_t s)
{
    if (!comm) return ncclInvalidArgument;
    Pub p;
    p.kind = 1;
    p.send = send;
    p.recv = recv;
    p.bytes = count * type_bytes(t);
    p.root = root;
    return run_op((Comm*)comm, p, s);
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t t, ncclComm_t comm, hipStream_t s)
{
    if (!comm) return ncclInvalidArgument;
    Pub p;
    p.kind = 2;
    p.send = send;
    p.recv = recv;
    p.bytes = count * type_bytes(t);
    return run_op((Comm*)comm, p, s);
}

ncclResult_t ncclGroupStart()
{
    ++tl_group_depth;
    return ncclSuccess;
}

static ncclResult_t flush_group()
{
    if (tl_group_ops.empty()) return ncclSuccess;
    Comm* c = tl_group_ops[0].first;
    hipStream_t s = tl_group_ops[0].second.second;
    Pub p;
    p.kind = 3;
    for (auto& e : tl_group_ops) {
        if (e.first != c || e.second.second != s) {
            tl_group_ops.clear();
            return ncclInvalidUsage;  // (the mock handles one communicator and one stream per group: all comm.hip uses)
        }
        p.ops.push_back(e.second.first);
    }
    tl_group_ops.clear();
    return run_op(c, p, s);
}

ncclResult_t ncclGroupEnd()
{
    if (tl_group_depth <= 0) return ncclInvalidUsage;
    if (--tl_group_depth > 0) return ncclSuccess;
    return flush_group();
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s)
{
    if (!comm) return ncclInvalidArgument;
    tl_group_ops.push_back({(Comm*)comm, {P2P{true, const_cast<void*>(buf), count * type_bytes(t), peer}, s}});
    return tl_group_depth > 0 ? ncclSuccess : flush_group();
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s)
{
    if (!comm) return ncclInvalidArgument;
    tl_group_ops.push_back({(Comm*)comm, {P2P{false, buf, count * type_bytes(t), peer}, s}});
    return tl_group_depth > 0 ? ncclSuccess : flush_group();
}

}  // extern "C"
