// ctx.hip -- context, error reporting, workspace pool, host<->device staging, event-based profiling.
#include <cstdarg>
#include <dlfcn.h>

#include "fr_internal.hpp"

namespace fr {

// ---- roctx ranges (tracing) --------------------------------------------------------------------------------
static int (*g_roctx_push)(const char*) = nullptr;
static int (*g_roctx_pop)() = nullptr;
static std::once_flag g_roctx_once;

static void roctx_load()
{
    const char* e = getenv("FRIEDRICH_AMD_ROCTX");
    if (!e || e[0] != '1') return;
    for (const char* lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
        void* h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
        if (!h) continue;
        g_roctx_push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
        g_roctx_pop = (int (*)())dlsym(h, "roctxRangePop");
        if (g_roctx_push && g_roctx_pop) return;
        g_roctx_push = nullptr;
        g_roctx_pop = nullptr;
    }
}

TraceScope::TraceScope(const char* name) : on(false)
{
    std::call_once(g_roctx_once, roctx_load);
    if (g_roctx_push) {
        (void)g_roctx_push(name);
        on = true;
    }
}

TraceScope::~TraceScope()
{
    if (on && g_roctx_pop) (void)g_roctx_pop();
}

int set_err(fr_ctx* ctx, int status, const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return status;
}

bool is_device_ptr(const void* p)
{
    if (!p) return false;
    hipPointerAttribute_t a;
    hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // plain malloc'd host memory: not an error for us
        return false;
    }
    return a.type == hipMemoryTypeDevice;
}

// Release every idle buffer of the workspace pool (after fr_grad_terms or a wide predict tens of GB can sit there).
size_t ws_trim(fr_ctx* ctx)
{
    size_t freed = 0;
    bool synced = false;
    for (auto& b : ctx->pool)
        if (!b.in_use && b.p) {
            if (!synced) {  // stream-ordered reuse: the last kernels that touched the buffers must be done
                (void)hipStreamSynchronize(ctx->stream);
                if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
                synced = true;
            }
            (void)hipFree(b.p);
            freed += b.cap;
            b.p = nullptr;
            b.cap = 0;
        }
    return freed;
}

// hipMalloc that retries once after trimming the pool: used for EVERY device allocation of the library, so that a factor
// (or its growth) never fails with FR_OUT_OF_MEMORY while idle workspaces hold the memory
hipError_t dev_malloc(fr_ctx* ctx, void** p, size_t bytes)
{
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipSuccess) return e;
    (void)hipGetLastError();
    if (ws_trim(ctx) == 0) return e;
    e = hipMalloc(p, bytes);
    if (e != hipSuccess) (void)hipGetLastError();
    return e;
}

void* ws_get(fr_ctx* ctx, size_t bytes)
{
    if (bytes == 0) bytes = 8;
    // best fit among free buffers
    int best = -1;
    for (int i = 0; i < (int)ctx->pool.size(); ++i) {
        DevBuf& b = ctx->pool[i];
        if (!b.in_use && b.cap >= bytes && (best < 0 || b.cap < ctx->pool[best].cap)) best = i;
    }
    if (best >= 0 && ctx->pool[best].cap <= bytes * 2 + (1u << 20)) {
        ctx->pool[best].in_use = true;
        return ctx->pool[best].p;
    }
    // allocate fresh; when memory is tight the idle buffers of the pool are released first
    void* p = nullptr;
    size_t cap = (bytes + 255) & ~size_t(255);
    hipError_t e = dev_malloc(ctx, &p, cap);
    if (e != hipSuccess) {
        set_err(ctx, FR_OUT_OF_MEMORY, "hipMalloc(%zu) failed: %s", cap, hipGetErrorString(e));
        return nullptr;
    }
    DevBuf nb;
    nb.p = p;
    nb.cap = cap;
    nb.in_use = true;
    // reuse an empty slot
    for (auto& b : ctx->pool)
        if (!b.p) {
            b = nb;
            return p;
        }
    ctx->pool.push_back(nb);
    return p;
}

void ws_put(fr_ctx* ctx, void* p)
{
    for (auto& b : ctx->pool)
        if (b.p == p) {
            b.in_use = false;
            return;
        }
}

// Pinned bounce buffer: pageable host memory is copied into it by the CPU and leaves it by DMA (a pageable hipMemcpy does
// the same through a driver-internal buffer, 64 KiB at a time and synchronously).  Transfers larger than the cap go direct.
constexpr size_t PINNED_MAX = (size_t)256 << 20;
void* pinned_get(fr_ctx* ctx, size_t bytes)
{
    if (bytes == 0 || bytes > PINNED_MAX) return nullptr;
    if (ctx->pinned_cap >= bytes) return ctx->pinned;
    if (ctx->pinned) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipHostFree(ctx->pinned);
        ctx->pinned = nullptr;
        ctx->pinned_cap = 0;
    }
    size_t cap = (size_t)1 << 20;
    while (cap < bytes) cap <<= 1;
    void* p = nullptr;
    if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    ctx->pinned = p;
    ctx->pinned_cap = cap;
    return p;
}

static bool usable_in_place(const double* p, int64_t rows, int64_t ld)
{
    return is_device_ptr(p) && ld >= (rows > 0 ? rows : 1);
}

int Staged::in(const double* src, int64_t r, int64_t c, int64_t ldsrc)
{
    rows = r;
    cols = c;
    if (r < 0 || c < 0 || ldsrc < (r > 0 ? r : 1)) return set_err(ctx, FR_SHAPE, "bad matrix shape %lld x %lld ld %lld",
                                                                   (long long)r, (long long)c, (long long)ldsrc);
    if (r == 0 || c == 0) {
        dev = nullptr;
        ld = 1;
        return FR_OK;
    }
    if (!src) return set_err(ctx, FR_INVALID_ARGUMENT, "null matrix pointer");
    if (usable_in_place(src, r, ldsrc)) {
        dev = const_cast<double*>(src);
        ld = ldsrc;
        owns = false;
        return FR_OK;
    }
    ld = round_up(r, kAlign);
    dev = (double*)ws_get(ctx, sizeof(double) * (size_t)ld * (size_t)c);
    if (!dev) return FR_OUT_OF_MEMORY;
    owns = true;
    const bool src_dev = is_device_ptr(src);
    if (!src_dev) {
        // host data: packed into the pinned bounce buffer by the CPU, then one DMA
        double* pin = (double*)pinned_get(ctx, sizeof(double) * (size_t)r * (size_t)c);
        if (pin) {
            FR_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the previous user of the bounce buffer
            for (int64_t j = 0; j < c; ++j) memcpy(pin + j * r, src + j * ldsrc, sizeof(double) * (size_t)r);
            FR_HIP(ctx, hipMemcpy2DAsync(dev, sizeof(double) * ld, pin, sizeof(double) * r, sizeof(double) * r, c,
                                         hipMemcpyHostToDevice, ctx->stream));
            return FR_OK;  // asynchronous: the caller's memory is already free, the bounce buffer is stream-ordered
        }
    }
    FR_HIP(ctx, hipMemcpy2DAsync(dev, sizeof(double) * ld, src, sizeof(double) * ldsrc, sizeof(double) * r, c,
                                 src_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
    // pageable host memory: make sure the runtime is done with the caller's buffer before we return to it
    if (!src_dev) FR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FR_OK;
}

int Staged::out(double* dst, int64_t r, int64_t c, int64_t lddst)
{
    rows = r;
    cols = c;
    if (r < 0 || c < 0 || lddst < (r > 0 ? r : 1)) return set_err(ctx, FR_SHAPE, "bad matrix shape %lld x %lld ld %lld",
                                                                   (long long)r, (long long)c, (long long)lddst);
    if (r == 0 || c == 0) {
        dev = nullptr;
        ld = 1;
        return FR_OK;
    }
    if (!dst) return set_err(ctx, FR_INVALID_ARGUMENT, "null output pointer");
    if (usable_in_place(dst, r, lddst)) {
        dev = dst;
        ld = lddst;
        owns = false;
        return FR_OK;
    }
    ld = round_up(r, kAlign);
    dev = (double*)ws_get(ctx, sizeof(double) * (size_t)ld * (size_t)c);
    if (!dev) return FR_OUT_OF_MEMORY;
    owns = true;
    host = dst;
    host_ld = lddst;
    return FR_OK;
}

int Staged::inout(double* p, int64_t r, int64_t c, int64_t ldp)
{
    FR_TRY(in(p, r, c, ldp));
    if (owns) {
        host = p;
        host_ld = ldp;
    }
    return FR_OK;
}

int Staged::commit()
{
    // A persistent solve launched since the last status check may have given up on a hand-off: the caller must learn that
    // BEFORE its memory is overwritten with a partial result (the entry point then repeats the work from the caller's intact
    // operand, solve_retry) -- and also when the destination is device memory, where there is otherwise nothing to wait for.
    const bool to_host = owns && host && rows > 0 && cols > 0 && !is_device_ptr(host);
    if (ctx->persistent_pending && to_host) {
        // one synchronisation instead of two: the result travels to the pinned bounce buffer first, the status is read, and
        // only then does the CPU copy it into the caller's memory
        double* pin = (double*)pinned_get(ctx, sizeof(double) * (size_t)rows * (size_t)cols);
        if (pin) {
            FR_HIP(ctx, hipMemcpy2DAsync(pin, sizeof(double) * rows, dev, sizeof(double) * ld, sizeof(double) * rows, cols, hipMemcpyDeviceToHost, ctx->stream));
            FR_HIP(ctx, hipStreamSynchronize(ctx->stream));
            FR_TRY(check_status_word(ctx));
            for (int64_t j = 0; j < cols; ++j) memcpy(host + j * host_ld, pin + j * rows, sizeof(double) * (size_t)rows);
            return FR_OK;
        }
    }
    if (ctx->persistent_pending) {
        FR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        FR_TRY(check_status_word(ctx));
    }
    if (!owns || !host || rows == 0 || cols == 0) return FR_OK;
    FR_HIP(ctx, hipMemcpy2DAsync(host, sizeof(double) * host_ld, dev, sizeof(double) * ld, sizeof(double) * rows, cols,
                                 is_device_ptr(host) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
    FR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return check_status_word(ctx);
}

// ---- device-side wait status -----------------------------------------------------------------------
int ensure_status_word(fr_ctx* ctx)
{
    if (ctx->dev_status) return FR_OK;
    void* h = nullptr;
    FR_HIP(ctx, hipHostMalloc(&h, 64, hipHostMallocMapped));
    memset(h, 0, 64);
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
        (void)hipHostFree(h);
        return set_err(ctx, FR_HIP_ERROR, "hipHostGetDevicePointer failed");
    }
    ctx->host_status = (unsigned*)h;
    ctx->dev_status = (unsigned*)d;
    return FR_OK;
}

int check_status_word(fr_ctx* ctx)
{
    ctx->persistent_pending = false;
    if (!ctx->host_status) return FR_OK;
    volatile unsigned* s = ctx->host_status;
    if (s[0] != 0) {
        s[0] = 0;
        ctx->solve_timeout_seen = true;
        return set_err(ctx, FR_HIP_ERROR, "a device-side wait timed out (persistent kernel hand-off); results are invalid");
    }
    return FR_OK;
}

void drain_stale_status(fr_ctx* ctx)
{
    if (!ctx->persistent_pending) return;
    (void)hipStreamSynchronize(ctx->stream);
    ctx->persistent_pending = false;
    if (!ctx->host_status) return;
    volatile unsigned* s = ctx->host_status;
    if (s[0] != 0) {
        s[0] = 0;
        ++ctx->stale_status_drops;
    }
}

// ---- profiling ------------------------------------------------------------------------------------
static hipEvent_t get_event(fr_ctx* ctx)
{
    if (!ctx->free_events.empty()) {
        hipEvent_t e = ctx->free_events.back();
        ctx->free_events.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

ProfScope::ProfScope(fr_ctx* c, int cls_, double flops, double bytes) : ctx(c), cls(cls_)
{
    if (!ctx->prof || !((ctx->prof_mask >> cls) & 1)) return;
    ctx->prof_launches[cls] += 1;
    ctx->prof_flops[cls] += flops;
    ctx->prof_bytes[cls] += bytes;
    a = get_event(ctx);
    b = get_event(ctx);
    if (a) (void)hipEventRecord(a, ctx->ls);
}

ProfScope::~ProfScope()
{
    if (!ctx->prof || !a || !b) return;
    (void)hipEventRecord(b, ctx->ls);
    ProfRec r;
    r.a = a;
    r.b = b;
    r.cls = cls;
    ctx->recs.push_back(r);
}

static void prof_collect(fr_ctx* ctx)
{
    if (ctx->recs.empty()) return;
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& r : ctx->recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) ctx->prof_ms[r.cls] += ms;
        ctx->free_events.push_back(r.a);
        ctx->free_events.push_back(r.b);
    }
    ctx->recs.clear();
}

}  // namespace fr

using namespace fr;

extern "C" {

int fr_abi_version(void) { return FR_ABI_VERSION; }

int fr_ctx_create(fr_ctx** out, int device)
{
    if (!out) return FR_INVALID_ARGUMENT;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        return FR_NO_DEVICE;
    }
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) return FR_NO_DEVICE;
    }
    if (device >= count) return FR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return FR_HIP_ERROR;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return FR_HIP_ERROR;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fprintf(stderr, "friedrich_amd: device %d is %s, this library is built for gfx950 only\n", device,
                prop.gcnArchName);
        return FR_NO_DEVICE;
    }
    fr_ctx* ctx = new fr_ctx();
    ctx->device = device;
    ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (const char* e = getenv("FRIEDRICH_AMD_TEST_FORCE_SOLVE_TIMEOUT")) ctx->test_force_timeout = e[0] == '1';
    if (const char* e = getenv("FRIEDRICH_AMD_TEST_MAX_WORKGROUPS")) ctx->test_max_wgs = atoi(e);
    if (const char* e = getenv("FRIEDRICH_AMD_SMALL_TILES")) ctx->small_tiles = atoi(e);
    if (const char* e = getenv("FRIEDRICH_AMD_TEST_COMM_HANG")) {  // "schedule,rank,nth": see comm.hip (test hook)
        long long a = -1, b = -1, c = 0;
        if (sscanf(e, "%lld,%lld,%lld", &a, &b, &c) == 3) {
            ctx->test_hang_schedule = a;
            ctx->test_hang_rank = b;
            ctx->test_hang_nth = c;
        }
    }
    if (const char* e = getenv("FRIEDRICH_AMD_COMM_TIMEOUT_MS")) {
        const long long v = atoll(e);
        if (v >= 0) ctx->comm_timeout_ms = v;
    }
    if (const char* e = getenv("FRIEDRICH_AMD_DIST_SCHEDULE")) {  // operator override of the sharded schedule (0, 1, 2) without touching the host program
        const int v = atoi(e);
        if (v >= 0 && v <= 2) ctx->dist_schedule = v;
    }
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return FR_HIP_ERROR;
    }
    ctx->own_stream = true;
    ctx->ls = ctx->stream;
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);  // hi = numerically lowest = highest priority
    // Only the panel / chain stream is high-priority.  The bulk stream of the chain-first schedule carries what is explicitly
    // NOT on the critical path (slice solves, copies, the large collectives): default priority, so that it neither competes
    // with the chain nor pre-empts the main stream's updates (round 1 measured a busy high-priority queue throttling the
    // dispatch of the others: +16 % on trailing updates).
    if (hipStreamCreateWithPriority(&ctx->stream2, hipStreamNonBlocking, hi) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->stream3, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_bulk, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_cols, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_panel, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_la, hipEventDisableTiming) != hipSuccess) {
        fr_ctx_destroy(ctx);
        return FR_HIP_ERROR;
    }
    if (hipMalloc((void**)&ctx->xcc_word, 64) != hipSuccess || hipMemset(ctx->xcc_word, 0, 64) != hipSuccess ||
        hipMalloc((void**)&ctx->claim_ring, sizeof(unsigned) * 2 * kClaimSlots) != hipSuccess ||
        hipMalloc((void**)&ctx->dyn_ring, sizeof(unsigned) * 2 * 256) != hipSuccess) {
        fr_ctx_destroy(ctx);
        return FR_HIP_ERROR;
    }
    *out = ctx;
    return FR_OK;
}

void fr_ctx_destroy(fr_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    // (every stream that may still hold records of the communicator's events, before those go)
    if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
    if (ctx->stream3) (void)hipStreamSynchronize(ctx->stream3);
    fr::comm_destroy_internal(ctx);
    for (auto& r : ctx->recs) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    for (auto e : ctx->free_events) (void)hipEventDestroy(e);
    for (auto& b : ctx->pool)
        if (b.p) (void)hipFree(b.p);
    if (ctx->trsv_gran) (void)hipFree(ctx->trsv_gran);
    if (ctx->trsmn_buf) (void)hipFree(ctx->trsmn_buf);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->readback) (void)hipHostFree(ctx->readback);
    if (ctx->xcc_word) (void)hipFree(ctx->xcc_word);
    if (ctx->cu_rank) (void)hipFree(ctx->cu_rank);
    if (ctx->host_status) (void)hipHostFree(ctx->host_status);
    if (ctx->claim_ring) (void)hipFree(ctx->claim_ring);
    if (ctx->chain_flags) (void)hipFree(ctx->chain_flags);
    if (ctx->chain_ts) (void)hipHostFree(ctx->chain_ts);
    if (ctx->dyn_ring) (void)hipFree(ctx->dyn_ring);
    if (ctx->stream3) {
        (void)hipStreamSynchronize(ctx->stream3);
        (void)hipStreamDestroy(ctx->stream3);
    }
    for (auto& kind : ctx->ev_ring)
        for (hipEvent_t e : kind)
            if (e) (void)hipEventDestroy(e);
    if (ctx->ev_bulk) (void)hipEventDestroy(ctx->ev_bulk);
    if (ctx->ev_cols) (void)hipEventDestroy(ctx->ev_cols);
    if (ctx->stream2) {
        (void)hipStreamSynchronize(ctx->stream2);
        (void)hipStreamDestroy(ctx->stream2);
    }
    if (ctx->ev_panel) (void)hipEventDestroy(ctx->ev_panel);
    if (ctx->ev_la) (void)hipEventDestroy(ctx->ev_la);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int fr_ctx_set_stream(fr_ctx* ctx, void* hip_stream)
{
    if (!ctx) return FR_INVALID_ARGUMENT;
    FR_LOCK(ctx);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    ctx->stream = (hipStream_t)hip_stream;
    ctx->ls = ctx->stream;
    ctx->own_stream = false;
    return FR_OK;
}

int fr_ctx_synchronize(fr_ctx* ctx)
{
    if (!ctx) return FR_INVALID_ARGUMENT;
    FR_LOCK(ctx);
    FR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return check_status_word(ctx);
}

const char* fr_last_error(const fr_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int fr_ctx_set_option(fr_ctx* ctx, const char* name, int64_t value)
{
    if (!ctx || !name) return FR_INVALID_ARGUMENT;
    FR_LOCK(ctx);
    if (!strcmp(name, "nb")) {
        if (value != 0 && (value < 128 || value % 128 != 0 || value > 4096))
            return set_err(ctx, FR_INVALID_ARGUMENT, "nb must be 0 (automatic) or a multiple of 128 in [128, 4096]");
        ctx->nb = value;
        return FR_OK;
    }
    if (!strcmp(name, "lookahead")) {
        ctx->lookahead = value != 0;
        return FR_OK;
    }
    if (!strcmp(name, "nb_big_rows")) {
        if (value < 0) return set_err(ctx, FR_INVALID_ARGUMENT, "nb_big_rows must be >= 0");
        ctx->nb_big_rows = value;
        return FR_OK;
    }
    if (!strcmp(name, "nb_switch_rows")) {
        if (value < 0) return set_err(ctx, FR_INVALID_ARGUMENT, "nb_switch_rows must be >= 0");
        ctx->nb_switch_rows = value;
        return FR_OK;
    }
    if (!strcmp(name, "xcd_reserve")) {
        if (value < -1 || value > 4) return set_err(ctx, FR_INVALID_ARGUMENT, "xcd_reserve must be in [-1, 4]");
        ctx->xcd_reserve = value;
        return FR_OK;
    }
    if (!strcmp(name, "xcd_reserve_big_rows")) {
        if (value < 0) return set_err(ctx, FR_INVALID_ARGUMENT, "xcd_reserve_big_rows must be >= 0");
        ctx->xcd_reserve_big_rows = value;
        return FR_OK;
    }
    if (!strcmp(name, "reserve_rows2_cu")) {
        if (value < 0) return set_err(ctx, FR_INVALID_ARGUMENT, "reserve_rows2_cu must be >= 0");
        ctx->reserve_rows2_cu = value;
        return FR_OK;
    }
    if (!strcmp(name, "reserve_rows1_cu")) {
        if (value < 0) return set_err(ctx, FR_INVALID_ARGUMENT, "reserve_rows1_cu must be >= 0");
        ctx->reserve_rows1_cu = value;
        return FR_OK;
    }
    if (!strcmp(name, "reserve_rows1") || !strcmp(name, "reserve_rows2") || !strcmp(name, "reserve_rows4")) {
        if (value < 0) return set_err(ctx, FR_INVALID_ARGUMENT, "reserve_rows* must be >= 0");
        (name[12] == '1' ? ctx->reserve_rows1 : (name[12] == '2' ? ctx->reserve_rows2 : ctx->reserve_rows4)) = value;
        return FR_OK;
    }
    if (!strcmp(name, "cu_reserve_min_rows")) {
        if (value < 0) return set_err(ctx, FR_INVALID_ARGUMENT, "cu_reserve_min_rows must be >= 0");
        ctx->cu_reserve_min_rows = value;
        return FR_OK;
    }
    if (!strcmp(name, "cu_reserve")) {
        if (value != 0 && value != 1) return set_err(ctx, FR_INVALID_ARGUMENT, "cu_reserve must be 0 or 1");
        ctx->cu_reserve = value;
        return FR_OK;
    }
    if (!strcmp(name, "panel_chain")) {
        if (value < 0 || value > 2) return set_err(ctx, FR_INVALID_ARGUMENT, "panel_chain must be 0 (launch chain), 1 (resident chain where measured) or 2 (wherever the shape fits)");
        ctx->panel_chain = value;
        return FR_OK;
    }
    if (!strcmp(name, "k4_flat")) {
        if (value < -1 || value > 1) return set_err(ctx, FR_INVALID_ARGUMENT, "k4_flat must be -1 (automatic), 0 or 1");
        ctx->k4_flat = value;
        return FR_OK;
    }
    if (!strcmp(name, "dist_schedule")) {
        if (value < 0 || value > 2) return set_err(ctx, FR_INVALID_ARGUMENT, "dist_schedule must be 0 (broadcast), 1 (split) or 2 (diagonal chain first)");
        ctx->dist_schedule = value;
        return FR_OK;
    }
    if (!strcmp(name, "comm_timeout_ms")) {
        if (value < 0) return set_err(ctx, FR_INVALID_ARGUMENT, "comm_timeout_ms must be >= 0 (0: wait for ever)");
        ctx->comm_timeout_ms = value;
        return FR_OK;
    }
    if (!strcmp(name, "splitk")) {
        ctx->splitk = value != 0;
        return FR_OK;
    }
    if (!strcmp(name, "narrow_max")) {
        if (value < 0 || value > 4096) return set_err(ctx, FR_INVALID_ARGUMENT, "narrow_max must be in [0, 4096]");
        ctx->narrow_max = value;
        return FR_OK;
    }
    if (!strcmp(name, "narrow_batched_max")) {
        if (value < -1 || value > 65536) return set_err(ctx, FR_INVALID_ARGUMENT, "narrow_batched_max must be in [-1, 65536]");
        ctx->narrow_batched_max = value;
        return FR_OK;
    }
    if (!strcmp(name, "bigleaf_min")) {
        if (value < -1) return set_err(ctx, FR_INVALID_ARGUMENT, "bigleaf_min must be >= -1");
        ctx->bigleaf_min = value;
        return FR_OK;
    }
    if (!strcmp(name, "bigleaf_max")) {
        if (value < -1) return set_err(ctx, FR_INVALID_ARGUMENT, "bigleaf_max must be >= -1");
        ctx->bigleaf_max = value;
        return FR_OK;
    }
    if (!strcmp(name, "leaf512")) {
        ctx->leaf512 = value != 0;
        return FR_OK;
    }
    if (!strcmp(name, "trsv")) {
        ctx->trsv = value != 0;
        return FR_OK;
    }
    if (!strcmp(name, "grad_shard_min")) {
        if (value < 0) return set_err(ctx, FR_INVALID_ARGUMENT, "grad_shard_min must be >= 0");
        ctx->grad_shard_min = value;
        return FR_OK;
    }
    if (!strcmp(name, "tri_inverse")) {
        ctx->tri_inverse = value != 0;
        return FR_OK;
    }
    if (!strcmp(name, "predict_assoc")) {
        if (value != 0 && value != 1) return set_err(ctx, FR_INVALID_ARGUMENT, "predict_assoc must be 0 or 1");
        ctx->predict_assoc = value;
        return FR_OK;
    }
    if (!strcmp(name, "refine")) {
        if (value < -1 || value > 1) return set_err(ctx, FR_INVALID_ARGUMENT, "refine must be -1 (automatic), 0 or 1");
        ctx->refine = value;
        return FR_OK;
    }
    if (!strcmp(name, "refine_threshold")) {
        if (value < 1) return set_err(ctx, FR_INVALID_ARGUMENT, "refine_threshold must be >= 1");
        ctx->refine_threshold = (double)value;
        return FR_OK;
    }
    return set_err(ctx, FR_INVALID_ARGUMENT, "unknown option %s", name);
}

int fr_ctx_get_counter(fr_ctx* ctx, const char* name, int64_t* out)
{
    if (!ctx || !name || !out) return FR_INVALID_ARGUMENT;
    FR_LOCK(ctx);
    if (!strcmp(name, "solve_retries")) {
        *out = ctx->solve_retries;
        return FR_OK;
    }
    if (!strncmp(name, "chain_ts:", 9)) {  // developer stamps of the last resident panel-chain launch (FRIEDRICH_AMD_CHAIN_TS=1)
        const int i = atoi(name + 9);
        if (!ctx->chain_ts || i < 0 || i >= 128) return set_err(ctx, FR_INVALID_ARGUMENT, "no such stamp");
        *out = (int64_t)((volatile unsigned long long*)ctx->chain_ts)[i];
        return FR_OK;
    }
    if (!strcmp(name, "panel_chain_launches")) {
        *out = ctx->panel_chain_launches;
        return FR_OK;
    }
    if (!strcmp(name, "panel_chain_fallbacks")) {
        *out = ctx->panel_chain_fallbacks;
        return FR_OK;
    }
    if (!strcmp(name, "stale_status_drops")) {
        *out = ctx->stale_status_drops;
        return FR_OK;
    }
    if (!strcmp(name, "comm_timeouts")) {
        *out = ctx->comm_timeouts;
        return FR_OK;
    }
    if (!strcmp(name, "pool_bytes")) {
        size_t b = 0;
        for (auto& d : ctx->pool) b += d.cap;
        *out = (int64_t)b;
        return FR_OK;
    }
    return set_err(ctx, FR_INVALID_ARGUMENT, "unknown counter %s", name);
}

int fr_inputs_to_device(fr_ctx* ctx, int layout, const void* data, int64_t n, int64_t d, int64_t stride, double** out_dev,
                        int64_t* out_ld)
{
    if (!ctx || !out_dev || !out_ld) return FR_INVALID_ARGUMENT;
    FR_LOCK(ctx);
    *out_dev = nullptr;
    *out_ld = 0;
    FR_HIP(ctx, hipSetDevice(ctx->device));
    if (n < 0 || d < 0) return set_err(ctx, FR_SHAPE, "negative input shape");
    if (layout == FR_LAYOUT_ROWPTRS && n > 0 && d == 0) return set_err(ctx, FR_SHAPE, "samples without features");
    if (layout == FR_LAYOUT_COLMAJOR && stride < (n > 0 ? n : 1)) return set_err(ctx, FR_SHAPE, "column stride below the row count");
    if (layout == FR_LAYOUT_ROWMAJOR && stride < (d > 0 ? d : 1)) return set_err(ctx, FR_SHAPE, "row stride below the feature count");
    if (layout < 0 || layout > 2) return set_err(ctx, FR_INVALID_ARGUMENT, "unknown input layout");
    if (n > 0 && d > 0 && !data) return set_err(ctx, FR_INVALID_ARGUMENT, "null input data");
    const int64_t ld = round_up(n > 0 ? n : 1, kAlign);
    double* out = nullptr;
    FR_HIP(ctx, dev_malloc(ctx, (void**)&out, sizeof(double) * (size_t)ld * (size_t)(d > 0 ? d : 1)));
    int st = FR_OK;
    if (n > 0 && d > 0) {
        const bool dev_src = layout != FR_LAYOUT_ROWPTRS && is_device_ptr(data);
        if (layout == FR_LAYOUT_COLMAJOR) {
            const double* src = (const double*)data;
            if (dev_src) {
                if (hipMemcpy2DAsync(out, sizeof(double) * ld, src, sizeof(double) * stride, sizeof(double) * n, d,
                                     hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)
                    st = set_err(ctx, FR_HIP_ERROR, "device copy failed");
            } else {
                double* pin = (double*)pinned_get(ctx, sizeof(double) * (size_t)n * (size_t)d);
                (void)hipStreamSynchronize(ctx->stream);
                if (pin) {
                    for (int64_t j = 0; j < d; ++j) memcpy(pin + j * n, src + j * stride, sizeof(double) * (size_t)n);
                    src = pin;
                    stride = n;
                }
                if (hipMemcpy2DAsync(out, sizeof(double) * ld, src, sizeof(double) * stride, sizeof(double) * n, d,
                                     hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
                    st = set_err(ctx, FR_HIP_ERROR, "upload failed");
                if (!pin) (void)hipStreamSynchronize(ctx->stream);
            }
        } else {
            // row-major: the samples reach the device as they are (d contiguous values each) and are transposed there
            WsGuard tmpg(ctx);
            const double* rm = (const double*)data;  // row-major n x d with row stride `stride` (device) ...
            int64_t rstride = stride;
            if (!dev_src) {
                double* tmp = tmpg.get(sizeof(double) * (size_t)n * (size_t)d);
                if (!tmp) {
                    (void)hipFree(out);
                    return FR_OUT_OF_MEMORY;
                }
                double* pin = (double*)pinned_get(ctx, sizeof(double) * (size_t)n * (size_t)d);
                (void)hipStreamSynchronize(ctx->stream);
                std::vector<double> pageable;
                if (!pin) {
                    pageable.resize((size_t)n * (size_t)d);
                    pin = pageable.data();
                }
                if (layout == FR_LAYOUT_ROWPTRS) {
                    const double* const* rows = (const double* const*)data;
                    for (int64_t r = 0; r < n; ++r) {
                        if (!rows[r]) {
                            (void)hipFree(out);
                            return set_err(ctx, FR_INVALID_ARGUMENT, "null sample pointer (row %lld)", (long long)r);
                        }
                        memcpy(pin + r * d, rows[r], sizeof(double) * (size_t)d);
                    }
                } else if (stride == d) {
                    memcpy(pin, data, sizeof(double) * (size_t)n * (size_t)d);
                } else {
                    for (int64_t r = 0; r < n; ++r) memcpy(pin + r * d, (const double*)data + r * stride, sizeof(double) * (size_t)d);
                }
                if (hipMemcpyAsync(tmp, pin, sizeof(double) * (size_t)n * (size_t)d, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
                    st = set_err(ctx, FR_HIP_ERROR, "upload failed");
                if (!pageable.empty()) (void)hipStreamSynchronize(ctx->stream);
                rm = tmp;
                rstride = d;
            }
            // a row-major n x d matrix is a column-major d x n one with leading dimension rstride: transpose it
            if (st == FR_OK) st = launch_transpose(ctx, rm, d, n, rstride, out, ld);
            if (st == FR_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = set_err(ctx, FR_HIP_ERROR, "staging failed");
        }
    }
    if (st != FR_OK) {
        (void)hipFree(out);
        return st;
    }
    *out_dev = out;
    *out_ld = ld;
    return FR_OK;
}

void fr_device_free(fr_ctx* ctx, double* dev)
{
    if (!ctx || !dev) return;
    FR_LOCK(ctx);
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(dev);
}

int fr_ctx_profile_enable(fr_ctx* ctx, int enable)
{
    if (!ctx) return FR_INVALID_ARGUMENT;
    FR_LOCK(ctx);
    prof_collect(ctx);
    ctx->prof = enable != 0;
    // enable == 1: every class; otherwise bit (cls + 1) selects class cls (bench.py times only the SYRK class so
    // the event records do not perturb the timed region)
    ctx->prof_mask = (enable == 1) ? ~0u : ((unsigned)enable >> 1);
    return FR_OK;
}

int fr_ctx_profile_reset(fr_ctx* ctx)
{
    if (!ctx) return FR_INVALID_ARGUMENT;
    FR_LOCK(ctx);
    prof_collect(ctx);
    for (int i = 0; i < FR_PROF_COUNT; ++i) {
        ctx->prof_ms[i] = 0;
        ctx->prof_launches[i] = 0;
        ctx->prof_flops[i] = 0;
        ctx->prof_bytes[i] = 0;
    }
    return FR_OK;
}

int fr_ctx_profile_get(fr_ctx* ctx, int cls, double* ms, int64_t* launches, double* flops, double* bytes)
{
    if (!ctx || cls < 0 || cls >= FR_PROF_COUNT) return FR_INVALID_ARGUMENT;
    FR_LOCK(ctx);
    prof_collect(ctx);
    if (ms) *ms = ctx->prof_ms[cls];
    if (launches) *launches = ctx->prof_launches[cls];
    if (flops) *flops = ctx->prof_flops[cls];
    if (bytes) *bytes = ctx->prof_bytes[cls];
    return FR_OK;
}

}  // extern "C"
