// fr_internal.hpp -- shared internals of libfriedrich_amd (gfx950 only; no CPU fallback).
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "friedrich_amd.h"
#include <mutex>

namespace fr {
// roctx range around an entry point (rocprofv3 --marker-trace shows the fr_* calls above their kernels).  Off unless the
// process sets FRIEDRICH_AMD_ROCTX=1: the marker library (librocprofiler-sdk-roctx / libroctx64) is dlopen'ed on first use.
struct TraceScope {
    bool on;
    explicit TraceScope(const char* name);
    ~TraceScope();
};
}  // namespace fr

#define FR_LOCK(ctxptr)                                                       \
    std::lock_guard<std::recursive_mutex> fr_lock_guard__((ctxptr)->mu);      \
    fr::TraceScope fr_trace_scope__(__func__)

namespace fr {

constexpr int64_t kAlign = 64;  // row padding (elements) of every internal column-major buffer

inline int64_t round_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }
// Leading dimensions stay plain multiples of kAlign.  Padding them away from powers of two was measured: an isolated
// GEMM at N = 16384 gains 10 % (scripts/gemm_ld_probe.py) and the Gram assembly 15 % (gram_ld_probe.py), but the fit as
// a whole is unchanged at every size tried (scripts/ld_ab.py, option "ld_pad", same process), so it is not applied.

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    bool in_use = false;
};

struct ProfRec {
    hipEvent_t a, b;
    int cls;
};

}  // namespace fr

struct fr_ctx {
    // Serialises the entry points of one context: friedrich's GaussianProcess is Send + Sync, so concurrent &self calls
    // (several threads predicting from one model) are legal and must stay correct; they share this context's streams
    // and workspaces, so they take turns.  Recursive: entry points call each other.
    std::recursive_mutex mu;
    int device = 0;
    hipStream_t stream = nullptr;   // main stream (memcpys, host synchronisation)
    bool own_stream = true;
    hipStream_t ls = nullptr;       // stream the kernel launchers enqueue on (== stream except inside look-ahead)
    hipStream_t stream2 = nullptr;  // high-priority panel stream of the look-ahead Cholesky
    hipStream_t stream3 = nullptr;  // sharded factorisation: the bulk rows of a panel (solves + all-gather) off the diagonal chain
    hipEvent_t ev_panel = nullptr, ev_la = nullptr, ev_cols = nullptr, ev_bulk = nullptr;
    int64_t cols_final_at = -1;     // factor_panel records ev_cols when the columns up to this one are final (look-ahead pipeline)
    std::string err;
    // grow-only workspace pool (stream-ordered reuse inside one context)
    std::vector<fr::DevBuf> pool;
    // ---- options (fr_ctx_set_option) ----
    int64_t nb = 0;             // outer Cholesky block; 0 = chosen from the matrix size (pick_nb)
    int64_t lookahead = 1;      // the next panel is factored on the panel stream under the trailing update
    int64_t nb_big_rows = 22528;     // automatic nb = 1024 only: panels of 2048 columns while more than this many rows remain (0: never)
    int64_t nb_switch_rows = 16384;  // automatic nb = 1024: panels of 512 columns once at most this many rows remain (0: never)
    // XCD reservation (gemm_f64.hip): the main stream's GEMM launches of a factorisation leave the first `reserve_now` XCDs
    // (counted from the one the diagonal-block kernels run on) to the panel stream -- their workgroups there exit at once
    int64_t xcd_reserve = -1;   // -1: chosen by the factorisation (single GPU, nb <= 512: 1 XCD below 16384 rows, 2 below 8192); 0: never; 1 .. 4: always
    int64_t dist_schedule = 1;  // sharded factorisation, panel step: 0 owner solves the whole panel + one broadcast; 1 (default) diagonal
                                // block broadcast, rows scattered / solved per rank / all-gathered; 2 as 1 with the diagonal chain running
                                // ahead (diagonal block sent to the next owner first, bulk rows on their own stream: chol.hip) -- opt-in
                                // until it has run over real RCCL on more than one GPU (bench.py preflights it and falls back)
    int64_t comm_timeout_ms = 120000;  // sharded operations: how long a wait for a stream holding collectives / a blocked RCCL call may
                                       // last before the communicators are aborted (comm.hip: comm_stream_sync, watchdog); 0: for ever
    int64_t splitk = 1;         // GEMMs with few result tiles and a deep contraction are cut along K (gemm_f64.hip)
    int64_t narrow_max = 16;    // solves with at most this many right-hand sides take the memory-bound kernels (chol.hip)
    int64_t narrow_batched_max = -1;  // right-hand sides up to which the persistent solve runs in column groups of 16 (trsm_narrow.hip); -1: chosen from n and m (chol.hip)
    int64_t leaf512 = 1;        // wide triangular solves end in 512-row leaves (explicit 512-block inverses); 0: 128-row leaves
    int64_t bigleaf_min = -1;   // ... and at least this many (-1: by measurement, chol.hip: use_big_leaves)
    int64_t bigleaf_max = -1;   // solves with at most this many right-hand sides (and >= 4096 rows) run left-looking over 2048-row blocks
                                // with explicit 2048-block inverses (chol.hip: trsm_big); -1: 4096; 0: never
    int64_t trsv = 1;           // solves with few right-hand sides as one persistent launch per direction (trsv.hip, trsm_narrow.hip)
    int64_t grad_shard_min = 4096;  // sharded contexts: gradient terms of factors with at least this many rows are split over the ranks (grad.hip)
    int64_t tri_inverse = 1;    // gradient terms: L^-1 and W^T W skip the structural zeros (chol_tri_inverse); 0: dense products
    int64_t predict_assoc = 0;  // 0: (K^-1 K*)^T y as the reference, 1: K*^T (K^-1 y)
    // Products with explicit inverse blocks lose a factor cond(L_bb) of backward accuracy against substitution.  refine:
    // -1 (default) automatic -- a factorisation whose diagonal blocks turn out ill-conditioned (estimate from the
    // diagonal-block kernel above refine_threshold) is repeated with one step of iterative refinement behind every such
    // product, and the handle keeps refining (factor, add_rows, solves); 0 never; 1 always
    int64_t refine = -1;
    double refine_threshold = 30.0;
    int64_t small_tiles = 64;  // products of at most this many 128 x 128 tiles use 32-row tiles (gemm_f64.hip; FRIEDRICH_AMD_SMALL_TILES overrides for A/B runs)
    // ---- state ----
    bool potf2_lds_set = false;  // dynamic-LDS attributes applied on this device (per context = per device)
    bool trsv_lds_set = false;
    bool trsmn_lds_set = false;
    bool prior_lds_set = false;
    bool rs16_lds_set = false;
    bool chain_lds_set = false;
    // resident panel chain (potf2.hip: panel_chain_kernel): progress flags of the launches in flight (a ring of blocks, values carry
    // the launch's epoch: nothing is zeroed between launches)
    int* chain_flags = nullptr;
    int64_t chain_epoch = 0;
    unsigned long long* chain_ts = nullptr;  // developer stamps of the last resident-chain launch (pinned host memory; FRIEDRICH_AMD_CHAIN_TS)
    int64_t panel_chain = 2;       // option: panels of 256 .. 512 columns factored by ONE resident launch (potf2.hip); 2 (default): wherever the shape fits, 1: only where the launch has the chip to itself (no second stream, sharded chain), 0: the launch chain
    int64_t panel_chain_launches = 0, panel_chain_fallbacks = 0;  // counters (fr_ctx_get_counter)
    // profiling
    bool prof = false;
    unsigned prof_mask = ~0u;
    std::vector<fr::ProfRec> recs;
    std::vector<hipEvent_t> free_events;
    double prof_ms[FR_PROF_COUNT] = {0};
    int64_t prof_launches[FR_PROF_COUNT] = {0};
    double prof_flops[FR_PROF_COUNT] = {0};
    double prof_bytes[FR_PROF_COUNT] = {0};
    int64_t xcd_reserve_big_rows = 0;  // experiment (scripts/headline_ab.py): with panels wider than 512 columns, one XCD is set aside while at most this many rows remain (0: never -- the measured default)
    int64_t reserve_rows1 = 16384, reserve_rows2 = 8192, reserve_rows4 = 4096;  // automatic reservation tiers (nb <= 512): 1 / 2 / 4 units while at most this many rows remain
    int64_t reserve_rows1_cu = 12288;    // ... the one-unit tier when the reservation is by CUs (N = 16384 / 20480 / 32768: -1.2 / -2.0 / -0.8 % against 16384)
    int64_t reserve_rows2_cu = 6144;     // ... the two-unit tier when the reservation is by CUs (measured: scripts/optset_ab.py)
    int64_t cu_reserve_min_rows = 4096;  // ... while more than this many rows remain (below, the chain's products are too small to saturate anything and gain more from staying inside one or two XCDs' L2)
    bool reserve_by_cu_now = false;  // state: the reservation in force is carried out by CUs
    int64_t cu_reserve = 1;        // reservation by CUs (R of every shader engine) instead of by XCDs: resident trailing-update workgroups, no idle panel workgroups (gemm_f64.hip)
    unsigned char* cu_rank = nullptr;  // device: [xcc][se][cu_id] -> rank of that CU among the active CUs of its shader engine (255: absent); built on first use
    int cu_rank_state = 0;         // 0: not built, 1: usable, -1: the probe did not see a regular chip (CU-level reservation off)
    int64_t k4_flat = -1;          // diagonal-block kernel: -1 = flat variant wherever the kernel has its CU to itself (potf2.hip), 0 = never, 1 = every full block
    bool k4_alone = false;         // the running factorisation has no second stream: nothing shares the diagonal-block kernel's CU
    int reserve_now = 0;           // XCDs reserved right now (set by the factorisation around the launches it applies to)
    unsigned panel_epoch = 0;      // number of the panel being factored on the panel stream (see claim_item)
    unsigned* xcc_word = nullptr;    // device: [0] = 1 + XCC_ID of the XCD the diagonal-block kernels run on (0: unknown), [1] = last panel whose chain is finished
    unsigned* dyn_ring = nullptr;    // device: counter pairs of `dynamic` launches outside a factorisation, zeroed one by one
    int64_t dyn_next = 0;
    unsigned* claim_ring = nullptr;  // device: {tile counter, retire counter} per reserved launch of a factorisation
    int64_t claim_next = 0;
    bool refine_now = false;      // state of the running operation (set under the context lock)
    double* cur_cest = nullptr;   // where the diagonal-block kernel of the running factorisation puts its estimates
    // device-side waits report a timeout here (host-mapped, so the host can read it after any synchronisation)
    unsigned* host_status = nullptr;
    unsigned* dev_status = nullptr;
    bool solve_timeout_seen = false;  // a persistent solve timed out on this context: the entry point re-runs on the recursive path
    bool persistent_pending = false;  // a persistent solve was launched since the last check_status_word
    int64_t solve_retries = 0;        // how often that happened (fr_ctx_get_counter)
    int64_t stale_status_drops = 0;   // time-outs left behind by an entry point that returned early, dropped by the next one (drain_stale_status)
    int test_max_wgs = 0;             // FRIEDRICH_AMD_TEST_MAX_WORKGROUPS at context creation: cap on the grid of the persistent solves (tests: many blocks per workgroup)
    bool test_force_timeout = false;  // FRIEDRICH_AMD_TEST_FORCE_SOLVE_TIMEOUT=1 at context creation: every persistent solve reports a timeout (tests of the retry)
    int num_cus = 256;
    // hand-off granules of the persistent solves (grow-only)
    void* trsv_gran = nullptr;
    size_t trsv_gran_cap = 0;
    // pinned bounce buffer of the host <-> device staging (grow-only)
    void* pinned = nullptr;
    size_t pinned_cap = 0;
    void* readback = nullptr;  // 4 KiB of pinned host memory for small results read back with an operation's own synchronisation
    // hand-off payload + flags of the multi-column persistent solves (trsm_narrow.hip), grow-only
    void* trsmn_buf = nullptr;
    size_t trsmn_buf_cap = 0;
    // RCCL
    void* comm = nullptr;   // ncclComm_t
    void* comm2 = nullptr;  // second communicator over the same ranks (bulk stream of the chain-first schedule)
    hipEvent_t ev_ring[5][4] = {};  // chain-first schedule: head / message / bulk / first look-ahead tile / nearest column, by panel % 4 (created on first use)
    int64_t* agree_buf = nullptr;  // 1 + world slots of the status agreement (comm_agree)
    void* local = nullptr;  // in-process ("local") communicator: ranks are host threads sharing one device
    void* watch = nullptr;  // watchdog thread of the RCCL communicators (comm.hip: CommWatch)
    bool comm_lost = false; // the communicators were aborted (time-out, failed peer): every collective fails until fr_ctx_comm_finalize
    int64_t comm_timeouts = 0;  // how often a wait ran out (fr_ctx_get_counter)
    // FRIEDRICH_AMD_TEST_COMM_HANG = "schedule,rank,nth" at context creation: that rank skips its nth collective under that schedule
    int64_t test_hang_schedule = -1, test_hang_rank = -1, test_hang_nth = 0, test_comm_calls = 0;
    int rank = 0;
    int world = 1;
};

// Device-resident factor.  A is capacity x capacity (ld = ld_a) column-major; its lower triangle holds L.
// dinv holds the explicit inverses of the nb x nb diagonal blocks of L (block b at dinv + b*nb*nb, ld nb),
// produced by K4/K5 during the factorisation and consumed by every triangular solve.
struct fr_chol {
    fr_ctx* ctx = nullptr;
    int64_t n = 0;         // logical size
    int64_t capacity = 0;  // row/col capacity of A (and row capacity of X)
    int64_t ld_a = 0;
    int64_t d = 0;  // feature count (0 for fr_chol_from_matrix)
    int64_t ld_x = 0;
    int64_t nb = 256;
    // the handle came out of a COLLECTIVE factorisation (fr_chol_from_inputs / fr_chol_refactor on a context with a communicator):
    // every rank holds it, so operations on it may themselves be collective (fr_grad_terms from grad_shard_min rows on).  A handle
    // that exists on one rank only -- fr_chol_from_matrix, fr_chol_upload_l -- never takes part in a collective (advisor, round 5).
    bool collective = false;
    double* A = nullptr;
    double* X = nullptr;     // capacity x d training inputs (EMatrix mirror)
    double* dinv = nullptr;  // ceil(capacity/128) explicit inverses of the 128 x 128 diagonal blocks
    // explicit inverses of the 512 x 512 diagonal blocks (leaves of the wide triangular solves), built on demand from
    // dinv; inv512_rows = rows covered by valid blocks (0 after every change of the factor)
    double* inv512 = nullptr;
    int64_t inv512_cap = 0;  // blocks allocated
    int64_t inv512_rows = 0;
    // explicit inverses of the 2048 x 2048 diagonal blocks (leaves of the solves with a few hundred right-hand sides: chol.hip,
    // ensure_invbig), built on demand from inv512 by two more block levels; invbig_rows = rows covered by valid blocks
    double* invbig = nullptr;
    int64_t invbig_cap = 0;  // blocks allocated
    int64_t invbig_rows = 0;
    int64_t narrow_solves = 0;  // solves with fewer than 192 right-hand sides since the factor last changed (use_big_leaves)
    int64_t* info = nullptr;  // device: [0] = 1 + first failing column (0: none), [1] = n_subst,
                              //         [2] = 1 if a zero diagonal was seen, [3..] substituted columns
    int64_t info_cap = 0;
    // transposed copy for the backward persistent solves with several right-hand sides (trsm_narrow.hip): the off-diagonal
    // blocks of L^T live in the strict upper triangle of A, the transposed inverse blocks in dinvt; valid for generation ut_gen
    double* dinvt = nullptr;
    int64_t dinvt_cap = 0;
    uint64_t ut_gen = 0;
    int64_t ut_n = -1;  // (rows the copy was built for: a generation can see two row counts, fr_chol_add_rows)
    // K9's chain products M_b = W_b L[b, b - 1] (forward, [0]) and W_b^T L[b + 1, b]^T (backward, [1]), one 128 x 128 block per
    // block row, valid for generation mchain_gen (trsm_narrow.hip: ensure_chain_products)
    // ([2], [3]: the same with the tile two blocks from the diagonal -- the wide column-group kernel); valid for (mchain_gen, mchain_n)
    double* mchain[4] = {nullptr, nullptr, nullptr, nullptr};
    int64_t mchain_cap[4] = {0, 0, 0, 0};
    uint64_t mchain_gen[4] = {0, 0, 0, 0};
    int64_t mchain_n[4] = {-1, -1, -1, -1};
    // conditioning estimates of the 128 x 128 diagonal blocks (device, one double per block), their maximum after the
    // last factorisation, and whether this handle applies iterative refinement (fr_ctx::refine)
    double* cest = nullptr;
    double max_cest = 0.0;
    bool refine = false;
    // cached alpha = K^-1 y (todo.md:10): the residual training outputs handed over with
    // fr_chol_set_targets and the solve they imply.  `gen` counts the changes of the factor (refactor, add_rows, upload);
    // alpha is recomputed lazily when its generation is stale, the targets must be handed over again when n changed.
    double* yt = nullptr;     // capacity doubles: targets
    double* alpha = nullptr;  // capacity doubles: K^-1 yt
    int64_t targets_n = -1;   // rows the targets were set for (-1: never)
    int64_t targets_cap = 0;
    uint64_t gen = 1, alpha_gen = 0;
    // result of the zero-diagonal check of the checked solves (mod.rs:203, 263, 345), valid for generation diag_gen: the factor does
    // not change between two predicts, so the check (a launch, a read-back, a synchronisation) runs once per factor
    uint64_t diag_gen = 0;
    bool diag_zero = false;
    // host mirror of info after the last factorisation
    int64_t fail_col = -1;
    int64_t n_subst = 0;
    std::vector<int64_t> subst;
};

namespace fr {

// ---- error handling -------------------------------------------------------------------------------
int set_err(fr_ctx* ctx, int status, const char* fmt, ...);

#define FR_HIP(ctx, call)                                                                              \
    do {                                                                                               \
        hipError_t e__ = (call);                                                                       \
        if (e__ != hipSuccess)                                                                         \
            return fr::set_err((ctx), FR_HIP_ERROR, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), \
                               __FILE__, __LINE__);                                                    \
    } while (0)

#define FR_TRY(call)                 \
    do {                             \
        int st__ = (call);           \
        if (st__ != FR_OK) return st__; \
    } while (0)

// ---- memory ---------------------------------------------------------------------------------------
bool is_device_ptr(const void* p);
// workspace: returns nullptr on failure (error recorded in ctx)
void* ws_get(fr_ctx* ctx, size_t bytes);
void ws_put(fr_ctx* ctx, void* p);
void* pinned_get(fr_ctx* ctx, size_t bytes);                    // the context's pinned bounce buffer, at least `bytes` (nullptr: none)
size_t ws_trim(fr_ctx* ctx);                                   // frees the idle pool buffers, returns the bytes released
hipError_t dev_malloc(fr_ctx* ctx, void** p, size_t bytes);    // hipMalloc, retried once after ws_trim

struct WsGuard {
    fr_ctx* ctx;
    void* p = nullptr;
    WsGuard(fr_ctx* c) : ctx(c) {}
    WsGuard(const WsGuard&) = delete;
    ~WsGuard()
    {
        if (p) ws_put(ctx, p);
    }
    double* get(size_t bytes)
    {
        if (p) ws_put(ctx, p);
        p = ws_get(ctx, bytes);
        return (double*)p;
    }
};

// A rows x cols column-major matrix made available on the device.  Host data are copied into an internal
// workspace; device data are used in place (zero copy).
struct Staged {
    fr_ctx* ctx;
    double* dev = nullptr;
    int64_t ld = 0;
    // copy-back bookkeeping
    double* host = nullptr;
    int64_t host_ld = 0;
    int64_t rows = 0, cols = 0;
    bool owns = false;
    Staged(fr_ctx* c) : ctx(c) {}
    Staged(const Staged&) = delete;
    ~Staged()
    {
        if (owns && dev) ws_put(ctx, dev);
    }
    // input: copy host->device if needed
    int in(const double* src, int64_t rows, int64_t cols, int64_t ld);
    // output: allocate a device image if dst is a host pointer (contents undefined)
    int out(double* dst, int64_t rows, int64_t cols, int64_t ld);
    // in/out
    int inout(double* p, int64_t rows, int64_t cols, int64_t ld);
    // copy the device image back to the host destination (no-op for device destinations); synchronises
    int commit();
};

// ---- profiling ------------------------------------------------------------------------------------
struct ProfScope {
    fr_ctx* ctx;
    hipEvent_t a = nullptr, b = nullptr;
    int cls;
    ProfScope(fr_ctx* c, int cls_, double flops, double bytes);
    ~ProfScope();
};

// ---- kernels (launchers; all enqueue on ctx->stream) ------------------------------------------------
// K1: Gram assembly
int launch_gram_cross(fr_ctx* ctx, const fr_kprog& prog, const double* A, int64_t n1, int64_t lda, const double* B,
                      int64_t n2, int64_t ldb, int64_t d, double* out, int64_t ldo);
int launch_gram_cross_dot(fr_ctx* ctx, const fr_kprog& prog, const double* A, int64_t n1, int64_t lda, const double* B,
                          int64_t n2, int64_t ldb, int64_t d, const double* v, double* out);
// lower triangle (full 128x128 diagonal tiles) + noise^2 on the diagonal, rows/cols [r0, n) x [c0, n)
// own_world > 1: only the block columns (width own_nb) owned by own_rank are assembled
int launch_gram_sym(fr_ctx* ctx, const fr_kprog& prog, const double* X, int64_t n, int64_t ldx, int64_t d,
                    double noise2, double* out, int64_t ldo, int own_world = 1, int own_rank = 0, int64_t own_nb = 1);
// out[i] = k(x_i, x_i) (+ add)
int launch_gram_diag(fr_ctx* ctx, const fr_kprog& prog, const double* X, int64_t n, int64_t ldx, int64_t d,
                     double add, double* out);
// out[j] = k(xq_j, xq_j) - U[:, j] . V[:, j]  (epilogue of predict_variance / predict_mean_variance, one launch)
int launch_variance_epilogue(fr_ctx* ctx, const fr_kprog& prog, const double* Xq, int64_t m, int64_t ldq, int64_t d, const double* U,
                             int64_t ldu, const double* V, int64_t ldv, int64_t n, double* out);
int launch_pairwise_distance_sum(fr_ctx* ctx, const double* X, int64_t n, int64_t ldx, int64_t d, double* out_dev);
int kprog_check(fr_ctx* ctx, const fr_kprog* p);

// K5/K6: FP64 MFMA GEMM.  D = alpha * op(A) op(B) + beta * Cin  (D may alias Cin)
//   a_kmajor = false: element (m,k) of op(A) is A[m + k*lda]  ("N")      true: A[k + m*lda]  ("T")
//   b_kmajor = true : element (k,n) of op(B) is B[k + n*ldb]  ("N")      false: B[n + k*ldb] ("T")
//   lower = true: only tiles intersecting the lower triangle of the M x M result are computed (SYRK use)
constexpr int64_t kClaimSlots = 4096;
constexpr int64_t kChainRing = 64;  // flag blocks of the resident panel chain, by launch epoch

struct GemmDesc {
    int64_t M, N, K;
    const double* A;
    int64_t lda;
    bool a_kmajor;
    const double* B;
    int64_t ldb;
    bool b_kmajor;
    const double* Cin;
    int64_t ldcin;
    double* D;
    int64_t ldd;
    double alpha, beta;
    bool lower;
    int prof_cls;
    // multi-GPU column-ownership filter (see gemm_f64.hip); defaults: disabled
    int own_world = 1, own_rank = 0;
    int64_t own_nb = 1, own_col0 = 0;
    // batched launch: `batch` independent problems of the same shape, operands `batch_*` elements apart
    int64_t batch = 1, batch_a = 0, batch_b = 0, batch_c = 0, batch_d = 0;
    bool whole_chip = false;  // ignore the XCD reservation for this launch (a panel-stream product while no diagonal-block kernel runs)
    int tri = 0;  // triangular operands: see GemmArgs::tri (gemm_tile.hpp)
    // tiles claimed in dispatch order instead of dealt per XCD: for launches whose tiles differ in length (tri), where equal
    // tile counts per XCD are unequal work
    bool dynamic = false;
    bool tri_splitk = false;   // a triangular-operand product may be cut along K like any other (slices of structural zeros retire at once)
    int64_t kslice = 0, k_total = 0;  // (set by the split-K path: slice length and whole contraction of the batched launch)
    bool mirror = false;       // a workgroup takes row tile i and then row tile (last - i): equal work per workgroup with a triangular left operand
    bool force_small = false;  // 32-row tiles whatever the tile count, triangular operands included (the big solve leaves: chol.hip)
};
int launch_gemm(fr_ctx* ctx, const GemmDesc& g);
// S (rows x kb, ld lds_) <- S L^-T against a factored kb x kb diagonal block and its 128-block inverses: one launch (gemm_f64.hip)
int launch_rows_solve(fr_ctx* ctx, double* S, int64_t lds_, int64_t rows, const double* L, int64_t ldl, int64_t kb, const double* dinv);
int launch_release_xcds(fr_ctx* ctx, unsigned epoch);  // on ctx->ls: the chain of panel `epoch` is finished
bool cu_table_ready(fr_ctx* ctx);     // option cu_reserve is set and the chip looks as expected (builds the CU rank table on first use)
bool cu_reserve_active(fr_ctx* ctx);  // the reservation in force is carried out by CUs

// K4: factor one diagonal block (nbk <= 128) and emit its explicit inverse (inv may be NULL).
//   mode 0: fail on non-positive pivot, 1: substitute sqrt(sub), 2: plain sqrt (NaN propagates; add_rows),
//   mode 3: the block already holds a factor, only the inverse is produced
int launch_potf2(fr_ctx* ctx, double* A, int64_t lda, int64_t nbk, int64_t col0, int mode, double sub,
                 double* inv, int64_t ldinv, int64_t* info, double* cest = nullptr);

// Resident panel chain: the kb x kb diagonal block at A (kb = 256 / 384 / 512) with its inverse blocks, and `rows - kb` rows below it
// solved against it, in ONE launch (potf2.hip).  Returns 1 when the shape is not taken (nothing launched: the caller keeps the chain
// of launches), FR_OK when launched.
int launch_panel_chain(fr_ctx* ctx, double* A, int64_t lda, int64_t kb, int64_t rows, int64_t col0, int mode, double sub, double* dinv,
                       int64_t* info, double* cest);

// small helpers (elementwise / reductions)
int launch_fill(fr_ctx* ctx, double* p, int64_t rows, int64_t cols, int64_t ld, double v);
int launch_blockdiag512(fr_ctx* ctx, const double* dinv128, double* w, int64_t nblocks);
int launch_copy(fr_ctx* ctx, const double* src, int64_t lds, double* dst, int64_t ldd, int64_t rows, int64_t cols);
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (process, device, kernel), under a lock: the contexts of several host
// threads (thread-ranks, a host's worker threads) share the device's function objects
int set_dyn_lds(fr_ctx* ctx, const void* fn, int bytes);
int launch_append_status(fr_ctx* ctx, const double* A, int64_t n, int64_t lda, const double* cest, int64_t nb, double* host_out);  // zero-diagonal flag + estimates -> pinned host memory
int launch_set_identity(fr_ctx* ctx, double* p, int64_t n, int64_t ld);
int launch_tri_fill(fr_ctx* ctx, double* p, int64_t n, int64_t ld, double v);            // strict upper := v
int launch_symmetrize(fr_ctx* ctx, double* p, int64_t n, int64_t ld);                    // upper := lower^T
int launch_col_norm2(fr_ctx* ctx, const double* V, int64_t n, int64_t m, int64_t ldv, double* out);
int launch_col_dot(fr_ctx* ctx, const double* U, int64_t ldu, const double* V, int64_t ldv, int64_t n, int64_t m,
                   double* out);
// out[j] = alpha * dot(V[:,j], y) + beta * out[j]
int launch_gemv_t(fr_ctx* ctx, const double* V, int64_t n, int64_t m, int64_t ldv, const double* y, double alpha,
                  double beta, double* out);
// out (cols x rows, ldo) = in (rows x cols, ldi)^T
int launch_transpose(fr_ctx* ctx, const double* in, int64_t rows, int64_t cols, int64_t ldi, double* out, int64_t ldo);
// a handful of right-hand sides (m <= ~32): A streamed once, the small operand in scalar registers
int launch_skinny_n(fr_ctx* ctx, const double* A, int64_t rows, int64_t cols, int64_t lda, const double* X, int64_t ldx,
                    int64_t m, double alpha, double beta, double* Y, int64_t ldy);
int launch_skinny_t(fr_ctx* ctx, const double* A, int64_t rows, int64_t cols, int64_t lda, const double* Y, int64_t ldy,
                    int64_t m, double alpha, double beta, double* OUT, int64_t ldo);
// y = alpha * A x + beta * y  (A rows x cols, column-major)
int launch_gemv_n(fr_ctx* ctx, const double* A, int64_t rows, int64_t cols, int64_t lda, const double* x, double alpha,
                  double beta, double* y);
int launch_axpby_vec(fr_ctx* ctx, int64_t n, double a, const double* x, double b, double* y);  // y = a*x + b*y
int launch_diag_check_zero(fr_ctx* ctx, const double* A, int64_t n, int64_t lda, int64_t* flag);
int launch_sum_log_abs(fr_ctx* ctx, const double* v, int64_t n, double* out);

// K8 (trsv.hip): b <- L^-1 b (fwd) or L^-T b with one right-hand side, one persistent launch
int launch_trsv(fr_ctx* ctx, const fr_chol* c, double* b, bool fwd, int prof_cls);
int ensure_status_word(fr_ctx* ctx);
// K9 (trsm_narrow.hip): B (n x m, 2 <= m <= 16) <- L^-1 B / L^-T B, one persistent launch on the matrix cores
int launch_trsm_narrow(fr_ctx* ctx, const fr_chol* c, double* B, int64_t m, int64_t ldb, bool fwd, int prof_cls);
// FR_HIP_ERROR if a device-side wait timed out since the last check (call after a synchronisation)
int check_status_word(fr_ctx* ctx);
// An entry point that launched a persistent solve and then left early (out of memory, bad argument) never read the status
// word: the flag -- and a possible time-out -- would be reported by the NEXT, unrelated check (a factorisation would return a
// spurious FR_HIP_ERROR).  Called at the start of solve_retry and of the factorisation entry points: waits for the stream,
// drops a stale time-out (counter "stale_status_drops").
void drain_stale_status(fr_ctx* ctx);

// Run the body of an entry point; if a persistent solve inside it gave up on a hand-off (bounded device-side wait -> status
// word -> check_status_word returns FR_HIP_ERROR and raises ctx->solve_timeout_seen), run the body ONCE more with the
// persistent kernels switched off: the same solve as a recursion of GEMM launches, which waits for nothing but stream order.
// The body must be restartable (inputs re-staged from the caller's memory, in-place device operands restored by the caller).
// collective: the body issues collectives on a sharded context -- a rank that repeated it ALONE would issue them a second time
// while its peers have moved on (mismatched collectives until the watchdog fires): no solo repeat then, the error stands.
template <class F>
int solve_retry(fr_ctx* ctx, F&& body, bool collective = false)
{
    drain_stale_status(ctx);
    ctx->solve_timeout_seen = false;
    int st = body();
    if (st == FR_HIP_ERROR && ctx->solve_timeout_seen && ctx->trsv && !collective) {
        ctx->solve_timeout_seen = false;
        (void)hipStreamSynchronize(ctx->stream);
        if (ctx->host_status) ((volatile unsigned*)ctx->host_status)[0] = 0;
        const int64_t saved = ctx->trsv;
        ctx->trsv = 0;
        ++ctx->solve_retries;
        st = body();
        ctx->trsv = saved;
    }
    return st;
}

// ---- collectives (comm.hip): RCCL over xGMI, or the in-process local transport; enqueue on ctx->ls ----------
int comm_bcast(fr_ctx* ctx, double* buf, size_t count, int root);
int comm_allgather_i64(fr_ctx* ctx, const int64_t* send, int64_t* recv, size_t count_per_rank);
// which: 0 = the first communicator, 1 = the second one (bulk stream of the chain-first schedule); see comm.hip
int comm_allgather(fr_ctx* ctx, const double* send, double* recv, size_t count_per_rank, int which = 0);
int comm_scatter(fr_ctx* ctx, double* buf, size_t count_per_rank, int root, int which = 0);  // slice r of the root's buffer -> rank r (same offset)
int comm_fanout(fr_ctx* ctx, double* buf, size_t count, int root, int which = 0);            // root's buffer -> every rank, grouped point-to-point
int comm_agree(fr_ctx* ctx, bool ok, bool* all_ok);  // collective: does EVERY rank report ok?  (synchronises)
void comm_abort(fr_ctx* ctx);                        // failing rank: tear the communicator down so that peers do not wait forever
int comm_stream_sync(fr_ctx* ctx, hipStream_t s, const char* what);  // wait for a stream that may hold collectives, with the deadline
// device -> PAGEABLE host memory on a stream that may hold collectives: such a copy blocks the host until it is done -- behind a
// collective whose peer never arrives, for ever, and in front of the bounded wait that was meant to find it (found by the mock-RCCL
// test with device-side waits, round 6: 60 s in hipMemcpyAsync, no time-out counted).  Waits for the stream WITH the deadline first.
int comm_d2h(fr_ctx* ctx, void* dst, const void* src, size_t bytes, hipStream_t s, const char* what);
void comm_drain(fr_ctx* ctx);                        // after comm_abort: give the context's three streams a bounded time to empty
void comm_destroy_internal(fr_ctx* ctx);             // fr_ctx_destroy: communicators, watchdog, events (comm.hip)
int ensure_comm2(fr_ctx* ctx);                       // collective: the second communicator exists on every rank (created on first use)

// ---- blocked algorithms (chol.hip) ------------------------------------------------------------------
// in-place Cholesky of the lower triangle of the n x n block at A (rows/cols offset col0 for bookkeeping)
int potrf_device(fr_ctx* ctx, fr_chol* c, int64_t j0, int64_t n, int mode, double sub);
// B (n x m, device) <- L^-1 B   /   B <- L^-T B
int trsm_lower_fwd(fr_ctx* ctx, const fr_chol* c, int64_t n, double* B, int64_t m, int64_t ldb, int prof_cls);
int trsm_lower_bwd(fr_ctx* ctx, const fr_chol* c, int64_t n, double* B, int64_t m, int64_t ldb, int prof_cls);
// W (n x n, leading dimension ldw) <- L^-1, strict upper triangle exactly zero; T: scratch of at least (n / 2 + 512)^2 doubles
int chol_tri_inverse(fr_ctx* ctx, const fr_chol* c, double* W, int64_t ldw, double* T, int prof_cls);
// B (k1 x m) <- L11^-T B with L11 the LEADING k1 x k1 block of the factor (k1 a multiple of 512, or the whole factor); stream-ordered paths only
int trsm_lower_bwd_leading(fr_ctx* ctx, const fr_chol* c, int64_t k1, double* B, int64_t m, int64_t ldb, int prof_cls);
int chol_alloc(fr_ctx* ctx, int64_t n, int64_t capacity, int64_t d, fr_chol** out);
int chol_fetch_info(fr_chol* c, bool with_cest = false);

}  // namespace fr
