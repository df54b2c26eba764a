// chol.hip -- blocked right-looking Cholesky with friedrich's pivot rules, GEMM-recast triangular solves,
// the bordered row-append update, and the C ABI around the device-resident factor.
//
// Reference behaviour being replaced (all single-threaded nalgebra loops there):
//   make_cholesky_cov_matrix        src/algebra/mod.rs:59-92      (Gram + noise^2 + Cholesky::new[_with_substitute])
//   add_rows_cholesky_cov_matrix    src/algebra/mod.rs:97-126     (one Cholesky::insert_column per new row)
//   Cholesky::solve_mut / solve     src/gaussian_process/mod.rs:235, 298, 379
//   Cholesky::l().solve_lower_triangular                          mod.rs:203, 260-263, 342-345
//   Cholesky::inverse               src/gaussian_process/optimizer.rs:32, 169
//   DMatrix::cholesky().unpack()    src/gaussian_process/multivariate_normal.rs:57
//
// Structure.  Block sizes:
//   128  K4 potf2 + "inverse block": one workgroup factors a 128 x 128 diagonal block and emits its explicit inverse
//        (potf2.hip); the inverses are kept next to the factor (fr_chol::dinv).  Every triangular solve inside the
//        factorisation is a GEMM against these inverses; with N (or M) <= 128 the GEMM has a single tile column (row), so
//        it can safely run in place.
//   512  leaves of the solves of the predict family: explicit inverses of the 512 x 512 diagonal blocks, assembled on
//        demand from the 128-block inverses by batched GEMMs (fr_chol::inv512).
//   nb   outer block (pick_nb: 1024 at N >= 18432 on one GPU -- 2048 while more than 22528 rows remain -- else 512): the trailing update A22 -= P P^T is one
//        lower-triangular SYRK launch with K = nb, the dominant FP64-MFMA kernel (n^3/3 of the flops).
// Solves pick their kernels by the number of right-hand sides: <= 16 memory-bound kernels (L streamed once), otherwise
// the recursive GEMM formulation.
#include "fr_internal.hpp"
#include <algorithm>
#include <chrono>

#include "kprog_device.hpp"

namespace fr {

constexpr int64_t IB = 128;            // inverse block
constexpr int64_t INV_ELEMS = IB * IB;

static inline int64_t imin(int64_t a, int64_t b) { return a < b ? a : b; }
static inline int64_t imax(int64_t a, int64_t b) { return a > b ? a : b; }

static int gemm(fr_ctx* ctx, int cls, int64_t M, int64_t N, int64_t K, const double* A, int64_t lda, bool a_kmajor,
                const double* B, int64_t ldb, bool b_kmajor, double alpha, double beta, double* D, int64_t ldd,
                bool lower = false)
{
    GemmDesc g;
    g.M = M;
    g.N = N;
    g.K = K;
    g.A = A;
    g.lda = lda;
    g.a_kmajor = a_kmajor;
    g.B = B;
    g.ldb = ldb;
    g.b_kmajor = b_kmajor;
    g.Cin = D;
    g.ldcin = ldd;
    g.D = D;
    g.ldd = ldd;
    g.alpha = alpha;
    g.beta = beta;
    g.lower = lower;
    g.prof_cls = cls;
    return launch_gemm(ctx, g);
}

static inline int64_t ctx_cest_col0(const fr_ctx*) { return 0; }  // estimates are only kept for whole-matrix factorisations (col0 == 0)

// Factor the sb x sb (sb <= 128) diagonal block at A and produce its inverse in `inv` (ld 128): one K4 launch.
static int factor_block128(fr_ctx* ctx, double* A, int64_t ld, int64_t sb, int64_t col, int mode, double sub, double* inv,
                           int64_t* info, double* /*T*/)
{
    return launch_potf2(ctx, A, ld, sb, col, mode, sub, inv, IB, info, ctx->cur_cest ? ctx->cur_cest + (col - ctx_cest_col0(ctx)) / IB : nullptr);
}

// Rebuild the inverse of an already factored 128-block (serde upload, add_rows re-alignment).
static int invert_block128(fr_ctx* ctx, double* A, int64_t ld, int64_t sb, double* inv, double* cest)
{
    return launch_potf2(ctx, A, ld, sb, 0, 3, 0.0, inv, IB, nullptr, cest);
}

// Factor the kb-wide column block starting at column k (rows k..n), which must already carry every update of the
// earlier panels.  Recursive halving down to the 128-wide inverse blocks: the second half of the block is updated by
// the first with ONE GEMM of depth kb/2 (instead of a depth-128 GEMM per 128 columns) -- the read-modify-write of the
// result tile is a fixed cost per tile, so the deeper the contraction the closer the GEMM runs to the MFMA rate.
// (the launcher's own shape test, potf2.hip: launch_panel_chain)
static inline bool panel_chain_fits(const fr_ctx* ctx, int64_t kb, int64_t rows, int mode)
{
    return ctx->panel_chain && ctx->trsv && !ctx->refine_now && mode != 3 && kb >= 2 * IB && kb <= 4 * IB && kb % IB == 0 && rows >= kb &&
           (rows - kb > 512 || 1 + (rows - IB + 15) / 16 <= ctx->num_cus - 8);
}

static int factor_panel(fr_ctx* ctx, double* A, int64_t ld, int64_t n, int64_t k, int64_t kb, int64_t col0, int mode,
                        double sub, double* dinv, int64_t* info, double* T)
{
    // The resident panel chain (potf2.hip: panel_chain_kernel): the kb x kb diagonal block -- and the rows below it, as many as stay
    // resident -- in ONE launch instead of kb / 128 diagonal-block kernels and 2 kb / 128 - 1 launch-bound products.  Taken where the
    // launch has CUs of its own: no second stream (small matrices, the Schur block of an append), the sharded chain (diagonal block
    // only), and -- option panel_chain = 2 -- wherever the shape fits.
    if (panel_chain_fits(ctx, kb, n - k, mode) && ctx->cols_final_at < 0 && (ctx->panel_chain == 2 || ctx->k4_alone || n == k + kb)) {
        const int64_t col = col0 + k;
        const int rc = launch_panel_chain(ctx, A + k + k * ld, ld, kb, n - k, col, mode, sub, dinv + (k / IB) * INV_ELEMS, info,
                                          ctx->cur_cest ? ctx->cur_cest + (col - ctx_cest_col0(ctx)) / IB : nullptr);
        if (rc == FR_OK) {
            ++ctx->panel_chain_launches;
            return FR_OK;
        }
        if (rc != 1) return rc;
    }
    if (kb <= IB) {
        double* inv = dinv + (k / IB) * INV_ELEMS;
        FR_TRY(factor_block128(ctx, A + k + k * ld, ld, kb, col0 + k, mode, sub, inv, info, T));
        const int64_t below = n - (k + kb);
        if (below > 0) {
            // K5: panel TRSM  B <- B * L_kk^-T  as a GEMM against the explicit inverse (in place: one tile column)
            double* B = A + (k + kb) + k * ld;
            if (ctx->refine_now && T) {
                // one step of iterative refinement against the triangular factor itself (its strict upper triangle was
                // zeroed by the diagonal-block kernel):  X0 = B W^T;  R = B - X0 L^T;  X = X0 + R W^T.  The explicit inverse
                // alone leaves a residual of cond(L_kk) u; after the step it is what substitution would leave.
                const double* Lkk = A + k + k * ld;
                GemmDesc g;
                g.M = below; g.N = kb; g.K = kb;
                g.A = B; g.lda = ld; g.a_kmajor = false;
                g.B = inv; g.ldb = IB; g.b_kmajor = false;
                g.Cin = T; g.ldcin = below; g.D = T; g.ldd = below;
                g.alpha = 1.0; g.beta = 0.0; g.lower = false; g.prof_cls = FR_PROF_GEMM_PANEL;
                FR_TRY(launch_gemm(ctx, g));                                      // X0 -> T
                FR_TRY(gemm(ctx, FR_PROF_GEMM_PANEL, below, kb, kb, T, below, false, Lkk, ld, false, -1.0, 1.0, B, ld));  // R (in B)
                g.A = B; g.lda = ld;
                g.Cin = T; g.ldcin = below; g.D = B; g.ldd = ld;
                g.alpha = 1.0; g.beta = 1.0;
                FR_TRY(launch_gemm(ctx, g));                                      // B <- X0 + R W^T  (in place: one tile column)
            } else {
                FR_TRY(gemm(ctx, FR_PROF_GEMM_PANEL, below, kb, kb, B, ld, false, inv, IB, false, 1.0, 0.0, B, ld));
            }
        }
        // the look-ahead pipeline asked to be told when the panel's columns up to here are final (see potrf_blocked)
        if (ctx->cols_final_at == k + kb && ctx->ev_cols) FR_HIP(ctx, hipEventRecord(ctx->ev_cols, ctx->ls));
        return FR_OK;
    }
    const int64_t kb1 = ((kb / IB + 1) / 2) * IB;  // first half, a multiple of 128
    FR_TRY(factor_panel(ctx, A, ld, n, k, kb1, col0, mode, sub, dinv, info, T));
    const int64_t below = n - (k + kb1);
    if (below > 0) {
        const double* P1 = A + (k + kb1) + k * ld;  // rows below the first half, its kb1 columns
        FR_TRY(gemm(ctx, FR_PROF_GEMM_PANEL, below, kb - kb1, kb1, P1, ld, false, P1, ld, false, -1.0, 1.0,
                    A + (k + kb1) + (k + kb1) * ld, ld));
    }
    return factor_panel(ctx, A, ld, n, k + kb1, kb - kb1, col0, mode, sub, dinv, info, T);
}

// Multi-GPU: block column b (width nb) of the matrix being factored is owned by rank b % world.
static inline int owner_of(int64_t col, int64_t nb, int world) { return (int)((col / nb) % world); }

// Replicate the factored panel (rows k..n of block column k, plus its inverse blocks) from its owner to every rank:
// pack -> one broadcast over xGMI -> unpack.  Enqueued on ctx->ls by every rank in the same order.
static int exchange_panel(fr_ctx* ctx, double* A, int64_t ld, int64_t n, int64_t k, int64_t kb, double* dinv, double* buf,
                          int owner)
{
    const int64_t rows = n - k;
    const int64_t nblk = (kb + IB - 1) / IB;
    double* tail = buf + rows * kb;
    double* dblk = dinv + (k / IB) * INV_ELEMS;
    if (ctx->rank == owner) {
        FR_TRY(launch_copy(ctx, A + k + k * ld, ld, buf, rows, rows, kb));
        FR_HIP(ctx, hipMemcpyAsync(tail, dblk, sizeof(double) * (size_t)(nblk * INV_ELEMS), hipMemcpyDeviceToDevice, ctx->ls));
    }
    FR_TRY(comm_bcast(ctx, buf, (size_t)(rows * kb + nblk * INV_ELEMS), owner));
    if (ctx->rank != owner) {
        FR_TRY(launch_copy(ctx, buf, rows, A + k + k * ld, ld, rows, kb));
        FR_HIP(ctx, hipMemcpyAsync(dblk, tail, sizeof(double) * (size_t)(nblk * INV_ELEMS), hipMemcpyDeviceToDevice, ctx->ls));
    }
    return FR_OK;
}

// rows x kb block S (leading dimension lds) of still unsolved rows of the panel at column k  ->  S L_kk^-T, against the factored
// kb x kb diagonal block in A and its 128-block inverses: ONE launch, a workgroup takes 32 rows through the left-looking
// sweep over the 128-column sub-panels (gemm_f64.hip: rows_solve_kernel).
static int solve_rows(fr_ctx* ctx, double* S, int64_t lds, int64_t rows, const double* A, int64_t ld, int64_t k, int64_t kb,
                      const double* dblk)
{
    return launch_rows_solve(ctx, S, lds, rows, A + k + k * ld, ld, kb, dblk);
}

// The split variant of a panel step (option dist_schedule = 1): the owner factors the kb x kb DIAGONAL block only and
// broadcasts it with its inverse blocks (2.5 MB at kb = 512); the still unsolved rows below it are scattered in W equal
// slices (each over its own xGMI link), every rank solves its slice against the diagonal block -- the serial part of the
// step shrinks from "the whole panel on one GPU" to "the diagonal block on one GPU" -- and one all-gather, which moves
// 1 / W of the panel over every link of every GPU, returns the slices to everybody.  buf: head (kb * kb + inverses) followed by
// W slices of slice_rows x kb (leading dimension slice_rows).
static int split_panel(fr_ctx* ctx, double* A, int64_t ld, int64_t n, int64_t k, int64_t kb, int64_t col0, int mode, double sub,
                       double* dinv, int64_t* info, double* T, double* buf, int64_t slice_rows, int owner)
{
    const int W = ctx->world, me = ctx->rank;
    const int64_t nblk = (kb + IB - 1) / IB;
    const int64_t head_count = round_up(kb * kb + nblk * INV_ELEMS, kAlign);
    const int64_t row0 = k + kb, below = n - row0;
    double* head = buf;
    double* slices = buf + head_count;
    double* dblk = dinv + (k / IB) * INV_ELEMS;
    auto rows_of = [&](int q) { return imax(0, imin(slice_rows, below - (int64_t)q * slice_rows)); };
    if (me == owner) {
        FR_TRY(factor_panel(ctx, A, ld, k + kb, k, kb, col0, mode, sub, dinv, info, T));  // rows k .. k + kb only
        FR_TRY(launch_copy(ctx, A + k + k * ld, ld, head, kb, kb, kb));
        FR_HIP(ctx, hipMemcpyAsync(head + kb * kb, dblk, sizeof(double) * (size_t)(nblk * INV_ELEMS), hipMemcpyDeviceToDevice, ctx->ls));
        for (int q = 0; q < W; ++q)
            if (rows_of(q) > 0)
                FR_TRY(launch_copy(ctx, A + row0 + (int64_t)q * slice_rows + k * ld, ld, slices + (int64_t)q * slice_rows * kb,
                                   slice_rows, rows_of(q), kb));
    }
    FR_TRY(comm_bcast(ctx, head, (size_t)head_count, owner));
    if (me != owner) {
        FR_TRY(launch_copy(ctx, head, kb, A + k + k * ld, ld, kb, kb));
        FR_HIP(ctx, hipMemcpyAsync(dblk, head + kb * kb, sizeof(double) * (size_t)(nblk * INV_ELEMS), hipMemcpyDeviceToDevice, ctx->ls));
    }
    if (below <= 0) return FR_OK;
    FR_TRY(comm_scatter(ctx, slices, (size_t)(slice_rows * kb), owner));
    if (rows_of(me) > 0) FR_TRY(solve_rows(ctx, slices + (int64_t)me * slice_rows * kb, slice_rows, rows_of(me), A, ld, k, kb, dblk));
    FR_TRY(comm_allgather(ctx, slices + (int64_t)me * slice_rows * kb, slices, (size_t)(slice_rows * kb)));
    for (int q = 0; q < W; ++q)
        if (rows_of(q) > 0)
            FR_TRY(launch_copy(ctx, slices + (int64_t)q * slice_rows * kb, slice_rows, A + row0 + (int64_t)q * slice_rows + k * ld, ld,
                               rows_of(q), kb));
    return FR_OK;
}

// ---- sharded factorisation, schedule 2: the chain of diagonal blocks first ------------------------------------------------
// What is serial in a right-looking Cholesky over block columns dealt round-robin to W ranks is the chain of diagonal
// blocks: D_p needs the row tile p of every earlier panel.  Schedules 0 / 1 put the whole panel step on that chain
// (factor all rows -> broadcast, or diagonal block -> scatter -> solves -> all-gather, one after the other on one stream and
// one communicator).  Here the chain carries only what the NEXT diagonal block waits for:
//   M_p = R1_p = L[panel p + 1's rows, p]   solved by the owner itself (one launch), fanned out to every rank   (~2 MB at nb = 512)
// on the panel stream / first communicator; the next owner applies R1_p to its diagonal block at once (u1) and factors D_p+1 --
// it needs nothing else of panel p.  Everything else follows on the bulk stream and the second communicator:
//   H_p = [D_p, its 128-block inverses]     fan-out from the owner to every rank        (~2.5 MB)
//   the rows below R1_p: scatter of the unsolved rows in W slices (one xGMI link each), every rank solves its slice against
//   D_p (one launch), one all-gather -- 1 / W of the panel over every link of every GPU.
// The trailing updates (main stream) take a panel once its bulk has arrived; each rank updates its NEAREST owned block
// column first (for the next owner: the rows of R1_p+1 before the others), because that is the column the chain waits for.
// Dependencies between the three streams are HIP events; the critical path per panel is
//   max( D factor + R1 solve + one fan-out + u1,  (bulk latency + first look-ahead tile + R1 solve + fan-out + u1) / 2 )
// instead of their sum (D_p+1 needs the bulk of panel p - 1, not of panel p).  DESIGN.md section 6 has the model.
// Every rank issues the same sequence of communication calls in the same host order, and that order is a topological order of
// the dependency graph -- which is what keeps two communicators used side by side free of deadlock (comm.hip).
enum { EV_HEAD = 0, EV_MSG = 1, EV_BULK = 2, EV_LA1 = 3, EV_NEAR = 4 };

static int potrf_dist_chain(fr_ctx* ctx, double* A, int64_t ld, int64_t n, int64_t col0, int mode, double sub, double* dinv,
                            int64_t* info, int64_t nb)
{
    const int W = ctx->world, me = ctx->rank;
    if (nb % IB != 0) return set_err(ctx, FR_INVALID_ARGUMENT, "multi-GPU factorisation needs nb %% 128 == 0");
    hipStream_t S0 = ctx->stream, S1 = ctx->stream2, S3 = ctx->stream3;
    FR_TRY(ensure_comm2(ctx));  // (collective; created on the first factorisation that takes this schedule)
    for (int kind = 0; kind < 5; ++kind)
        for (int i = 0; i < 4; ++i)
            if (!ctx->ev_ring[kind][i]) FR_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_ring[kind][i], hipEventDisableTiming));
    const int64_t P = (n + nb - 1) / nb;
    auto kof = [&](int64_t p) { return imin(p * nb, n); };
    auto owner = [&](int64_t p) { return (int)(p % W); };
    auto ev = [&](int kind, int64_t p) { return ctx->ev_ring[kind][p & 3]; };
    const int64_t kbm = imin(nb, n);
    const int64_t head_cap = round_up(kbm * kbm + ((kbm + IB - 1) / IB) * INV_ELEMS, kAlign);
    const int64_t sr_max = round_up((n + W - 1) / W, IB);
    WsGuard hg(ctx), mg(ctx), sg(ctx);
    double* hb = hg.get(sizeof(double) * (size_t)head_cap);
    double* mb = mg.get(sizeof(double) * (size_t)(kbm * kbm));
    double* sb = sg.get(sizeof(double) * (size_t)W * (size_t)sr_max * (size_t)kbm);
    // every allocation of this rank is behind it: agree with the peers BEFORE the first exchange, so that a rank that ran
    // out of memory does not leave the others inside a transfer that never completes
    bool all_ok = true;
    FR_TRY(comm_agree(ctx, hb && mb && sb, &all_ok));
    if (!hb || !mb || !sb) return FR_OUT_OF_MEMORY;
    if (!all_ok) return set_err(ctx, FR_OUT_OF_MEMORY, "a peer rank could not allocate its panel buffers: sharded factorisation abandoned on every rank");
    if (ctx->xcd_reserve != 0 && ctx->claim_ring) {
        FR_HIP(ctx, hipMemsetAsync(ctx->claim_ring, 0, sizeof(unsigned) * 2 * kClaimSlots, S0));
        ctx->claim_next = 0;
    }
    int st = FR_OK;
    auto fail = [&](int code) {
        comm_abort(ctx);  // a host-side failure past this point: the peers must not wait for this rank
        ctx->ls = S0;
        ctx->reserve_now = 0;
        comm_drain(ctx);
        return code;
    };
#define CH_HIP(call)                              \
    do {                                          \
        if ((call) != hipSuccess) {               \
            (void)hipGetLastError();              \
            return fail(set_err(ctx, FR_HIP_ERROR, "%s failed (%s:%d)", #call, __FILE__, __LINE__)); \
        }                                         \
    } while (0)
#define CH_TRY(call)                              \
    do {                                          \
        st = (call);                              \
        if (st != FR_OK) return fail(st);         \
    } while (0)
    // both side streams start after everything already queued on the main stream (Gram assembly)
    CH_HIP(hipEventRecord(ctx->ev_la, S0));
    CH_HIP(hipStreamWaitEvent(S1, ctx->ev_la, 0));
    CH_HIP(hipStreamWaitEvent(S3, ctx->ev_la, 0));
    ++ctx->panel_epoch;
    for (int64_t p = 0; p < P; ++p) {
        const int64_t k = kof(p), k1 = kof(p + 1), k2 = kof(p + 2), k3 = kof(p + 3);
        const int64_t kb = k1 - k, kb1 = k2 - k1, kb2 = k3 - k2, below = n - k2;
        const int own = owner(p);
        const int64_t nblk = (kb + IB - 1) / IB;
        const int64_t head_count = round_up(kb * kb + nblk * INV_ELEMS, kAlign);
        double* dblk = dinv + (k / IB) * INV_ELEMS;
        // The chain bounds a sharded factorisation from the first panel on (a rank's share of a trailing update is 1 / W of it):
        // the main and bulk streams' products keep off the XCDs of the diagonal-block kernels while a rank's share of the
        // trailing update is shorter than a chain step (~0.6 ms: (n - k)^2 nb / W flop at 60 TF/s)
        ctx->reserve_now = 0;
        if (ctx->claim_ring && kb <= 512) {
            if (ctx->xcd_reserve < 0) {
                const double share_ms = (double)(n - k1) * (double)(n - k1) * (double)kb / (double)W / 6.0e10;
                ctx->reserve_now = share_ms < 0.6 ? 2 : (share_ms < 1.2 ? 1 : 0);
            } else {
                ctx->reserve_now = (int)ctx->xcd_reserve;
            }
        }
        // ---- 1. chain: D_p on its owner (its block carries every update: u1 of round p - 1 is ahead on this stream)
        ctx->ls = S1;
        if (me == own) {
            CH_TRY(factor_panel(ctx, A, ld, k1, k, kb, col0, mode, sub, dinv, info, nullptr));  // rows k .. k1 only
            CH_HIP(hipEventRecord(ev(EV_HEAD, p), S1));
        }
        // ---- 2. M_p = R1_p: the rows of the next panel's diagonal block, solved by the owner in place and fanned out -- the
        //      ONE message of the chain: the next diagonal block needs nothing else of panel p
        if (kb1 > 0) {
            double* R1 = A + k1 + k * ld;
            if (me == own) {
                if (p > 0) CH_HIP(hipStreamWaitEvent(S1, ev(EV_LA1, p), 0));  // these rows carry panel p - 1's update (main stream)
                CH_TRY(solve_rows(ctx, R1, ld, kb1, A, ld, k, kb, dblk));
                CH_TRY(launch_copy(ctx, R1, ld, mb, kb1, kb1, kb));
            }
            CH_TRY(comm_fanout(ctx, mb, (size_t)(kb1 * kb), own, 0));
            if (me != own) CH_TRY(launch_copy(ctx, mb, kb1, R1, ld, kb1, kb));
            CH_HIP(hipEventRecord(ev(EV_MSG, p), S1));
            if (me == owner(p + 1)) {
                // u1: the next diagonal block takes panel p's update here, on the chain; the updates of the panels before
                // (this rank's main stream, nearest column first) are ordered in front of it
                if (p > 0) CH_HIP(hipStreamWaitEvent(S1, ev(EV_NEAR, p - 1), 0));
                CH_TRY(gemm(ctx, FR_PROF_GEMM_PANEL, kb1, kb1, kb, R1, ld, false, R1, ld, false, -1.0, 1.0, A + k1 + k1 * ld, ld, true));
            }
        }
        // ---- 3. bulk stream / second communicator: H_p = [D_p, inverses] to every rank (the slices are solved against it; the
        //      next owner does NOT wait for it), then the rows from k2 on in W slices of whole 128-row blocks
        ctx->ls = S3;
        if (me == own) {
            CH_HIP(hipStreamWaitEvent(S3, ev(EV_HEAD, p), 0));
            CH_TRY(launch_copy(ctx, A + k + k * ld, ld, hb, kb, kb, kb));
            CH_HIP(hipMemcpyAsync(hb + kb * kb, dblk, sizeof(double) * (size_t)(nblk * INV_ELEMS), hipMemcpyDeviceToDevice, S3));
        }
        CH_TRY(comm_fanout(ctx, hb, (size_t)head_count, own, 1));
        if (me != own) {
            CH_TRY(launch_copy(ctx, hb, kb, A + k + k * ld, ld, kb, kb));
            CH_HIP(hipMemcpyAsync(dblk, hb + kb * kb, sizeof(double) * (size_t)(nblk * INV_ELEMS), hipMemcpyDeviceToDevice, S3));
        }
        if (below > 0) {
            const int64_t sr = round_up((below + W - 1) / W, IB);
            auto rows_of = [&](int q) { return imax(0, imin(sr, below - (int64_t)q * sr)); };
            if (me == own) {
                if (p > 0) CH_HIP(hipStreamWaitEvent(S3, ev(EV_NEAR, p - 1), 0));  // column p carries every update of the panels before
                for (int q = 0; q < W; ++q)
                    if (rows_of(q) > 0)
                        CH_TRY(launch_copy(ctx, A + k2 + (int64_t)q * sr + k * ld, ld, sb + (int64_t)q * sr * kb, sr, rows_of(q), kb));
            }
            CH_TRY(comm_scatter(ctx, sb, (size_t)(sr * kb), own, 1));
            if (rows_of(me) > 0) CH_TRY(solve_rows(ctx, sb + (int64_t)me * sr * kb, sr, rows_of(me), A, ld, k, kb, dblk));
            CH_TRY(comm_allgather(ctx, sb + (int64_t)me * sr * kb, sb, (size_t)(sr * kb), 1));
            for (int q = 0; q < W; ++q)
                if (rows_of(q) > 0)
                    CH_TRY(launch_copy(ctx, sb + (int64_t)q * sr * kb, sr, A + k2 + (int64_t)q * sr + k * ld, ld, rows_of(q), kb));
        }
        CH_HIP(hipEventRecord(ev(EV_BULK, p), S3));
        // ---- 5. trailing updates with panel p (main stream): the nearest owned block column first
        if (kb1 > 0) {
            ctx->ls = S0;
            CH_HIP(hipStreamWaitEvent(S0, ev(EV_MSG, p), 0));
            if (below > 0) CH_HIP(hipStreamWaitEvent(S0, ev(EV_BULK, p), 0));  // (H_p alone is of no interest to the updates)
            const int64_t q = p + 1 + (((int64_t)me - (p + 1)) % W + W) % W;  // nearest panel after p that this rank owns
            const double* Lp = A + k * ld;  // block column p of the factor: row r at Lp + r
            if (q == p + 1) {
                // the next panel is mine: its diagonal block was updated on the chain (u1); the rows of R1_p+1 go first
                if (kb2 > 0)
                    CH_TRY(gemm(ctx, FR_PROF_GEMM_PANEL, kb2, kb1, kb, Lp + k2, ld, false, Lp + k1, ld, false, -1.0, 1.0, A + k2 + k1 * ld, ld));
                CH_HIP(hipEventRecord(ev(EV_LA1, p + 1), S0));
                if (n > k3)
                    CH_TRY(gemm(ctx, FR_PROF_GEMM_PANEL, n - k3, kb1, kb, Lp + k3, ld, false, Lp + k1, ld, false, -1.0, 1.0, A + k3 + k1 * ld, ld));
            } else if (q < P) {
                const int64_t kq = kof(q), wq = kof(q + 1) - kq;
                CH_TRY(gemm(ctx, FR_PROF_GEMM_PANEL, n - kq, wq, kb, Lp + kq, ld, false, Lp + kq, ld, false, -1.0, 1.0, A + kq + kq * ld, ld));
            }
            CH_HIP(hipEventRecord(ev(EV_NEAR, p), S0));
            const int64_t ks = kof(q + 1);  // the owned block columns behind the nearest one
            if (q < P && n > ks) {
                GemmDesc g;
                g.M = n - ks; g.N = n - ks; g.K = kb;
                g.A = Lp + ks; g.lda = ld; g.a_kmajor = false;
                g.B = Lp + ks; g.ldb = ld; g.b_kmajor = false;
                g.D = A + ks + ks * ld; g.ldd = ld;
                g.Cin = g.D; g.ldcin = ld;
                g.alpha = -1.0; g.beta = 1.0; g.lower = true; g.prof_cls = FR_PROF_SYRK;
                g.own_world = W; g.own_rank = me; g.own_nb = nb; g.own_col0 = ks;
                CH_TRY(launch_gemm(ctx, g));
            }
        }
    }
    // the factorisation is complete when all three streams are
    ctx->ls = S1;
    if (ctx->xcd_reserve != 0) CH_TRY(launch_release_xcds(ctx, ctx->panel_epoch));
    CH_HIP(hipEventRecord(ctx->ev_panel, S1));
    CH_HIP(hipEventRecord(ctx->ev_bulk, S3));
    CH_HIP(hipStreamWaitEvent(S0, ctx->ev_panel, 0));
    CH_HIP(hipStreamWaitEvent(S0, ctx->ev_bulk, 0));
#undef CH_HIP
#undef CH_TRY
    ctx->ls = S0;
    ctx->reserve_now = 0;
    return FR_OK;
}

// In-place blocked Cholesky of the n x n lower triangle at A.  dinv receives the inverses of the diagonal
// 128-blocks (block i of this sub-matrix at dinv + i*INV_ELEMS).
//
// Look-ahead: the panel path (K4 + K5: ~10 small, latency-bound launches per 128 columns) would otherwise
// serialise with the big trailing SYRK.  Step k's trailing update is therefore split: the next panel's nb columns
// are updated first, then that panel is factored on a second, high-priority HIP stream while the main stream
// updates the rest of the trailing matrix (K6).  The two streams touch disjoint columns; events order them.
//
// dist (and ctx->world > 1): the block columns are dealt round-robin to the ranks.  Only the owner factors a panel;
// the panel then travels to every rank with one broadcast on the panel stream (so it overlaps the trailing updates
// still running on the main stream), and every rank updates only the trailing block columns it owns.  Because every
// panel is broadcast, each rank ends up holding the complete factor -- no final all-gather is needed.
static int potrf_blocked(fr_ctx* ctx, double* A, int64_t ld, int64_t n, int64_t col0, int mode, double sub, double* dinv,
                         int64_t* info, int64_t nb, bool dist = false)
{
    if (n <= 0) return FR_OK;
    WsGuard tg(ctx), pg(ctx);
    double* T = nullptr;  // scratch of the refined panel solves (rows below a block x 128)
    if (ctx->refine_now) {
        T = tg.get(sizeof(double) * (size_t)n * (size_t)IB);
        if (!T) return FR_OUT_OF_MEMORY;
    }
    const int world = dist ? ctx->world : 1;
    const int rank = ctx->rank;
    if (world > 1 && ctx->dist_schedule == 2 && ctx->stream3 && !ctx->refine_now && mode != 3)
        return potrf_dist_chain(ctx, A, ld, n, col0, mode, sub, dinv, info, nb);
    if (world > 1 && nb % IB != 0) return set_err(ctx, FR_INVALID_ARGUMENT, "multi-GPU factorisation needs nb %% 128 == 0");
    const bool la = (world > 1) || (ctx->lookahead && ctx->stream2 && n > 2 * nb && ctx->ls == ctx->stream);
    if (!la) {
        // (no second stream: the diagonal-block kernels have their CU to themselves -- the uncapped symbol, potf2.hip)
        struct Alone {
            fr_ctx* ctx;
            bool saved;
            ~Alone() { ctx->k4_alone = saved; }
        } alone{ctx, ctx->k4_alone};
        ctx->k4_alone = true;
        for (int64_t k = 0; k < n; k += nb) {
            const int64_t kb = imin(nb, n - k);
            FR_TRY(factor_panel(ctx, A, ld, n, k, kb, col0, mode, sub, dinv, info, T));
            const int64_t rest = n - (k + kb);
            if (rest > 0) {
                // K6: trailing update, lower triangle only
                const double* P = A + (k + kb) + k * ld;
                FR_TRY(gemm(ctx, FR_PROF_SYRK, rest, rest, kb, P, ld, false, P, ld, false, -1.0, 1.0,
                            A + (k + kb) + (k + kb) * ld, ld, true));
            }
        }
        return FR_OK;
    }
    double* pbuf = nullptr;
    int64_t split_rows = 0;
    // (a refined factorisation -- rare: ill-conditioned diagonal blocks -- takes the whole-panel schedule whatever the option
    // says: its owner solves the panel's rows with the refinement step behind every inverse product, factor_panel)
    const bool split = world > 1 && ctx->dist_schedule != 0 && !ctx->refine_now;
    if (world > 1) {
        // (split variant: head + W slices of whole 128-row blocks)
        split_rows = round_up((n + world - 1) / world, IB);
        const int64_t kbm = imin(nb, n);
        const int64_t split_elems = round_up(kbm * kbm + ((kbm + IB - 1) / IB) * INV_ELEMS, kAlign) + (int64_t)world * split_rows * kbm;
        const int64_t plain_elems = n * kbm + ((nb + IB - 1) / IB) * INV_ELEMS;
        pbuf = pg.get(sizeof(double) * (size_t)imax(split_elems, plain_elems));
        // every allocation of this rank is behind it: agree with the peers BEFORE the first panel exchange, so that a rank
        // that ran out of memory does not leave the others inside a broadcast that never completes
        bool all_ok = true;
        FR_TRY(comm_agree(ctx, pbuf != nullptr, &all_ok));
        if (!pbuf) return FR_OUT_OF_MEMORY;
        if (!all_ok) return set_err(ctx, FR_OUT_OF_MEMORY, "a peer rank could not allocate its panel buffers: sharded factorisation abandoned on every rank");
    }
    hipStream_t S0 = ctx->stream, S1 = ctx->stream2;
    int st = FR_OK;
    const bool cu_ok = ctx->cu_reserve != 0 && ctx->xcd_reserve != 0 && cu_table_ready(ctx);  // (builds the CU rank table on first use: one small launch + a synchronisation, here rather than between two panels)
    if (world == 1 && ctx->xcd_reserve != 0 && ctx->claim_ring) {
        // claim counters of the launches that keep off the panel stream's XCD (gemm_f64.hip): one pair per launch
        FR_HIP(ctx, hipMemsetAsync(ctx->claim_ring, 0, sizeof(unsigned) * 2 * kClaimSlots, S0));
        ctx->claim_next = 0;
    }
    auto fail = [&](int code) {
        ctx->ls = S0;
        ctx->reserve_now = 0;
        ctx->reserve_by_cu_now = false;
        ctx->cols_final_at = -1;
        if (world > 1) {
            comm_abort(ctx);  // a host-side failure past this point: the peers must not wait for this rank
            comm_drain(ctx);
        } else {
            (void)hipStreamSynchronize(S1);
        }
        return code;
    };
    // the panel stream starts after everything already queued on the main stream (Gram assembly)
    FR_HIP(ctx, hipEventRecord(ctx->ev_la, S0));
    FR_HIP(ctx, hipStreamWaitEvent(S1, ctx->ev_la, 0));
    ctx->ls = S1;
    // Panel width per step.  Fixed (nb) except on a single GPU with nb > 512 (the automatic choice at large n): nb columns while the
    // trailing update dominates (each pass over the trailing matrix costs a read and a write of it whatever its depth), 512
    // once the panel chain does -- from there on the factorisation is the one of an n = nb_switch_rows matrix, XCD
    // reservation (nb <= 512) included.  With the automatic nb = 1024 the first panels are 2048 wide while more than nb_big_rows
    // rows remain: there one pass over the trailing matrix is > 4 GB of traffic and half as many of them is worth more than
    // the longer panels cost (in-process A/B, N = 30720 / 32768: 157.0 -> 155.4 / 188.4 -> 185.4 ms; neutral at 24576 / 26624).
    auto width = [&](int64_t remaining) -> int64_t {
        if (world == 1 && nb > 512 && ctx->nb_switch_rows > 0 && remaining <= ctx->nb_switch_rows) return imin(512, remaining);
        if (world == 1 && ctx->nb == 0 && nb == 1024 && ctx->nb_big_rows > 0 && remaining > ctx->nb_big_rows) return imin(2048, remaining);
        return imin(nb, remaining);
    };
    const int64_t kb0 = width(n);
    // The look-ahead update of the NEXT panel's columns by the panel being factored sits between two panels on the critical
    // path (measured at N = 8192: 110 us of the ~600 us per 512 columns).  Most of it does not have to: once all but the last 128
    // columns of a panel are final (the leaf before the last says so through ev_cols), the main stream applies those columns'
    // part of the update (K = kb - 128) while the panel stream still factors the last block; what remains between the panels
    // is the K = 128 part.  Single GPU only (the sharded schedules have their own look-ahead).
    bool la_split = false;  // the first part of the update by the panel just finished has been issued
    bool la_on_panel = false;  // ... and its remainder too, on the panel stream: the update is complete in that stream's order
    auto la_hook = [&](int64_t kk, int64_t kbb) {
        ctx->cols_final_at = -1;
        if (ctx->panel_chain == 2 && panel_chain_fits(ctx, kbb, n - kk, mode)) return;  // (one resident launch for the whole panel: nothing becomes final early)
        if (world == 1 && ctx->ev_cols && kbb > IB && kbb % IB == 0 && n - (kk + kbb) > 0) ctx->cols_final_at = kk + kbb - IB;
    };
    auto la_first_part = [&](int64_t kk, int64_t kbb) -> int {  // on S0, behind whatever trailing update is queued there
        la_split = false;
        const int64_t after = n - (kk + kbb);
        if (ctx->panel_chain == 2 && panel_chain_fits(ctx, kbb, n - kk, mode)) return FR_OK;
        if (!(world == 1 && ctx->ev_cols && kbb > IB && kbb % IB == 0 && after > 0)) return FR_OK;
        FR_HIP(ctx, hipStreamWaitEvent(S0, ctx->ev_cols, 0));
        const double* Pn = A + (kk + kbb) + kk * ld;
        FR_TRY(gemm(ctx, FR_PROF_GEMM_PANEL, after, width(after), kbb - IB, Pn, ld, false, Pn, ld, false, -1.0, 1.0,
                    A + (kk + kbb) + (kk + kbb) * ld, ld));
        la_split = true;
        return FR_OK;
    };
    // once the panel chain is longer than the trailing update, the update's launches leave XCDs / CUs to the panel stream
    // (gemm_f64.hip / gemm_tile.hpp: claim_item, gemm_f64_persist_body); `remaining` = rows from the panel about to be factored on
    auto set_reservation = [&](int64_t remaining, int64_t kbw) {
        ctx->reserve_now = 0;
        ctx->reserve_by_cu_now = false;
        ++ctx->panel_epoch;
        if (world == 1 && ctx->claim_ring) {
            if (ctx->xcd_reserve < 0) {
                // measured (scripts/xcd_reserve_ab.py): N = 4096 / 8192 / 16384 fit -3 / -9 / -6 %; with nb = 1024 the panel's
                // own products are too large for one or two XCDs and every setting is neutral or worse
                if (kbw <= 512) {
                    // (defaults 4096 / 8192 / 16384; 12288 .. 16384: the same to 1 %.  By CUs -- above cu_reserve_min_rows -- the
                    // two-unit tier ends at 6144 rows: the panel's products no longer carry idle workgroups, and the trailing
                    // update, which then bounds the step as often as the chain does, gets the CUs)
                    const int64_t rows2 = (cu_ok && remaining > ctx->cu_reserve_min_rows && ctx->reserve_rows2_cu < ctx->reserve_rows2) ? ctx->reserve_rows2_cu : ctx->reserve_rows2;
                    // (... and the one-unit tier at 12288: between 12288 and 16384 rows the trailing update bounds the step, and it
                    // is faster on the whole chip than on 7 / 8 of it even with the staged diagonal-block kernel next to it)
                    const int64_t rows1 = (cu_ok && ctx->reserve_rows1_cu < ctx->reserve_rows1) ? ctx->reserve_rows1_cu : ctx->reserve_rows1;
                    ctx->reserve_now = remaining <= ctx->reserve_rows4 ? 4 : (remaining <= rows2 ? 2 : (remaining <= rows1 ? 1 : 0));
                }
                else if (remaining <= ctx->xcd_reserve_big_rows) ctx->reserve_now = 1;  // (experiment, off by default: DESIGN.md section 5, round 5)
            } else {
                ctx->reserve_now = (int)ctx->xcd_reserve;  // explicit: that many XCDs for the whole factorisation
            }
            ctx->reserve_by_cu_now = cu_ok && ctx->reserve_now > 0 && remaining > ctx->cu_reserve_min_rows;
        }
    };
    // The FIRST panel too (round 5): nothing runs next to it but the early look-ahead update, which then keeps off like every
    // later one -- and the panel's diagonal blocks get the flat kernel (31 us) instead of the staged one that fits beside a GEMM
    // workgroup (48 us): 4 x 17 us of a fit of 2.3 ms at N = 4096
    // A panel factored by ONE resident launch (panel_chain = 2) has no "all but the last 128 columns are final" moment: the whole
    // look-ahead update (K = the panel's width) follows the launch on the PANEL stream, in front of ev_panel -- no stream hop between
    // a panel and the next, as with the K = 128 remainder below.
    bool ev_panel_early = false;
    static const int64_t la_overlap_rows = getenv("FRIEDRICH_AMD_LA_OVERLAP_ROWS") ? atoll(getenv("FRIEDRICH_AMD_LA_OVERLAP_ROWS")) : 4096;
    auto chain_panel = [&](int64_t kk, int64_t kbb) { return world == 1 && ctx->panel_chain == 2 && panel_chain_fits(ctx, kbb, n - kk, mode); };
    auto la_full_on_panel = [&](int64_t kk, int64_t kbb, bool hop) -> int {
        const int64_t after = n - (kk + kbb);
        if (after <= 0) return FR_OK;
        if (hop && (hipEventRecord(ctx->ev_la, S0) != hipSuccess || hipStreamWaitEvent(S1, ctx->ev_la, 0) != hipSuccess)) return FR_HIP_ERROR;
        hipStream_t saved = ctx->ls;
        ctx->ls = S1;
        GemmDesc g;
        g.M = after; g.N = width(after); g.K = kbb;
        g.A = A + (kk + kbb) + kk * ld; g.lda = ld; g.a_kmajor = false;
        g.B = g.A; g.ldb = ld; g.b_kmajor = false;
        g.D = A + (kk + kbb) + (kk + kbb) * ld; g.ldd = ld;
        g.Cin = g.D; g.ldcin = ld;
        g.alpha = -1.0; g.beta = 1.0; g.lower = false; g.prof_cls = FR_PROF_GEMM_PANEL;
        g.whole_chip = true;
        // (32-row tiles while at most 4096 rows remain -- N = 4096 / 8192 fits 1.91 / 5.81 -> 1.86 / 5.69 ms; up to 8192 rows: 6.0)
        static const int la_small = getenv("FRIEDRICH_AMD_LA_SMALL") ? atoi(getenv("FRIEDRICH_AMD_LA_SMALL")) : 4096;
        g.force_small = la_small != 0 && after <= (int64_t)la_small;
        const int st2 = launch_gemm(ctx, g);
        ctx->ls = saved;
        if (st2 == FR_OK) la_on_panel = true;
        return st2;
    };
    set_reservation(n, kb0);
    {
        if (split) {
            st = split_panel(ctx, A, ld, n, 0, kb0, col0, mode, sub, dinv, info, T, pbuf, split_rows, owner_of(0, nb, world));
        } else {
            la_hook(0, kb0);
            if (world == 1 || rank == owner_of(0, nb, world)) st = factor_panel(ctx, A, ld, n, 0, kb0, col0, mode, sub, dinv, info, T);
            ctx->cols_final_at = -1;
            if (st == FR_OK && ctx->reserve_now && !cu_reserve_active(ctx) && !(ctx->xcd_reserve < 0 && ctx->reserve_now == 4))
                st = launch_release_xcds(ctx, ctx->panel_epoch);
            if (st == FR_OK && world > 1) st = exchange_panel(ctx, A, ld, n, 0, kb0, dinv, pbuf, owner_of(0, nb, world));
        }
        if (st != FR_OK) return fail(st);
        if (chain_panel(0, kb0)) {
            // (while the trailing update bounds the step it starts WITH the look-ahead update, not behind it: ev_panel first)
            ev_panel_early = n - kb0 > la_overlap_rows;
            if (ev_panel_early && hipEventRecord(ctx->ev_panel, S1) != hipSuccess) return fail(FR_HIP_ERROR);
            st = la_full_on_panel(0, kb0, false);
            if (st != FR_OK) return fail(st);
        }
    }
    if (!ev_panel_early && hipEventRecord(ctx->ev_panel, S1) != hipSuccess) return fail(FR_HIP_ERROR);
    ctx->ls = S0;
    st = la_first_part(0, kb0);
    if (st != FR_OK) return fail(st);
    for (int64_t k = 0, kb = kb0, kb_next = 0; k < n; k += kb, kb = kb_next) {
        const int64_t rest = n - (k + kb);
        ctx->ls = S0;
        if (hipStreamWaitEvent(S0, ctx->ev_panel, 0) != hipSuccess) return fail(FR_HIP_ERROR);
        if (rest <= 0) break;
        const int64_t kb2 = kb_next = width(rest);
        const double* P = A + (k + kb) + k * ld;
        // once the panel chain is longer than the trailing update, the update's launches leave the panel stream's XCD(s) to it
        // (gemm_f64.hip / gemm_tile.hpp: claim_item)
        set_reservation(rest, kb2);
        const bool own_next = world == 1 || rank == owner_of(k + kb, nb, world);
        // look-ahead part of the trailing update: the next panel's columns (all rows below the current block)
        if (own_next && !la_on_panel) {
            // (profile class: panel -- the SYRK class times exactly the syrk_lower_f64_kernel launches)
            if (la_split)  // the first kb - 128 columns' part ran under the panel's last block: the last 128 columns remain
                st = gemm(ctx, FR_PROF_GEMM_PANEL, rest, kb2, IB, P + (kb - IB) * ld, ld, false, P + (kb - IB) * ld, ld, false, -1.0, 1.0,
                          A + (k + kb) + (k + kb) * ld, ld);
            else
                st = gemm(ctx, FR_PROF_GEMM_PANEL, rest, kb2, kb, P, ld, false, P, ld, false, -1.0, 1.0,
                          A + (k + kb) + (k + kb) * ld, ld);
            if (st != FR_OK) return fail(st);
        }
        // (with the whole look-ahead update already applied on the panel stream itself there is nothing to wait for: no hop)
        if (!la_on_panel && (hipEventRecord(ctx->ev_la, S0) != hipSuccess || hipStreamWaitEvent(S1, ctx->ev_la, 0) != hipSuccess))
            return fail(FR_HIP_ERROR);
        ctx->ls = S1;
        if (split) {
            st = split_panel(ctx, A, ld, n, k + kb, kb2, col0, mode, sub, dinv, info, T, pbuf, split_rows, owner_of(k + kb, nb, world));
        } else {
            la_hook(k + kb, kb2);
            if (own_next) st = factor_panel(ctx, A, ld, n, k + kb, kb2, col0, mode, sub, dinv, info, T);
            ctx->cols_final_at = -1;
            // (on the panel stream.  By CUs there is nothing to hand back: a launch keeps off for as long as it runs.  With four
            // XCDs set aside -- at most 4096 trailing rows -- the trailing update is at most 528 tiles on half the chip, ~120 us
            // against a chain of >= 260: it is long done, and the launch with its dispatch bubble, 11 us between the panel's
            // last solve and the look-ahead remainder, hands back nothing)
            if (st == FR_OK && ctx->reserve_now && !cu_reserve_active(ctx) && !(ctx->xcd_reserve < 0 && ctx->reserve_now == 4))
                st = launch_release_xcds(ctx, ctx->panel_epoch);
            if (st == FR_OK && world > 1) st = exchange_panel(ctx, A, ld, n, k + kb, kb2, dinv, pbuf, owner_of(k + kb, nb, world));
        }
        if (st != FR_OK) return fail(st);
        ctx->ls = S0;
        const int64_t rest2 = rest - kb2;
        if (rest2 > 0) {
            // K6: the rest of the trailing update runs under the next panel; multi-GPU: owned block columns only
            const double* P2 = P + kb2;
            GemmDesc g;
            g.M = rest2; g.N = rest2; g.K = kb;
            g.A = P2; g.lda = ld; g.a_kmajor = false;
            g.B = P2; g.ldb = ld; g.b_kmajor = false;
            g.D = A + (k + kb + kb2) + (k + kb + kb2) * ld; g.ldd = ld;
            g.Cin = g.D; g.ldcin = ld;
            g.alpha = -1.0; g.beta = 1.0; g.lower = true; g.prof_cls = FR_PROF_SYRK;
            g.own_world = world; g.own_rank = rank; g.own_nb = nb; g.own_col0 = k + kb + kb2;
            st = launch_gemm(ctx, g);
            if (st != FR_OK) return fail(st);
        }
        st = la_first_part(k + kb, kb2);  // (behind the trailing update on this stream; waits for the panel stream's ev_cols)
        if (st != FR_OK) return fail(st);
        la_on_panel = false;
        ev_panel_early = false;
        if (la_split && ctx->reserve_now > 0) {
            // Chain-bound: the K = 128 remainder of the look-ahead update goes on the PANEL stream, right behind the panel it
            // completes and in front of ev_panel -- the next panel then starts in stream order, without the two stream hops
            // (18 + 14 us per panel at N = 8192).  The trailing update of the panel before was launched at the start of this
            // round and is long done (chain-bound), so the product has the chip to itself; and the trailing update with THIS
            // panel starts behind ev_panel, i.e. behind it.  (Round 3's first attempt put it on the panel stream AFTER
            // ev_panel: it then raced the trailing update for the CUs and the fit got slower.)
            if (hipEventRecord(ctx->ev_la, S0) != hipSuccess || hipStreamWaitEvent(S1, ctx->ev_la, 0) != hipSuccess) return fail(FR_HIP_ERROR);
            ctx->ls = S1;
            const int64_t kk = k + kb, after = n - (kk + kb2);
            GemmDesc g;
            g.M = after; g.N = width(after); g.K = IB;
            g.A = A + (kk + kb2) + (kk + kb2 - IB) * ld; g.lda = ld; g.a_kmajor = false;
            g.B = g.A; g.ldb = ld; g.b_kmajor = false;
            g.D = A + (kk + kb2) + (kk + kb2) * ld; g.ldd = ld;
            g.Cin = g.D; g.ldcin = ld;
            g.alpha = -1.0; g.beta = 1.0; g.lower = false; g.prof_cls = FR_PROF_GEMM_PANEL;
            g.whole_chip = true;  // (on the panel stream's own XCDs only it takes twice as long: 8.5 vs 8.1 ms per fit at N = 8192)
            st = launch_gemm(ctx, g);
            if (st != FR_OK) return fail(st);
            la_on_panel = true;
            la_split = false;
            ctx->ls = S0;
        } else if (chain_panel(k + kb, kb2)) {
            ev_panel_early = n - (k + kb + kb2) > la_overlap_rows;
            if (ev_panel_early && hipEventRecord(ctx->ev_panel, S1) != hipSuccess) return fail(FR_HIP_ERROR);
            st = la_full_on_panel(k + kb, kb2, true);
            if (st != FR_OK) return fail(st);
        }
        if (!ev_panel_early && hipEventRecord(ctx->ev_panel, S1) != hipSuccess) return fail(FR_HIP_ERROR);
    }
    ctx->ls = S0;
    ctx->reserve_now = 0;
    ctx->reserve_by_cu_now = false;
    ctx->cols_final_at = -1;
    return FR_OK;
}

static inline int64_t split128(int64_t n)
{
    const int64_t nbk = (n + IB - 1) / IB;
    return (nbk / 2) * IB;
}

// ---- explicit inverses of the 512 x 512 diagonal blocks ----------------------------------------------------------------
// A triangular solve is a chain of n / 128 dependent leaf products against the 128-block inverses, each a ~25 us launch
// on 1/8 of the chip; with the updates of the two smallest recursion levels they made 20 % of a solve at N = 32768,
// m = 4096.  The wide solves (predict / predict_variance / sample_at / solve) use 512-row leaves instead: W_q = inverse of
// the q-th 512 x 512 diagonal block of L, assembled from the 128-block inverses by the 2 x 2 block formula
// inv([[P, 0], [C, Q]]) = [[P^-1, 0], [-Q^-1 C P^-1, Q^-1]], twice, as six batched GEMM launches for ALL blocks
// (~0.3 ms).  Built on first use after the factor changed; only blocks lying entirely inside the matrix.
constexpr int64_t LB = 512;  // leaf rows of the wide solves

static int batched_gemm(fr_ctx* ctx, int cls, int64_t M, int64_t N, int64_t K, const double* A, int64_t lda, bool a_kmajor,
                        int64_t sa, const double* B, int64_t ldb, bool b_kmajor, int64_t sb, double alpha, double* D,
                        int64_t ldd, int64_t sd, int64_t batch)
{
    GemmDesc g;
    g.M = M; g.N = N; g.K = K;
    g.A = A; g.lda = lda; g.a_kmajor = a_kmajor;
    g.B = B; g.ldb = ldb; g.b_kmajor = b_kmajor;
    g.Cin = D; g.ldcin = ldd; g.D = D; g.ldd = ldd;
    g.alpha = alpha; g.beta = 0.0; g.lower = false; g.prof_cls = cls;
    g.batch = batch; g.batch_a = sa; g.batch_b = sb; g.batch_c = sd; g.batch_d = sd;
    return launch_gemm(ctx, g);
}

static int ensure_inv512(fr_ctx* ctx, const fr_chol* cc, int cls)
{
    fr_chol* c = const_cast<fr_chol*>(cc);  // a cache: logically const
    const int64_t nq = c->n / LB;
    if (nq <= 0 || c->inv512_rows >= nq * LB) return FR_OK;
    if (c->inv512_cap < nq) {
        // (grown: the blocks already built are copied over -- add_rows extends the cache by one block per 512 rows)
        const int64_t cap = imax(nq, c->capacity / LB);
        double* fresh = nullptr;
        FR_HIP(ctx, dev_malloc(ctx, (void**)&fresh, sizeof(double) * (size_t)cap * LB * LB));
        if (c->inv512 && c->inv512_rows > 0)
            FR_HIP(ctx, hipMemcpyAsync(fresh, c->inv512, sizeof(double) * (size_t)(c->inv512_rows / LB) * LB * LB, hipMemcpyDeviceToDevice, ctx->ls));
        if (c->inv512) {
            FR_HIP(ctx, hipStreamSynchronize(ctx->ls));
            (void)hipFree(c->inv512);
        }
        c->inv512 = fresh;
        c->inv512_cap = cap;
    }
    // only the blocks that are not valid yet (after add_rows: the new ones)
    const int64_t q0 = c->inv512_rows / LB, nn = nq - q0;
    const int64_t ld = c->ld_a;
    WsGuard w(ctx);
    double* T = w.get(sizeof(double) * (size_t)nn * 256 * 256);
    if (!T) return FR_OUT_OF_MEMORY;
    double* W = c->inv512 + q0 * LB * LB;
    const double* A0 = c->A + q0 * LB + q0 * LB * ld;
    const double* D0 = c->dinv + 4 * q0 * INV_ELEMS;
    FR_TRY(launch_blockdiag512(ctx, D0, W, nn));
    // level 256: for both 256-halves (par) of every block q, 128-blocks a (upper) and b = a + 1:  W_ba = -X_b (L_ba X_a)
    for (int par = 0; par < 2; ++par) {
        const int64_t off = 256 * par;  // first row of the half inside its 512-block
        FR_TRY(batched_gemm(ctx, cls, IB, IB, IB, A0 + (off + IB) + off * ld, ld, false, LB + LB * ld,
                            D0 + (2 * par) * INV_ELEMS, IB, true, 4 * INV_ELEMS, 1.0, T, IB, INV_ELEMS, nn));
        FR_TRY(batched_gemm(ctx, cls, IB, IB, IB, D0 + (2 * par + 1) * INV_ELEMS, IB, false, 4 * INV_ELEMS, T, IB, true,
                            INV_ELEMS, -1.0, W + (off + IB) + off * LB, LB, LB * LB, nn));
    }
    // level 512: halves P (rows 0..255) and Q (256..511) of every block:  W_QP = -W_QQ (L_QP W_PP)
    FR_TRY(batched_gemm(ctx, cls, 256, 256, 256, A0 + 256, ld, false, LB + LB * ld, W, LB, true, LB * LB, 1.0, T, 256,
                        256 * 256, nn));
    FR_TRY(batched_gemm(ctx, cls, 256, 256, 256, W + 256 + 256 * LB, LB, false, LB * LB, T, 256, true, 256 * 256, -1.0,
                        W + 256, LB, LB * LB, nn));
    c->inv512_rows = nq * LB;
    return FR_OK;
}

// ---- explicit inverses of the 2048 x 2048 diagonal blocks ----------------------------------------------------------------
// A solve with a few hundred right-hand sides is neither a bandwidth problem (the persistent column-group kernel streams the
// factor once per 16 columns and lives on a chain of n / 128 hand-offs) nor yet a throughput problem: the recursion over
// 512-row leaves is ~6 n / 1024 dependent launches of 25 - 45 us that use a quarter of the chip each (n = 8192, 512 columns:
// 47 launches, 1.1 ms for 3.4e10 flop; configs[4]'s L21 solves, sample_at(256), predict_variance(1024)).  Fewer, fatter links:
// 2048-row diagonal blocks with an explicit inverse (two more block levels on top of the 512 ones, the same 2 x 2 formula),
// applied as ONE triangular-operand product per block on 32-row tiles claimed in dispatch order, and a LEFT-looking sweep
// between them -- one deep update (contraction = all rows solved so far: split along K) per block: 2 n / 2048 - 1 launches.
constexpr int64_t GB = 2048;

static int ensure_invbig(fr_ctx* ctx, const fr_chol* cc, int cls)
{
    fr_chol* c = const_cast<fr_chol*>(cc);
    const int64_t ng = c->n / GB;
    if (ng <= 0 || c->invbig_rows >= ng * GB) return FR_OK;
    FR_TRY(ensure_inv512(ctx, c, cls));
    if (c->invbig_cap < ng) {
        const int64_t cap = imax(ng, c->capacity / GB);
        double* fresh = nullptr;
        FR_HIP(ctx, dev_malloc(ctx, (void**)&fresh, sizeof(double) * (size_t)cap * GB * GB));
        if (c->invbig && c->invbig_rows > 0)
            FR_HIP(ctx, hipMemcpyAsync(fresh, c->invbig, sizeof(double) * (size_t)(c->invbig_rows / GB) * GB * GB, hipMemcpyDeviceToDevice, ctx->ls));
        if (c->invbig) {
            FR_HIP(ctx, hipStreamSynchronize(ctx->ls));
            (void)hipFree(c->invbig);
        }
        c->invbig = fresh;
        c->invbig_cap = cap;
    }
    const int64_t g0 = c->invbig_rows / GB, nn = ng - g0;
    const int64_t ld = c->ld_a;
    WsGuard w(ctx);
    double* T = w.get(sizeof(double) * (size_t)nn * 1024 * 1024);
    if (!T) return FR_OUT_OF_MEMORY;
    double* W = c->invbig + g0 * GB * GB;
    const double* A0 = c->A + g0 * GB + g0 * GB * ld;
    // zero (the products below read whole sub-blocks), then the four 512-block inverses of every block on its diagonal
    FR_HIP(ctx, hipMemsetAsync(W, 0, sizeof(double) * (size_t)nn * GB * GB, ctx->ls));
    for (int64_t b = 0; b < nn; ++b)
        for (int j = 0; j < 4; ++j)
            FR_TRY(launch_copy(ctx, c->inv512 + (4 * (g0 + b) + j) * LB * LB, LB, W + b * GB * GB + j * LB + j * LB * GB, GB, LB, LB));
    // level 1024: both 1024-halves (par) of every block, 512-blocks P (upper) and Q:  W_QP = -W_QQ (L_QP W_PP)
    for (int par = 0; par < 2; ++par) {
        const int64_t off = 1024 * par;
        FR_TRY(batched_gemm(ctx, cls, LB, LB, LB, A0 + (off + LB) + off * ld, ld, false, GB + GB * ld, W + off + off * GB, GB, true,
                            GB * GB, 1.0, T, LB, LB * LB, nn));
        FR_TRY(batched_gemm(ctx, cls, LB, LB, LB, W + (off + LB) + (off + LB) * GB, GB, false, GB * GB, T, LB, true, LB * LB, -1.0,
                            W + (off + LB) + off * GB, GB, GB * GB, nn));
    }
    // level 2048: halves P (rows 0 .. 1023) and Q (1024 .. 2047)
    FR_TRY(batched_gemm(ctx, cls, 1024, 1024, 1024, A0 + 1024, ld, false, GB + GB * ld, W, GB, true, GB * GB, 1.0, T, 1024,
                        1024 * 1024, nn));
    FR_TRY(batched_gemm(ctx, cls, 1024, 1024, 1024, W + 1024 + 1024 * GB, GB, false, GB * GB, T, 1024, true, 1024 * 1024, -1.0,
                        W + 1024, GB, GB * GB, nn));
    c->invbig_rows = ng * GB;
    return FR_OK;
}

// one 512-row leaf: B (512 x m) <- W B  (forward) or W^T B (backward); not in place (four tile rows): through a copy
static int leaf512(fr_ctx* ctx, const fr_chol* c, int64_t row0, double* B, int64_t m, int64_t ldb, int cls, bool fwd,
                   double* tmp)
{
    const double* W = c->inv512 + (row0 / LB) * LB * LB;
    FR_TRY(launch_copy(ctx, B, ldb, tmp, LB, LB, m));
    return gemm(ctx, cls, LB, m, LB, W, LB, !fwd, tmp, LB, true, 1.0, 0.0, B, ldb);
}

// One 128-row leaf with a step of iterative refinement (handles with fr_chol::refine):  X0 = W B;  R = B - L_bb X0;
// X = X0 + W R  (forward; backward with the transposes).  tmp: n x m scratch (ld 128).
static int refined_leaf(fr_ctx* ctx, const fr_chol* c, int64_t row0, int64_t n, double* B, int64_t m, int64_t ldb, int cls,
                        bool fwd, double* tmp)
{
    const double* W = c->dinv + (row0 / IB) * INV_ELEMS;
    const double* Lbb = c->A + row0 + row0 * c->ld_a;
    GemmDesc g;
    g.M = n; g.N = m; g.K = n;
    g.A = W; g.lda = IB; g.a_kmajor = !fwd;
    g.B = B; g.ldb = ldb; g.b_kmajor = true;
    g.Cin = tmp; g.ldcin = IB; g.D = tmp; g.ldd = IB;
    g.alpha = 1.0; g.beta = 0.0; g.lower = false; g.prof_cls = cls;
    FR_TRY(launch_gemm(ctx, g));  // X0 -> tmp
    FR_TRY(gemm(ctx, cls, n, m, n, Lbb, c->ld_a, !fwd, tmp, IB, true, -1.0, 1.0, B, ldb));  // R = B - op(L_bb) X0 (in B)
    g.Cin = tmp; g.ldcin = IB; g.D = B; g.ldd = ldb;
    g.beta = 1.0;
    return launch_gemm(ctx, g);  // B <- X0 + op(W) R   (in place: one tile row)
}

// B (n x m) <- L^-1 B.  row0 = first row of this sub-problem in the factor; tmp != nullptr enables the 512-row leaves
static int trsm_fwd_rec(fr_ctx* ctx, const fr_chol* c, int64_t row0, int64_t n, double* B, int64_t m, int64_t ldb, int cls,
                        double* tmp)
{
    const double* L = c->A + row0 + row0 * c->ld_a;
    const int64_t ld = c->ld_a;
    if (!c->refine && tmp && n == LB && row0 % LB == 0 && row0 + LB <= c->inv512_rows) return leaf512(ctx, c, row0, B, m, ldb, cls, true, tmp);
    if (n <= IB) {
        const double* W = c->dinv + (row0 / IB) * INV_ELEMS;
        if (c->refine && tmp) return refined_leaf(ctx, c, row0, n, B, m, ldb, cls, true, tmp);
        return gemm(ctx, cls, n, m, n, W, IB, false, B, ldb, true, 1.0, 0.0, B, ldb);
    }
    // split at a multiple of 512 while the problem is larger than a leaf (so that the leaves line up with the blocks)
    const int64_t n1 = (!c->refine && tmp && n > LB) ? (((n + LB - 1) / LB) / 2) * LB : split128(n);
    FR_TRY(trsm_fwd_rec(ctx, c, row0, n1, B, m, ldb, cls, tmp));
    FR_TRY(gemm(ctx, cls, n - n1, m, n1, L + n1, ld, false, B, ldb, true, -1.0, 1.0, B + n1, ldb));
    return trsm_fwd_rec(ctx, c, row0 + n1, n - n1, B + n1, m, ldb, cls, tmp);
}

// B (n x m) <- L^-T B
static int trsm_bwd_rec(fr_ctx* ctx, const fr_chol* c, int64_t row0, int64_t n, double* B, int64_t m, int64_t ldb, int cls,
                        double* tmp)
{
    const double* L = c->A + row0 + row0 * c->ld_a;
    const int64_t ld = c->ld_a;
    if (!c->refine && tmp && n == LB && row0 % LB == 0 && row0 + LB <= c->inv512_rows) return leaf512(ctx, c, row0, B, m, ldb, cls, false, tmp);
    if (n <= IB) {
        const double* W = c->dinv + (row0 / IB) * INV_ELEMS;
        if (c->refine && tmp) return refined_leaf(ctx, c, row0, n, B, m, ldb, cls, false, tmp);
        return gemm(ctx, cls, n, m, n, W, IB, true, B, ldb, true, 1.0, 0.0, B, ldb);
    }
    const int64_t n1 = (!c->refine && tmp && n > LB) ? (((n + LB - 1) / LB) / 2) * LB : split128(n);
    FR_TRY(trsm_bwd_rec(ctx, c, row0 + n1, n - n1, B + n1, m, ldb, cls, tmp));
    // B1 -= L21^T * B2    (op(A)[m][k] = L21[k][m]: k-major)
    FR_TRY(gemm(ctx, cls, n1, m, n - n1, L + n1, ld, true, B + n1, ldb, true, -1.0, 1.0, B, ldb));
    return trsm_bwd_rec(ctx, c, row0, n1, B, m, ldb, cls, tmp);
}

// X (k x n) <- X L^-T
static int trsm_right_rec(fr_ctx* ctx, const double* L, int64_t ld, const double* dinv, int64_t n, double* X, int64_t k,
                          int64_t ldx, int cls)
{
    if (n <= IB) return gemm(ctx, cls, k, n, n, X, ldx, false, dinv, IB, false, 1.0, 0.0, X, ldx);
    const int64_t n1 = split128(n);
    FR_TRY(trsm_right_rec(ctx, L, ld, dinv, n1, X, k, ldx, cls));
    // X2 -= X1 * L21^T
    FR_TRY(gemm(ctx, cls, k, n - n1, n1, X, ldx, false, L + n1, ld, false, -1.0, 1.0, X + n1 * ldx, ldx));
    return trsm_right_rec(ctx, L + n1 + n1 * ld, ld, dinv + (n1 / IB) * INV_ELEMS, n - n1, X + n1 * ldx, k, ldx, cls);
}

// one 2048-row leaf: B (2048 x m) <- W B (forward) / W^T B (backward) through a copy; the product skips the structural zeros
// of the triangular inverse tile by tile
static int leaf_big(fr_ctx* ctx, const fr_chol* c, int64_t row0, double* B, int64_t m, int64_t ldb, int cls, bool fwd, double* tmp,
                    bool tmp_holds_operand)
{
    const double* W = c->invbig + (row0 / GB) * GB * GB;
    if (!tmp_holds_operand) FR_TRY(launch_copy(ctx, B, ldb, tmp, GB, GB, m));
    GemmDesc g;
    g.M = GB; g.N = m; g.K = GB;
    g.A = W; g.lda = GB; g.a_kmajor = !fwd;
    g.B = tmp; g.ldb = GB; g.b_kmajor = true;
    g.Cin = B; g.ldcin = ldb; g.D = B; g.ldd = ldb;
    g.alpha = 1.0; g.beta = 0.0; g.lower = false; g.prof_cls = cls;
    g.tri = fwd ? 4 : 1;  // W (r, k) = 0 for k > r  /  W^T (r, k) = 0 for k < r
    // Two variants, by measurement (scripts/solve_mid.py, forward solves at N = 4096 / 8192 in one process):
    //  * up to 640 columns: cut along K like the other products with few result tiles (slices that only meet structural zeros
    //    write zeros and retire) -- 256 columns 0.37 / 0.92 -> 0.28 / 0.75 ms, 512 columns equal;
    //  * above: 32-row tiles in mirrored pairs -- a workgroup takes row tile i and then row tile 63 - i, a contraction of 2080
    //    for every workgroup -- 1024 columns 0.59 / 1.76 -> 0.50 / 1.61 ms (148 against 166 + 13 us per leaf; claimed one by
    //    one in dispatch order the 32-row tiles took 180 us: the launch is as long as its deepest tiles).  Both run at half
    //    the rate of the dense updates: the paired tiles are bound by the L2 (a 32 x 128 tile moves 20 KiB per 131 kflop), the
    //    slices by the workgroups that hold two deep ones on one CU.  FRIEDRICH_AMD_LEAF_MIRROR = 0 / 1 forces a variant.
    //    (Also measured: 128-row tiles claimed in dispatch order, deepest contractions first, where 16 x m / 128 of them fill the
    //    chip -- 2048 / 4096 columns at N = 8192 / 32768: 2.84 / 66.6 ms against 2.64 / 64.3 with the mirrored pairs; and 128-row
    //    tiles in mirrored pairs, 8 pairs x m / 128 workgroups with a contraction of 2176 each -- 4096 columns at N = 16384 /
    //    32768: 16.6 / 64.3 ms against 16.7 / 64.5, 2048 columns 9.8 against 8.6.  Neither kept.)
    static const int leaf_force = getenv("FRIEDRICH_AMD_LEAF_MIRROR") ? atoi(getenv("FRIEDRICH_AMD_LEAF_MIRROR")) : -1;
    const int variant = leaf_force >= 0 ? leaf_force : (m > 640 ? 1 : 0);
    if (variant == 1) {
        g.force_small = true;
        g.mirror = true;
    } else {
        g.tri_splitk = true;
    }
    return launch_gemm(ctx, g);
}

// T (rows x m, ld GB) = Bblk - op(A) X: the left-looking update of a block lands in the leaf's scratch directly (no copy)
static int update_into(fr_ctx* ctx, int cls, int64_t rows, int64_t m, int64_t K, const double* A, int64_t lda, bool a_kmajor, const double* X,
                       int64_t ldx, const double* Bblk, int64_t ldb, double* T)
{
    GemmDesc g;
    g.M = rows; g.N = m; g.K = K;
    g.A = A; g.lda = lda; g.a_kmajor = a_kmajor;
    g.B = X; g.ldb = ldx; g.b_kmajor = true;
    g.Cin = Bblk; g.ldcin = ldb; g.D = T; g.ldd = GB;
    g.alpha = -1.0; g.beta = 1.0; g.lower = false; g.prof_cls = cls;
    return launch_gemm(ctx, g);
}

// B (n x m) <- L^-1 B (fwd) / L^-T B, left-looking over 2048-row blocks; the rows behind the last whole block go through the
// 512-leaf recursion (tmp: 2048 x m scratch)
static int trsm_big(fr_ctx* ctx, const fr_chol* c, int64_t n, double* B, int64_t m, int64_t ldb, int cls, bool fwd, double* tmp)
{
    const int64_t ld = c->ld_a;
    const int64_t ng = c->invbig_rows / GB < n / GB ? c->invbig_rows / GB : n / GB;
    const int64_t n0 = ng * GB, tail = n - n0;
    if (fwd) {
        for (int64_t b = 0; b < ng; ++b) {
            const int64_t r0 = b * GB;
            if (b > 0) FR_TRY(update_into(ctx, cls, GB, m, r0, c->A + r0, ld, false, B, ldb, B + r0, ldb, tmp));
            FR_TRY(leaf_big(ctx, c, r0, B + r0, m, ldb, cls, true, tmp, b > 0));
        }
        if (tail > 0) {
            if (n0 > 0) FR_TRY(gemm(ctx, cls, tail, m, n0, c->A + n0, ld, false, B, ldb, true, -1.0, 1.0, B + n0, ldb));
            FR_TRY(trsm_fwd_rec(ctx, c, n0, tail, B + n0, m, ldb, cls, tail >= LB ? tmp : nullptr));
        }
        return FR_OK;
    }
    if (tail > 0) FR_TRY(trsm_bwd_rec(ctx, c, n0, tail, B + n0, m, ldb, cls, tail >= LB ? tmp : nullptr));
    for (int64_t b = ng - 1; b >= 0; --b) {
        const int64_t r0 = b * GB, below = n - (r0 + GB);
        // B_b - L[rows below, block b]^T X[rows below]     (op(A)[m][k] = L[k][m]: k-major)
        if (below > 0) FR_TRY(update_into(ctx, cls, GB, m, below, c->A + (r0 + GB) + r0 * ld, ld, true, B + r0 + GB, ldb, B + r0, ldb, tmp));
        FR_TRY(leaf_big(ctx, c, r0, B + r0, m, ldb, cls, false, tmp, below > 0));
    }
    return FR_OK;
}

// When the 2048-row leaves are taken.  Measured, forward solve of a device operand (scripts/solve_mid.py; "other" = what the
// library did before: column groups of 16 / the 512-leaf recursion / K9 up to 16 columns), milliseconds big | other:
//   N = 4096:   2 columns 0.120 | 0.192,  16: 0.124 | 0.209,  17: 0.124 | 0.310,  64: 0.140 | 0.308,  128: 0.141 | 0.313,  192: 0.199 | 0.355
//   N = 8192:   2: 0.316 | 0.384,  16: 0.319 | 0.384,  24: 0.336 | 0.634,  64: 0.361 | 0.628,  128: 0.368 | 0.710,  1024: 1.63 | 2.65 (variance)
//   N = 12288:  16: 0.584 | 0.596,  17: 0.593 | 0.972,  128: 0.645 | 1.40
//   N = 16384:  2: 0.875 | 0.790,  16: 0.910 | 0.796 (K9's chain wins),  24: 0.93 | 1.29,  128: 1.00 | 2.44,  1024: 5.11 | 7.38 (variance)
//   N = 32768:  24: 2.94 | 2.94,  32: 2.97 | 2.91,  48: 3.05 | 3.91,  128: 3.10 | 4.33,  4096: 64.3 | 72.8
// i.e. the time of a leaf sweep hardly depends on the number of columns up to 128 (2 n / 2048 - 1 launches of one tile column).
// The leaves need their inverse blocks first -- two more block levels per factor, about the cost of two such solves
// (ensure_invbig) -- so with fewer than 192 right-hand sides they are taken from the THIRD narrow solve against a factor on
// (or at once when the blocks exist: after add_rows the caches are only extended).
static bool use_big_leaves(fr_ctx* ctx, const fr_chol* cc, int64_t n, int64_t m, bool count)
{
    fr_chol* c = const_cast<fr_chol*>(cc);  // (the solve counter is a cache statistic: logically const)
    const int64_t mmax = ctx->bigleaf_max >= 0 ? ctx->bigleaf_max : 4096;
    if (!(ctx->leaf512 != 0 && !c->refine && n == c->n && n >= 2 * GB && m >= 2 && m <= mmax)) return false;
    if (ctx->bigleaf_min >= 0) return m >= ctx->bigleaf_min;  // explicit (A/B runs)
    if (m <= 16 && n >= 12288) return false;                  // K9
    if (m < 48 && n >= 24576) return false;                   // column groups: the same time, no blocks to build
    if (m >= 192) return true;
    const bool ready = c->invbig_rows > 0 && c->invbig_rows + GB >= (n / GB) * GB;  // (at most one block to add)
    if (ready) return true;
    if (count) ++c->narrow_solves;
    return c->narrow_solves >= 3;
}


// ---- a few right-hand sides (likelihood, K^-1 y, predicting a handful of points) -----------------------------------------
// Same recursion on memory-bound kernels that read L once per 16 columns (a matrix-vector kernel for one column): the GEMM's
// 128-wide tiles would be mostly padding.  Leaves against the 512-block inverses, out of place through `tmp` (512 x m).
static int narrow_fwd_rec(fr_ctx* ctx, const fr_chol* c, int64_t row0, int64_t n, double* b, int64_t m, int64_t ldb,
                          double* tmp)
{
    const int64_t ld = c->ld_a;
    const double* L = c->A + row0 + row0 * ld;
    const bool leaf512_ok = n == LB && row0 % LB == 0 && row0 + LB <= c->inv512_rows;
    if (leaf512_ok || n <= IB) {
        const double* W = leaf512_ok ? c->inv512 + (row0 / LB) * LB * LB : c->dinv + (row0 / IB) * INV_ELEMS;
        const int64_t ldw = leaf512_ok ? LB : IB;
        FR_TRY(launch_copy(ctx, b, ldb, tmp, LB, n, m));
        if (m == 1) return launch_gemv_n(ctx, W, n, n, ldw, tmp, 1.0, 0.0, b);
        return launch_skinny_n(ctx, W, n, n, ldw, tmp, LB, m, 1.0, 0.0, b, ldb);
    }
    const int64_t n1 = (n > LB) ? (((n + LB - 1) / LB) / 2) * LB : split128(n);
    FR_TRY(narrow_fwd_rec(ctx, c, row0, n1, b, m, ldb, tmp));
    if (m == 1)
        FR_TRY(launch_gemv_n(ctx, L + n1, n - n1, n1, ld, b, -1.0, 1.0, b + n1));
    else
        FR_TRY(launch_skinny_n(ctx, L + n1, n - n1, n1, ld, b, ldb, m, -1.0, 1.0, b + n1, ldb));
    return narrow_fwd_rec(ctx, c, row0 + n1, n - n1, b + n1, m, ldb, tmp);
}

static int narrow_bwd_rec(fr_ctx* ctx, const fr_chol* c, int64_t row0, int64_t n, double* b, int64_t m, int64_t ldb,
                          double* tmp)
{
    const int64_t ld = c->ld_a;
    const double* L = c->A + row0 + row0 * ld;
    const bool leaf512_ok = n == LB && row0 % LB == 0 && row0 + LB <= c->inv512_rows;
    if (leaf512_ok || n <= IB) {
        const double* W = leaf512_ok ? c->inv512 + (row0 / LB) * LB * LB : c->dinv + (row0 / IB) * INV_ELEMS;
        const int64_t ldw = leaf512_ok ? LB : IB;
        FR_TRY(launch_copy(ctx, b, ldb, tmp, LB, n, m));
        if (m == 1) return launch_gemv_t(ctx, W, n, n, ldw, tmp, 1.0, 0.0, b);  // b = W^T tmp
        return launch_skinny_t(ctx, W, n, n, ldw, tmp, LB, m, 1.0, 0.0, b, ldb);
    }
    const int64_t n1 = (n > LB) ? (((n + LB - 1) / LB) / 2) * LB : split128(n);
    FR_TRY(narrow_bwd_rec(ctx, c, row0 + n1, n - n1, b + n1, m, ldb, tmp));
    if (m == 1)
        FR_TRY(launch_gemv_t(ctx, L + n1, n - n1, n1, ld, b + n1, -1.0, 1.0, b));  // b1 -= L21^T b2
    else
        FR_TRY(launch_skinny_t(ctx, L + n1, n - n1, n1, ld, b + n1, ldb, m, -1.0, 1.0, b, ldb));
    return narrow_bwd_rec(ctx, c, row0, n1, b, m, ldb, tmp);
}

static int narrow_solve(fr_ctx* ctx, const fr_chol* c, int64_t n, double* b, int64_t m, int64_t ldb, int cls, bool fwd)
{
    WsGuard w(ctx);
    double* tmp = w.get(sizeof(double) * (size_t)LB * (size_t)m);
    if (!tmp) return FR_OUT_OF_MEMORY;
    if (n >= 2 * LB) FR_TRY(ensure_inv512(ctx, c, cls));
    return fwd ? narrow_fwd_rec(ctx, c, 0, n, b, m, ldb, tmp) : narrow_bwd_rec(ctx, c, 0, n, b, m, ldb, tmp);
}

// 512-row leaves: fewer, larger links in the chain of dependent launches (measured down to 16 right-hand sides:
// scripts/narrow_predict_ab.py); a single right-hand side takes the matrix-vector path above
static bool wide_solve(const fr_ctx* ctx, const fr_chol* c, int64_t n, int64_t m)
{
    return ctx->leaf512 != 0 && n == c->n && n >= 2 * LB && m >= 2;
}

// The persistent matrix-core solve in column groups of 16 (trsm_narrow.hip) against the recursive GEMM formulation, by
// measurement (scripts/narrow_batched_ab.py, forward solve): N = 32768: 3.5 vs 14.3 ms at 32 columns, 9.8 vs 12.2 at 128,
// 18 vs 14.5 at 256;  N = 8192: 0.64 vs 3.1 ms at 32, 1.8 vs 2.8 at 256, 3.3 vs 3.0 at 512 (every group re-reads the
// factor from L2 / Infinity Cache, the GEMMs amortise it over 128 columns but are a chain of ~100 launches).
static bool use_column_groups(const fr_ctx* ctx, const fr_chol* c, int64_t n, int64_t m)
{
    if (n != c->n || !ctx->trsv || m < 2) return false;
    if (m <= 16) return m <= ctx->narrow_max;
    if (ctx->narrow_batched_max >= 0) return m <= ctx->narrow_batched_max;  // explicit setting
    // measured (scripts/narrow_batched_ab.py) after the split-K rule let the GEMM path's lower levels fill the chip: the column
    // groups win up to m = 256 / 192 / 96 / 64 columns at n = 4096 / 8192 / 16384 / 32768 (e.g. n = 32768: m = 64 4.9 vs 6.4 ms,
    // m = 128 7.7 vs 6.0 ms), i.e. while m n stays below about two million
    // ... and for any number of columns while m n <= 1.3e6 (n = 512 / 1024 / 2048: up to 1024 / 1024 / 512 columns, e.g. the
    // n x n identity of the gradient's small cases: 0.13 vs 0.29 ms at n = 512)
    if ((m <= 256 && m * n <= 2200000) || m * n <= 1300000) return true;
    // the GEMM path's 512-row leaves need the 512 x 512 inverse blocks, which a changed factor has to rebuild first (seven
    // launches): a one-off solve right after add_rows -- its own L21 solve above all -- stays with the groups (configs[4]:
    // eight appends of 512 rows 15.6 -> 15.0 ms)
    return m <= 512 && n <= 8192 && c->inv512_rows < (n / LB) * LB;
}

int trsm_lower_fwd(fr_ctx* ctx, const fr_chol* c, int64_t n, double* B, int64_t m, int64_t ldb, int cls)
{
    if (n <= 0 || m <= 0) return FR_OK;
    if (c->refine) {  // ill-conditioned diagonal blocks: 128-row leaves with a refinement step each
        WsGuard w(ctx);
        double* tmp = w.get(sizeof(double) * (size_t)IB * (size_t)m);
        if (!tmp) return FR_OUT_OF_MEMORY;
        return trsm_fwd_rec(ctx, c, 0, n, B, m, ldb, cls, tmp);
    }
    if (m == 1 && n == c->n && ctx->trsv) return launch_trsv(ctx, c, B, true, cls);
    if (use_big_leaves(ctx, c, n, m, true) && !(ctx->narrow_batched_max > 0 && m <= ctx->narrow_batched_max)) {
        // (the rows behind the last whole 2048-block take the 512-row leaves).  The inverse blocks are a cache of n / 2048 x 32 MiB:
        // if it cannot be built (memory), the solve takes the paths below instead of failing
        WsGuard wb(ctx);
        double* tmpb = nullptr;
        if (ensure_inv512(ctx, c, cls) == FR_OK && ensure_invbig(ctx, c, cls) == FR_OK) {
            // (scratch of 2048 x m: every other path needs a workspace of that order too -- no fall-through on this one, so that
            // the only way to the persistent kernels below is a cache that could not be built, which solve_in_place rules out first)
            if ((tmpb = wb.get(sizeof(double) * (size_t)GB * (size_t)m)) == nullptr) return FR_OUT_OF_MEMORY;
            return trsm_big(ctx, c, n, B, m, ldb, cls, true, tmpb);
        }
        (void)hipGetLastError();
    }
    if (use_column_groups(ctx, c, n, m)) return launch_trsm_narrow(ctx, c, B, m, ldb, true, cls);
    if (m <= ctx->narrow_max && n == c->n && n >= 4 * IB) return narrow_solve(ctx, c, n, B, m, ldb, cls, true);
    WsGuard w(ctx);
    double* tmp = nullptr;
    if (wide_solve(ctx, c, n, m)) {
        FR_TRY(ensure_inv512(ctx, c, cls));
        tmp = w.get(sizeof(double) * (size_t)LB * (size_t)m);
        if (!tmp) return FR_OUT_OF_MEMORY;
    }
    return trsm_fwd_rec(ctx, c, 0, n, B, m, ldb, cls, tmp);
}

// ---- W = L^-1 for the gradient of the marginal likelihood (K^-1 = W^T W, grad.hip) -------------------------------------------
// The forward solve of the identity spends n^3 flops, two thirds of them on the structural zeros of W.  Recursively,
//   L = [L11 0; L21 L22]  ->  W = [W11 0; -W22 (L21 W11) W22],
// and the two products skip the zero halves of their triangular operand tile by tile (GemmArgs::tri): n^3 / 3 in all.
// Below TRINV_BASE rows a block is inverted by the ordinary solve of its identity (512-row leaves and all).
constexpr int64_t TRINV_BASE = 2048;

static int trinv_rec(fr_ctx* ctx, const fr_chol* c, int64_t row0, int64_t n, double* W, int64_t ldw, double* T, double* tmp, int cls)
{
    if (n <= TRINV_BASE) {
        // a whole 2048-row diagonal block whose explicit inverse is cached (ensure_invbig: two batched block levels for ALL blocks
        // of the factor, where the solve of the identity below is ~25 launches per block): the base case is a copy
        if (n == GB && row0 % GB == 0 && row0 + GB <= c->invbig_rows)
            return launch_copy(ctx, c->invbig + (row0 / GB) * GB * GB, GB, W, ldw, GB, GB);
        FR_TRY(launch_set_identity(ctx, W, n, ldw));
        return trsm_fwd_rec(ctx, c, row0, n, W, n, ldw, cls, tmp);
    }
    const int64_t n1 = (((n + LB - 1) / LB) / 2) * LB, n2 = n - n1;
    FR_TRY(trinv_rec(ctx, c, row0, n1, W, ldw, T, tmp, cls));
    FR_TRY(trinv_rec(ctx, c, row0 + n1, n2, W + n1 + n1 * ldw, ldw, T, tmp, cls));
    const int64_t ldt = round_up(n2, kAlign);
    GemmDesc g;
    // T = L21 W11: W11 (k, n) is zero for k < n
    g.M = n2; g.N = n1; g.K = n1;
    g.A = c->A + (row0 + n1) + row0 * c->ld_a; g.lda = c->ld_a; g.a_kmajor = false;
    g.B = W; g.ldb = ldw; g.b_kmajor = true;
    g.Cin = T; g.ldcin = ldt; g.D = T; g.ldd = ldt;
    g.alpha = 1.0; g.beta = 0.0; g.lower = false; g.prof_cls = cls; g.tri = 2; g.dynamic = true;
    FR_TRY(launch_gemm(ctx, g));
    // W21 = -W22 T: W22 (m, k) is zero for k > m
    g.M = n2; g.N = n1; g.K = n2;
    g.A = W + n1 + n1 * ldw; g.lda = ldw; g.a_kmajor = false;
    g.B = T; g.ldb = ldt; g.b_kmajor = true;
    g.Cin = W + n1; g.ldcin = ldw; g.D = W + n1; g.ldd = ldw;
    g.alpha = -1.0; g.beta = 0.0; g.tri = 4;
    return launch_gemm(ctx, g);
}

int chol_tri_inverse(fr_ctx* ctx, const fr_chol* c, double* W, int64_t ldw, double* T, int cls)
{
    const int64_t n = c->n;
    if (n <= 0) return FR_OK;
    if (c->refine || n <= TRINV_BASE || !ctx->tri_inverse) {  // (ill-conditioned handles keep the refined solve of the whole identity)
        FR_TRY(launch_set_identity(ctx, W, n, ldw));
        return trsm_lower_fwd(ctx, c, n, W, n, ldw, cls);
    }
    FR_HIP(ctx, hipMemsetAsync(W, 0, sizeof(double) * (size_t)ldw * (size_t)n, ctx->ls));
    FR_TRY(ensure_inv512(ctx, c, cls));
    // (n = 2048 x a power of two: the halving below ends in whole 2048-row diagonal blocks, whose inverses ensure_invbig builds
    // for the whole factor in six batched launches; best effort)
    if (n % GB == 0 && ((n / GB) & (n / GB - 1)) == 0 && n >= 2 * GB && ensure_invbig(ctx, c, cls) != FR_OK) (void)hipGetLastError();
    WsGuard w(ctx);
    double* tmp = w.get(sizeof(double) * (size_t)LB * (size_t)TRINV_BASE);
    if (!tmp) return FR_OUT_OF_MEMORY;
    return trinv_rec(ctx, c, 0, n, W, ldw, T, tmp, cls);
}

int trsm_lower_bwd(fr_ctx* ctx, const fr_chol* c, int64_t n, double* B, int64_t m, int64_t ldb, int cls)
{
    if (n <= 0 || m <= 0) return FR_OK;
    if (c->refine) {
        WsGuard w(ctx);
        double* tmp = w.get(sizeof(double) * (size_t)IB * (size_t)m);
        if (!tmp) return FR_OUT_OF_MEMORY;
        return trsm_bwd_rec(ctx, c, 0, n, B, m, ldb, cls, tmp);
    }
    if (m == 1 && n == c->n && ctx->trsv) return launch_trsv(ctx, c, B, false, cls);
    if (use_big_leaves(ctx, c, n, m, true) && !(ctx->narrow_batched_max > 0 && m <= ctx->narrow_batched_max)) {
        WsGuard wb(ctx);
        double* tmpb = nullptr;
        if (ensure_inv512(ctx, c, cls) == FR_OK && ensure_invbig(ctx, c, cls) == FR_OK) {
            if ((tmpb = wb.get(sizeof(double) * (size_t)GB * (size_t)m)) == nullptr) return FR_OUT_OF_MEMORY;
            return trsm_big(ctx, c, n, B, m, ldb, cls, false, tmpb);
        }
        (void)hipGetLastError();
    }
    if (use_column_groups(ctx, c, n, m)) return launch_trsm_narrow(ctx, c, B, m, ldb, false, cls);
    if (m <= ctx->narrow_max && n == c->n && n >= 4 * IB) return narrow_solve(ctx, c, n, B, m, ldb, cls, false);
    WsGuard w(ctx);
    double* tmp = nullptr;
    if (wide_solve(ctx, c, n, m)) {
        FR_TRY(ensure_inv512(ctx, c, cls));
        tmp = w.get(sizeof(double) * (size_t)LB * (size_t)m);
        if (!tmp) return FR_OUT_OF_MEMORY;
    }
    return trsm_bwd_rec(ctx, c, 0, n, B, m, ldb, cls, tmp);
}

// The leading k1 x k1 block of the factor as a solve of its own (the sharded gradient terms: grad.hip).  Only paths that wait
// for nothing but stream order; the 2048-row leaves where k1 is a whole number of them, the 512-row leaves otherwise.
int trsm_lower_bwd_leading(fr_ctx* ctx, const fr_chol* c, int64_t k1, double* B, int64_t m, int64_t ldb, int cls)
{
    if (k1 <= 0 || m <= 0) return FR_OK;
    if (k1 > c->n) return set_err(ctx, FR_INVALID_ARGUMENT, "leading block larger than the factor");
    if (c->refine) {
        WsGuard w(ctx);
        double* tmp = w.get(sizeof(double) * (size_t)IB * (size_t)m);
        if (!tmp) return FR_OUT_OF_MEMORY;
        return trsm_bwd_rec(ctx, c, 0, k1, B, m, ldb, cls, tmp);
    }
    if (k1 >= 2 * LB && ctx->leaf512 != 0) FR_TRY(ensure_inv512(ctx, c, cls));
    WsGuard w(ctx);
    if (ctx->leaf512 != 0 && m >= 2 && k1 >= 2 * GB && k1 % GB == 0 && ensure_invbig(ctx, c, cls) == FR_OK) {
        double* tmpb = w.get(sizeof(double) * (size_t)GB * (size_t)m);
        if (!tmpb) return FR_OUT_OF_MEMORY;
        return trsm_big(ctx, c, k1, B, m, ldb, cls, false, tmpb);
    }
    (void)hipGetLastError();
    double* tmp = nullptr;
    if (ctx->leaf512 != 0 && k1 >= 2 * LB && k1 % LB == 0 && m >= 2) {
        tmp = w.get(sizeof(double) * (size_t)LB * (size_t)m);
        if (!tmp) return FR_OUT_OF_MEMORY;
    }
    return trsm_bwd_rec(ctx, c, 0, k1, B, m, ldb, cls, tmp);
}

static void chol_release(fr_chol* c)
{
    if (c->A) (void)hipFree(c->A);
    if (c->X) (void)hipFree(c->X);
    if (c->dinv) (void)hipFree(c->dinv);
    if (c->info) (void)hipFree(c->info);
    if (c->inv512) (void)hipFree(c->inv512);
    if (c->invbig) (void)hipFree(c->invbig);
    c->invbig = nullptr;
    c->invbig_cap = c->invbig_rows = 0;
    c->narrow_solves = 0;
    if (c->cest) (void)hipFree(c->cest);
    c->cest = nullptr;
    if (c->dinvt) (void)hipFree(c->dinvt);
    c->dinvt = nullptr;
    c->dinvt_cap = 0;
    c->ut_gen = 0;
    for (int i = 0; i < 4; ++i) {
        if (c->mchain[i]) (void)hipFree(c->mchain[i]);
        c->mchain[i] = nullptr;
        c->mchain_cap[i] = 0;
        c->mchain_gen[i] = 0;
    }
    if (c->yt) (void)hipFree(c->yt);
    if (c->alpha) (void)hipFree(c->alpha);
    c->yt = c->alpha = nullptr;
    c->targets_cap = 0;
    c->targets_n = -1;
    c->A = c->X = c->dinv = c->inv512 = nullptr;
    c->info = nullptr;
    c->inv512_cap = c->inv512_rows = 0;
}

// Outer block of the right-looking factorisation (= K of the trailing SYRK).  Measured in one process
// (scripts/nb_ab.py): every nb in 256..1024 is within 2 % up to N = 16384; at N = 32768 the fit takes 218 / 200 / 194 ms
// with 256 / 512 / 1024 (each pass over the trailing matrix costs a read and a write of C whatever its depth).  Sharded
// runs keep 512: block columns are dealt round-robin, and 32 panels over 8 ranks would balance poorly.
static int64_t pick_nb(const fr_ctx* ctx, int64_t n)
{
    if (ctx->nb > 0) return ctx->nb;
    // (round 5, scripts/optset_ab.py, nb = 1024 with 512 below 16384 rows against 512 throughout: N = 17408 / 18432 / 20480 / 22528
    // 35.4 / 40.8 / 53.6 / 69.7 -> 35.4 / 40.7 / 53.0 / 68.2 ms -- the threshold was 24576 before)
    return (ctx->world <= 1 && n >= 18432) ? 1024 : 512;
}

static int chol_alloc_buffers(fr_ctx* ctx, fr_chol* c, int64_t capacity, int64_t d)
{
    c->capacity = imax(capacity, 1);
    // a multiple of 128: the persistent solves (trsv.hip) read whole 128-row blocks, the rows behind the last one included
    c->ld_a = round_up(c->capacity, IB);
    c->ld_x = c->ld_a;
    c->d = d;
    const int64_t nblk = (c->capacity + IB - 1) / IB;
    c->info_cap = 3 + c->capacity;
    // (columns rounded up to a whole 128-block: the persistent solves read whole blocks of the transposed copy)
    hipError_t e = dev_malloc(ctx, (void**)&c->A, sizeof(double) * (size_t)c->ld_a * (size_t)round_up(c->capacity, IB));
    if (e == hipSuccess && d > 0) e = dev_malloc(ctx, (void**)&c->X, sizeof(double) * (size_t)c->ld_x * (size_t)d);
    if (e == hipSuccess) e = dev_malloc(ctx, (void**)&c->dinv, sizeof(double) * (size_t)nblk * INV_ELEMS);
    if (e == hipSuccess) e = dev_malloc(ctx, (void**)&c->info, sizeof(int64_t) * (size_t)c->info_cap);
    if (e == hipSuccess) e = dev_malloc(ctx, (void**)&c->cest, sizeof(double) * (size_t)nblk);
    if (e == hipSuccess) e = hipMemsetAsync(c->cest, 0, sizeof(double) * (size_t)nblk, ctx->stream);  // (estimates exist only for blocks factored / inverted here)
    if (e != hipSuccess) {
        (void)hipGetLastError();
        chol_release(c);
        return set_err(ctx, FR_OUT_OF_MEMORY, "cannot allocate a %lld x %lld factor: %s", (long long)capacity,
                       (long long)capacity, hipGetErrorString(e));
    }
    return FR_OK;
}

int chol_alloc(fr_ctx* ctx, int64_t n, int64_t capacity, int64_t d, fr_chol** out)
{
    fr_chol* c = new fr_chol();
    c->ctx = ctx;
    c->n = n;
    c->nb = pick_nb(ctx, n);
    int st = chol_alloc_buffers(ctx, c, imax(capacity, n), d);
    if (st != FR_OK) {
        delete c;
        return st;
    }
    *out = c;
    return FR_OK;
}

// with_cest: the conditioning estimates of the diagonal blocks come back behind the SAME synchronisation (round 5: they used to be a
// second read-back with a synchronisation of its own -- ~40 us of idle GPU per factorisation, 8 % of an optimizer iteration at N = 512)
int chol_fetch_info(fr_chol* c, bool with_cest)
{
    fr_ctx* ctx = c->ctx;
    int64_t head[3] = {0, 0, 0};
    FR_TRY(comm_d2h(ctx, head, c->info, sizeof(head), ctx->stream, "the sharded factorisation"));
    const int64_t nblk = (c->n + IB - 1) / IB;
    std::vector<double> hc;
    if (with_cest && nblk > 0 && c->cest) {
        hc.resize((size_t)nblk);
        FR_TRY(comm_d2h(ctx, hc.data(), c->cest, sizeof(double) * (size_t)nblk, ctx->stream, "the sharded factorisation"));
    }
    FR_TRY(comm_stream_sync(ctx, ctx->stream, "the sharded factorisation"));  // (single rank: a plain hipStreamSynchronize)
    FR_TRY(check_status_word(ctx));  // a bounded device-side wait of the factorisation (hand-offs, counted tiles) gave up
    if (with_cest) {
        c->max_cest = 0.0;
        for (double v : hc)
            if (v > c->max_cest) c->max_cest = v;  // (NaN: a failed block, reported through fail_col)
    }
    c->fail_col = head[0] - 1;
    c->n_subst = head[1];
    if (c->n_subst < 0 || c->n_subst > c->n) {
        char msg[128];
        snprintf(msg, sizeof(msg), "corrupt substitution log: %lld entries for n = %lld", (long long)head[1], (long long)c->n);
        c->n_subst = 0;
        return set_err(ctx, FR_HIP_ERROR, msg);
    }
    c->subst.resize((size_t)c->n_subst);
    if (c->n_subst > 0) {
        FR_HIP(ctx, hipMemcpyAsync(c->subst.data(), c->info + 3, sizeof(int64_t) * (size_t)c->n_subst,
                                   hipMemcpyDeviceToHost, ctx->stream));
        FR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return FR_OK;
}

// Multi-GPU: every rank logged only the pivots of the panels it owned and holds the conditioning estimates of the diagonal
// blocks it factored; ONE all-gather carries both (the estimates ride behind the log as bit patterns), the host merges:
// the ranks end with the same substitution list, failure column and largest estimate -- and therefore take the same
// refinement decision (assemble_and_factor).
static int merge_info(fr_chol* c)
{
    fr_ctx* ctx = c->ctx;
    const int W = ctx->world;
    const int64_t nblk = (c->n + IB - 1) / IB;
    const int64_t len = 3 + c->n + nblk;
    WsGuard g(ctx), sg(ctx);
    int64_t* all = (int64_t*)g.get(sizeof(int64_t) * (size_t)(len * W));
    int64_t* mine = (int64_t*)sg.get(sizeof(int64_t) * (size_t)len);
    if (!all || !mine) return FR_OUT_OF_MEMORY;
    FR_HIP(ctx, hipMemcpyAsync(mine, c->info, sizeof(int64_t) * (size_t)(3 + c->n), hipMemcpyDeviceToDevice, ctx->stream));
    if (nblk > 0) {
        if (c->cest)
            FR_HIP(ctx, hipMemcpyAsync(mine + 3 + c->n, c->cest, sizeof(double) * (size_t)nblk, hipMemcpyDeviceToDevice, ctx->stream));
        else
            FR_HIP(ctx, hipMemsetAsync(mine + 3 + c->n, 0, sizeof(double) * (size_t)nblk, ctx->stream));
    }
    FR_TRY(comm_allgather_i64(ctx, mine, all, (size_t)len));
    std::vector<int64_t> host((size_t)(len * W));
    FR_TRY(comm_d2h(ctx, host.data(), all, sizeof(int64_t) * host.size(), ctx->stream, "the merge of the substitution logs"));
    FR_TRY(comm_stream_sync(ctx, ctx->stream, "the merge of the substitution logs"));
    FR_TRY(check_status_word(ctx));  // a bounded device-side wait of the factorisation (hand-offs, counted tiles) gave up
    int64_t fail = -1;
    std::vector<int64_t> subst;
    double max_cest = 0.0;
    for (int r = 0; r < W; ++r) {
        const int64_t* h = host.data() + (size_t)r * len;
        if (h[0] > 0 && (fail < 0 || h[0] - 1 < fail)) fail = h[0] - 1;
        for (int64_t i = 0; i < h[1] && i < c->n; ++i) subst.push_back(h[3 + i]);
        for (int64_t b = 0; b < nblk; ++b) {
            double v;
            memcpy(&v, h + 3 + c->n + b, sizeof(double));
            if (v > max_cest) max_cest = v;  // (NaN: a failed block, reported through fail_col)
        }
    }
    std::sort(subst.begin(), subst.end());
    c->fail_col = fail;
    c->n_subst = (int64_t)subst.size();
    c->subst = subst;
    c->max_cest = max_cest;
    return FR_OK;
}

int potrf_device(fr_ctx* ctx, fr_chol* c, int64_t j0, int64_t n, int mode, double sub)
{
    c->inv512_rows = 0;
    c->invbig_rows = 0;
    c->narrow_solves = 0;
    ++c->gen;
    return potrf_blocked(ctx, c->A + j0 + j0 * c->ld_a, c->ld_a, n, j0, mode, sub, c->dinv + (j0 / IB) * INV_ELEMS, c->info,
                         c->nb);
}

// Factor an arbitrary device matrix in place with scratch inverse blocks / info (sample_at's m x m factor).
int potrf_matrix_ws(fr_ctx* ctx, double* A, int64_t ld, int64_t n, int mode, double sub, int64_t* fail_col)
{
    *fail_col = -1;
    if (n <= 0) return FR_OK;
    WsGuard dg(ctx), ig(ctx);
    const int64_t nblk = (n + IB - 1) / IB;
    double* dinv_tmp = dg.get(sizeof(double) * (size_t)nblk * INV_ELEMS);
    int64_t* info = (int64_t*)ig.get(sizeof(int64_t) * (size_t)(3 + n));
    if (!dinv_tmp || !info) return FR_OUT_OF_MEMORY;
    FR_HIP(ctx, hipMemsetAsync(info, 0, sizeof(int64_t) * 3, ctx->stream));
    FR_TRY(potrf_blocked(ctx, A, ld, n, 0, mode, sub, dinv_tmp, info, pick_nb(ctx, n)));
    int64_t head = 0;
    FR_HIP(ctx, hipMemcpyAsync(&head, info, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    FR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *fail_col = head - 1;
    return FR_OK;
}

// Gram (lower + noise^2) from the resident inputs, then the factorisation; fills the host info mirror.
static int assemble_and_factor_once(fr_chol* c, const fr_kprog* kernel, double noise, int has_eps, double eps);

// The largest conditioning estimate of the diagonal blocks of the factorisation that just ran (stream synchronised).
static int fetch_max_cest(fr_chol* c)
{
    fr_ctx* ctx = c->ctx;
    const int64_t nblk = (c->n + IB - 1) / IB;
    c->max_cest = 0.0;
    if (nblk <= 0 || !c->cest) return FR_OK;
    std::vector<double> h((size_t)nblk);
    FR_TRY(comm_d2h(ctx, h.data(), c->cest, sizeof(double) * (size_t)nblk, ctx->stream, "the read-back of the conditioning estimates"));
    FR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (double v : h)
        if (v > c->max_cest) c->max_cest = v;  // (NaN: a failed block, reported through fail_col)
    return FR_OK;
}

// Gram + factorisation with the refinement policy of fr_ctx::refine: automatic = factor with the plain explicit-inverse
// products; if a diagonal block turns out ill-conditioned, factor once more with a refinement step behind every such
// product, and keep refining on this handle until a refined factorisation sees well-conditioned blocks again.
static int assemble_and_factor(fr_chol* c, const fr_kprog* kernel, double noise, int has_eps, double eps)
{
    fr_ctx* ctx = c->ctx;
    // Sharded: the same policy.  The largest estimate is the maximum over every rank's blocks (merge_info), c->refine only
    // ever changes as a function of it and of the (rank-uniform) options, so every rank decides alike; the repeat with
    // refinement runs on the whole-panel schedule (potrf_blocked), whose owner solves its panel with the refined products.
    if (ctx->refine == 0) c->refine = false;
    if (ctx->refine == 1) c->refine = true;
    ctx->solve_timeout_seen = false;
    int st = assemble_and_factor_once(c, kernel, noise, has_eps, eps);
    if (st == FR_HIP_ERROR && ctx->solve_timeout_seen && ctx->panel_chain && ctx->panel_chain_launches > 0 && ctx->world == 1) {
        // a hand-off of the resident panel chain timed out (its workgroups did not get CUs side by side: another context's resident
        // kernels on this GPU, a tool that serialises workgroups): once more on the chain of launches, which waits for nothing but
        // stream order.  Single rank only -- a sharded factorisation is a collective, a rank must not repeat it alone.
        ctx->solve_timeout_seen = false;
        ++ctx->panel_chain_fallbacks;
        const int64_t saved = ctx->panel_chain;
        ctx->panel_chain = 0;
        st = assemble_and_factor_once(c, kernel, noise, has_eps, eps);
        ctx->panel_chain = saved;
    }
    if (ctx->refine != -1 || (st != FR_OK && st != FR_NOT_POSITIVE_DEFINITE)) return st;
    const bool ill = c->max_cest > ctx->refine_threshold;
    if (!c->refine && ill) {
        c->refine = true;
        st = assemble_and_factor_once(c, kernel, noise, has_eps, eps);
    } else if (c->refine && c->max_cest * 4.0 < ctx->refine_threshold) {
        c->refine = false;  // the next factorisation takes the plain products again
    }
    return st;
}

static int assemble_and_factor_once(fr_chol* c, const fr_kprog* kernel, double noise, int has_eps, double eps)
{
    fr_ctx* ctx = c->ctx;
    struct Scope {  // per-operation state of the context (entry points hold its lock)
        fr_ctx* ctx;
        ~Scope()
        {
            ctx->refine_now = false;
            ctx->cur_cest = nullptr;
        }
    } scope{ctx};
    ctx->refine_now = c->refine;
    ctx->cur_cest = c->cest;
    c->inv512_rows = 0;
    c->invbig_rows = 0;
    c->narrow_solves = 0;
    ++c->gen;
    drain_stale_status(ctx);  // (a time-out left behind by an earlier call must not be read as this factorisation's)
    FR_HIP(ctx, hipMemsetAsync(c->info, 0, sizeof(int64_t) * 3, ctx->stream));
    FR_HIP(ctx, hipMemsetAsync(c->cest, 0, sizeof(double) * (size_t)((c->capacity + IB - 1) / IB), ctx->stream));
    FR_TRY(launch_gram_sym(ctx, *kernel, c->X, c->n, c->ld_x, c->d, noise * noise, c->A, c->ld_a, ctx->world, ctx->rank, c->nb));
    FR_TRY(potrf_blocked(ctx, c->A, c->ld_a, c->n, 0, has_eps ? 1 : 0, eps, c->dinv, c->info, c->nb, true));
    FR_TRY(chol_fetch_info(c, ctx->world <= 1));
    if (ctx->world > 1) FR_TRY(merge_info(c));  // (also the largest conditioning estimate over every rank's blocks)
    c->collective = ctx->world > 1;  // every rank holds this handle: fr_grad_terms / fr_chol_add_rows on it may be collective
    if (c->fail_col >= 0)
        return set_err(ctx, FR_NOT_POSITIVE_DEFINITE,
                       has_eps ? "Cholesky decomposition failed even though we used `cholesky_epsilon` value of %g (column %lld)"
                               : "Cholesky decomposition failed, consider setting `cholesky_epsilon` via "
                                 "`GaussianProcessBuilder` (eps %g unused; column %lld)",
                       eps, (long long)c->fail_col);
    return FR_OK;
}

static int upload_rows(fr_ctx* ctx, const double* src, int64_t ldsrc, double* dst, int64_t lddst, int64_t rows,
                       int64_t cols)
{
    if (rows <= 0 || cols <= 0) return FR_OK;
    const bool dev = is_device_ptr(src);
    if (!dev) {
        // host data: packed into the pinned bounce buffer by the CPU, then one DMA the host does not wait for (a pageable
        // source would make the copy synchronous and cost a round trip per call: fr_chol_add_rows uploads 512 x d doubles)
        double* pin = (double*)pinned_get(ctx, sizeof(double) * (size_t)rows * (size_t)cols);
        if (pin) {
            FR_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the previous user of the bounce buffer (a no-op on an idle stream)
            for (int64_t j = 0; j < cols; ++j) memcpy(pin + j * rows, src + j * ldsrc, sizeof(double) * (size_t)rows);
            FR_HIP(ctx, hipMemcpy2DAsync(dst, sizeof(double) * lddst, pin, sizeof(double) * rows, sizeof(double) * rows, cols,
                                         hipMemcpyHostToDevice, ctx->stream));
            return FR_OK;
        }
    }
    FR_HIP(ctx, hipMemcpy2DAsync(dst, sizeof(double) * lddst, src, sizeof(double) * ldsrc, sizeof(double) * rows, cols,
                                 dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
    if (!dev) FR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FR_OK;
}

// EMatrix::add_rows growth policy (extendable_matrix.rs:33-43): max(required, 3*capacity/2)
static int chol_grow(fr_chol* c, int64_t required)
{
    fr_ctx* ctx = c->ctx;
    if (required <= c->capacity) return FR_OK;
    const int64_t new_cap = imax(required, (3 * c->capacity) / 2);
    fr_chol nc;
    nc.ctx = ctx;
    FR_TRY(chol_alloc_buffers(ctx, &nc, new_cap, c->d));
    int st = FR_OK;
    hipError_t e = hipSuccess;
    if (c->n > 0) {
        e = hipMemcpy2DAsync(nc.A, sizeof(double) * nc.ld_a, c->A, sizeof(double) * c->ld_a, sizeof(double) * c->n, c->n,
                             hipMemcpyDeviceToDevice, ctx->stream);
        if (e == hipSuccess && c->d > 0)
            e = hipMemcpy2DAsync(nc.X, sizeof(double) * nc.ld_x, c->X, sizeof(double) * c->ld_x, sizeof(double) * c->n,
                                 c->d, hipMemcpyDeviceToDevice, ctx->stream);
        const int64_t nblk = (c->n + IB - 1) / IB;
        if (e == hipSuccess)
            e = hipMemcpyAsync(nc.dinv, c->dinv, sizeof(double) * (size_t)nblk * INV_ELEMS, hipMemcpyDeviceToDevice,
                               ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(nc.cest, c->cest, sizeof(double) * (size_t)nblk, hipMemcpyDeviceToDevice, ctx->stream);
    }
    if (e == hipSuccess)
        e = hipMemcpyAsync(nc.info, c->info, sizeof(int64_t) * (size_t)imin(c->info_cap, nc.info_cap),
                           hipMemcpyDeviceToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        st = set_err(ctx, FR_HIP_ERROR, "growing the factor failed: %s", hipGetErrorString(e));
        chol_release(&nc);
        return st;
    }
    chol_release(c);
    c->A = nc.A;
    c->X = nc.X;
    c->dinv = nc.dinv;
    c->info = nc.info;
    c->cest = nc.cest;
    nc.cest = nullptr;
    c->capacity = nc.capacity;
    c->ld_a = nc.ld_a;
    c->ld_x = nc.ld_x;
    c->info_cap = nc.info_cap;
    nc.A = nc.X = nc.dinv = nullptr;
    nc.info = nullptr;
    return FR_OK;
}

}  // namespace fr

using namespace fr;

extern "C" {

static int check_zero_diag(fr_chol* c, const char* what);

int fr_chol_from_inputs(fr_ctx* ctx, const fr_kprog* kernel, const double* X, int64_t n, int64_t ldx, int64_t d,
                        double noise, int has_eps, double eps, int64_t capacity_hint, fr_chol** out)
{
    if (!ctx || !out) return FR_INVALID_ARGUMENT;
    FR_LOCK(ctx);
    *out = nullptr;
    FR_HIP(ctx, hipSetDevice(ctx->device));
    FR_TRY(kprog_check(ctx, kernel));
    if (n < 0 || d < 0 || ldx < imax(n, 1)) return set_err(ctx, FR_SHAPE, "bad training input shape");
    if (n > 0 && d > 0 && !X) return set_err(ctx, FR_INVALID_ARGUMENT, "null training inputs");
    fr_chol* c = nullptr;
    FR_TRY(chol_alloc(ctx, n, imax(capacity_hint, n), d, &c));
    int st = upload_rows(ctx, X, ldx, c->X, c->ld_x, n, d);
    if (st == FR_OK) st = assemble_and_factor(c, kernel, noise, has_eps, eps);
    if (st != FR_OK && st != FR_NOT_POSITIVE_DEFINITE) {
        fr_chol_free(c);
        return st;
    }
    *out = c;
    return st;
}

int fr_chol_refactor(fr_chol* c, const fr_kprog* kernel, double noise, int has_eps, double eps)
{
    if (!c) return FR_INVALID_ARGUMENT;
    FR_LOCK(c->ctx);
    fr_ctx* ctx = c->ctx;
    FR_HIP(ctx, hipSetDevice(ctx->device));
    FR_TRY(kprog_check(ctx, kernel));
    if (c->d == 0 && c->n > 0 && !c->X) return set_err(ctx, FR_INVALID_ARGUMENT, "factor holds no training inputs");
    c->nb = pick_nb(ctx, c->n);
    return assemble_and_factor(c, kernel, noise, has_eps, eps);
}

int fr_chol_from_matrix(fr_ctx* ctx, const double* A, int64_t n, int64_t lda, int has_eps, double eps, fr_chol** out)
{
    if (!ctx || !out) return FR_INVALID_ARGUMENT;
    FR_LOCK(ctx);
    *out = nullptr;
    FR_HIP(ctx, hipSetDevice(ctx->device));
    if (n < 0 || lda < imax(n, 1)) return set_err(ctx, FR_SHAPE, "bad matrix shape");
    fr_chol* c = nullptr;
    FR_TRY(chol_alloc(ctx, n, n, 0, &c));
    // refinement policy as in assemble_and_factor: factor; if a diagonal block turns out ill-conditioned (automatic mode),
    // upload the matrix again and factor once more with the refinement step behind every product with an inverse block
    c->refine = ctx->refine == 1;
    int st = FR_OK;
    const int64_t chain_saved = ctx->panel_chain;
    struct ChainRestore {
        fr_ctx* ctx;
        int64_t v;
        ~ChainRestore() { ctx->panel_chain = v; }
    } chain_restore{ctx, chain_saved};
    for (int attempt = 0; attempt < 3; ++attempt) {
        ctx->solve_timeout_seen = false;
        st = upload_rows(ctx, A, lda, c->A, c->ld_a, n, n);
        if (st == FR_OK && (hipMemsetAsync(c->info, 0, sizeof(int64_t) * 3, ctx->stream) != hipSuccess ||
                            hipMemsetAsync(c->cest, 0, sizeof(double) * (size_t)((c->capacity + IB - 1) / IB), ctx->stream) != hipSuccess))
            st = set_err(ctx, FR_HIP_ERROR, "memset failed");
        if (st == FR_OK) {
            ctx->refine_now = c->refine;
            ctx->cur_cest = c->cest;
            drain_stale_status(ctx);
            st = potrf_device(ctx, c, 0, n, has_eps ? 1 : 0, eps);
            ctx->refine_now = false;
            ctx->cur_cest = nullptr;
        }
        if (st == FR_OK) st = chol_fetch_info(c, true);
        if (st == FR_HIP_ERROR && ctx->solve_timeout_seen && ctx->panel_chain && ctx->world == 1) {
            ++ctx->panel_chain_fallbacks;  // (a timed-out hand-off of the resident panel chain: once more on the launch chain)
            ctx->panel_chain = 0;
            continue;
        }
        if (st == FR_OK && ctx->refine == -1 && !c->refine && c->max_cest > ctx->refine_threshold) {
            c->refine = true;
            continue;
        }
        break;
    }
    if (st == FR_OK && c->fail_col >= 0)
        st = set_err(ctx, FR_NOT_POSITIVE_DEFINITE, "Cholesky decomposition failed at column %lld", (long long)c->fail_col);
    if (st != FR_OK && st != FR_NOT_POSITIVE_DEFINITE) {
        fr_chol_free(c);
        return st;
    }
    *out = c;
    return st;
}

int fr_chol_add_rows(fr_chol* c, const fr_kprog* kernel, const double* Xall, int64_t n_all, int64_t ldx, int64_t d,
                     int64_t nb_new, double noise)
{
    if (!c) return FR_INVALID_ARGUMENT;
    FR_LOCK(c->ctx);
    fr_ctx* ctx = c->ctx;
    FR_HIP(ctx, hipSetDevice(ctx->device));
    FR_TRY(kprog_check(ctx, kernel));
    const int64_t n_old = n_all - nb_new;  // algebra/mod.rs:102
    if (nb_new < 0 || n_old != c->n) return set_err(ctx, FR_SHAPE, "add_rows: factor holds %lld rows, inputs imply %lld",
                                                     (long long)c->n, (long long)n_old);
    if (d != c->d) return set_err(ctx, FR_SHAPE, "add_rows: feature count %lld != %lld", (long long)d, (long long)c->d);
    if (ldx < imax(n_all, 1)) return set_err(ctx, FR_SHAPE, "add_rows: bad leading dimension");
    if (nb_new == 0) return FR_OK;
    if (c->fail_col >= 0)
        return set_err(ctx, FR_NOT_POSITIVE_DEFINITE, "add_rows: the factor is not valid (its factorisation failed at column %lld)",
                       (long long)c->fail_col);
    // Cholesky::insert_column solves L11 r = col with the CHECKED solve and asserts on a zero diagonal
    // ("Unable to solve lower triangular system!", nalgebra cholesky.rs); the explicit-inverse solve below would write
    // inf / NaN silently instead
    if (n_old > 0) FR_TRY(check_zero_diag(c, "Cholesky::insert_column: Unable to solve lower triangular system!"));
    // (the cached 512-block inverses stay valid: rows below n_old are appended, blocks inside the old factor do not change)
    FR_TRY(chol_grow(c, n_all));
    ++c->gen;  // cached alpha is stale; the targets cover n_old rows only and have to be handed over again
    c->nb = pick_nb(ctx, n_all);
    // new rows of the EMatrix mirror
    FR_TRY(upload_rows(ctx, Xall + n_old, ldx, c->X + n_old, c->ld_x, nb_new, d));
    // The append itself is restartable (everything it writes is recomputed from X and the old factor), which serves two
    // purposes: a persistent L21 solve that gave up on a hand-off is repeated on the recursive path (solve_retry), and an
    // append whose new diagonal blocks turn out ill-conditioned is repeated with iterative refinement (the policy of
    // assemble_and_factor; the estimates come from the re-aligned inverse blocks).
    bool readback_ok = false;
    int64_t ext_inv512_before = -1, ext_invbig_before = -1;  // what the caches covered before an append extended them (-1: it did not)
    if (!ctx->readback && hipHostMalloc(&ctx->readback, 4096, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        ctx->readback = nullptr;  // (falls back to separate read-backs)
    }
    auto append = [&]() -> int {
        struct Scope {
            fr_ctx* ctx;
            ~Scope() { ctx->refine_now = false; }
        } scope{ctx};
        ctx->refine_now = c->refine;
        c->n = n_old;
        if (ext_inv512_before >= 0) {  // a repeat: what the first attempt added to the caches described blocks that are rewritten now
            c->inv512_rows = ext_inv512_before < c->inv512_rows ? ext_inv512_before : c->inv512_rows;
            c->invbig_rows = ext_invbig_before < c->invbig_rows ? ext_invbig_before : c->invbig_rows;
            ext_inv512_before = ext_invbig_before = -1;
        }
        const int64_t ld = c->ld_a;
        double* A21 = c->A + n_old;               // nb_new x n_old
        double* A22 = c->A + n_old + n_old * ld;  // nb_new x nb_new
        // K21 = k(new, old), K22 = lower(k(new, new)) + noise^2 I       (algebra/mod.rs:115-121)
        FR_TRY(launch_gram_sym(ctx, *kernel, c->X + n_old, nb_new, c->ld_x, d, noise * noise, A22, ld));
        // L21 = K21 L11^-T ; K22 -= L21 L21^T ; L22 = chol(K22) with insert_column's plain sqrt (mode 2)
        if (n_old >= 4 * IB) {
            // a short, wide block: solving from the right would be GEMMs of nb_new rows (four tile rows, ~120 launches; with ONE new
            // row -- the Bayesian-optimisation loop of readme.md:7 -- a recursion of ~3 n / 128 one-row products: 5.0 / 19.4 ms at
            // N = 8192 / 32768, measured in round 4); the transposed problem  L21^T = L11^-1 K12  is a forward solve with nb_new
            // right-hand sides -- whatever trsm_lower_fwd picks for that count: the single-column kernel, K9, the 2048-row
            // leaves -- then one transposition into place
            //
            // Sharded: the nb_new right-hand sides -- the new rows -- are dealt to the ranks in equal
            // slices; a rank assembles and solves its slice only (n_old^2 nb_new / W flop instead of all of them), ONE
            // all-gather returns L21^T to everybody, and the rest -- transposition, the nb_new x nb_new Schur complement and its
            // factorisation: 1 / (n_old / nb_new) of the work -- is repeated on every rank with the same deterministic kernels
            // on the same data, so every rank ends with the bit-identical grown factor and no broadcast is needed.  The slice
            // solves keep to the stream-ordered paths (no persistent kernel: a rank repeating ITS solve after a timed-out
            // hand-off would issue the all-gather twice).
            WsGuard wg(ctx);
            const int64_t ldw = round_up(n_old, kAlign);
            const int Wn = ctx->world, me = ctx->rank;
            const int64_t slice = Wn > 1 ? (nb_new + Wn - 1) / Wn : nb_new;
            double* W = wg.get(sizeof(double) * (size_t)ldw * (size_t)(slice * Wn));
            if (Wn > 1) {
                bool all_ok = true;
                FR_TRY(comm_agree(ctx, W != nullptr, &all_ok));
                if (W && !all_ok) return set_err(ctx, FR_OUT_OF_MEMORY, "a peer rank could not allocate its add_rows workspace: append abandoned on every rank");
            }
            if (!W) return FR_OUT_OF_MEMORY;
            if (Wn > 1) {
                const int64_t lo = imin(nb_new, (int64_t)me * slice), cols = imin(nb_new, lo + slice) - lo;
                double* Ws = W + lo * ldw;
                struct NoPersistent {
                    fr_ctx* ctx;
                    int64_t saved;
                    ~NoPersistent() { ctx->trsv = saved; }
                } np{ctx, ctx->trsv};
                ctx->trsv = 0;
                if (cols > 0) {
                    FR_TRY(launch_gram_cross(ctx, *kernel, c->X, n_old, c->ld_x, c->X + n_old + lo, cols, c->ld_x, d, Ws, ldw));
                    FR_TRY(trsm_lower_fwd(ctx, c, n_old, Ws, cols, ldw, FR_PROF_GEMM_PANEL));
                }
                FR_TRY(comm_allgather(ctx, W + (int64_t)me * slice * ldw, W, (size_t)(slice * ldw)));
            } else {
                FR_TRY(launch_gram_cross(ctx, *kernel, c->X, n_old, c->ld_x, c->X + n_old, nb_new, c->ld_x, d, W, ldw));
                FR_TRY(trsm_lower_fwd(ctx, c, n_old, W, nb_new, ldw, FR_PROF_GEMM_PANEL));
            }
            FR_TRY(launch_transpose(ctx, W, n_old, nb_new, ldw, A21, ld));
            FR_TRY(gemm(ctx, FR_PROF_SYRK, nb_new, nb_new, n_old, A21, ld, false, A21, ld, false, -1.0, 1.0, A22, ld, true));
        } else {
            FR_TRY(launch_gram_cross(ctx, *kernel, c->X + n_old, nb_new, c->ld_x, c->X, n_old, c->ld_x, d, A21, ld));
            if (n_old > 0) {
                FR_TRY(trsm_right_rec(ctx, c->A, ld, c->dinv, n_old, A21, nb_new, ld, FR_PROF_GEMM_PANEL));
                FR_TRY(gemm(ctx, FR_PROF_SYRK, nb_new, nb_new, n_old, A21, ld, false, A21, ld, false, -1.0, 1.0, A22, ld, true));
            }
        }
        const bool aligned = n_old % IB == 0;  // the new rows start on the 128-grid of the inverse blocks
        {
            WsGuard tmp(ctx);
            const int64_t nblk = (nb_new + IB - 1) / IB;
            // aligned (the usual case: appends in multiples of 128 rows): the factorisation's diagonal-block kernels write the
            // inverse blocks and their conditioning estimates straight into place -- the four extra launches per 512 rows that
            // rebuilt them were 0.13 ms of every add_samples(512) (configs[4])
            double* dinv_new = aligned ? c->dinv + (n_old / IB) * INV_ELEMS : tmp.get(sizeof(double) * (size_t)nblk * INV_ELEMS);
            if (!dinv_new) return FR_OUT_OF_MEMORY;
            double* saved_cest = ctx->cur_cest;
            ctx->cur_cest = aligned ? c->cest : nullptr;  // (K4 indexes it by global column / 128: potrf_blocked gets col0 = n_old)
            const int st2 = potrf_blocked(ctx, A22, ld, nb_new, n_old, 2, 0.0, dinv_new, c->info, c->nb);
            ctx->cur_cest = saved_cest;
            FR_TRY(st2);
        }
        c->n = n_all;
        // otherwise re-align the inverse blocks with the global 128-grid over the rows that changed (and take their estimates)
        if (!aligned)
            for (int64_t b = n_old / IB; b * IB < n_all; ++b) {
                const int64_t j = b * IB, sb = imin(IB, n_all - j);
                FR_TRY(invert_block128(ctx, c->A + j + j * ld, ld, sb, c->dinv + b * INV_ELEMS, c->cest + b));
            }
        // ONE synchronisation per append: the status of a persistent solve, the conditioning estimates of the blocks that
        // changed and the zero-diagonal flag of the new rows (what the next append's checked solve asks first) come back
        // together.  (Round 3: a synchronisation each for the append, the estimates, the next call's diagonal check and its
        // upload of the new inputs -- ~250 us of idle GPU per add_samples(512), measured on a kernel trace.)
        const int64_t b_lo = n_old / IB, b_hi = (n_all + IB - 1) / IB;
        readback_ok = ctx->readback && (size_t)(b_hi - b_lo + 1) * sizeof(double) <= 4096;
        if (readback_ok) FR_TRY(launch_append_status(ctx, c->A + n_old + n_old * ld, nb_new, ld, c->cest + b_lo, b_hi - b_lo, (double*)ctx->readback));
        // The inverse-block caches, when they are in use (the append's own L21 solve took them), are extended over the new rows
        // IN FRONT of the synchronisation: the few small launches queue behind the Schur block's factorisation while the host
        // still has its hands free, instead of one by one behind the synchronisation with the GPU idle between them (66 us of
        // gaps per add_samples(512) on round 5's kernel trace).  They only read what the append has written; should the append
        // be repeated (a timed-out hand-off, an ill-conditioned block) the caches are cut back to the old rows first.
        if (!c->refine && c->inv512_rows > 0) {
            const int64_t inv512_before = c->inv512_rows, invbig_before = c->invbig_rows;
            const int st2 = ensure_inv512(ctx, c, FR_PROF_GEMM_PANEL);
            if (st2 == FR_OK && invbig_before > 0) (void)ensure_invbig(ctx, c, FR_PROF_GEMM_PANEL);
            ext_inv512_before = inv512_before;
            ext_invbig_before = invbig_before;
            (void)hipGetLastError();
        }
        FR_TRY(comm_stream_sync(ctx, ctx->stream, "add_rows"));  // (sharded: the stream holds an all-gather -- bounded wait)
        return check_status_word(ctx);
    };
    // a failed append leaves the old factor: so do its caches.  Whatever the abandoned attempt queued into the inverse-block caches
    // described rows that are not part of the factor -- a later append of OTHER rows at the same n_old would find
    // inv512_rows >= its need and skip the rebuild (advisor finding, round 5): every failing exit cuts them back
    auto cut_caches_back = [&]() {
        if (ext_inv512_before >= 0) {
            c->inv512_rows = ext_inv512_before < c->inv512_rows ? ext_inv512_before : c->inv512_rows;
            c->invbig_rows = ext_invbig_before < c->invbig_rows ? ext_invbig_before : c->invbig_rows;
            ext_inv512_before = ext_invbig_before = -1;
        }
    };
    // sharded: the body holds collectives, so a rank must not repeat it on its own (solve_retry: collective = true)
    int st = solve_retry(ctx, append, ctx->world > 1);
    if (st != FR_OK) {
        c->n = n_old;
        cut_caches_back();
        return st;
    }
    if (readback_ok) {
        // (the old rows' diagonal was checked before the append: the flag of the grown factor is that of its new rows)
        c->diag_zero = *(const int64_t*)ctx->readback != 0;
        c->diag_gen = c->gen;
    }
    if (ctx->refine == -1) {
        if (readback_ok) {
            const double* h = (const double*)ctx->readback + 1;
            for (int64_t b = 0; b < (n_all + IB - 1) / IB - n_old / IB; ++b)
                if (h[b] > c->max_cest) c->max_cest = h[b];  // (the estimates of the old blocks are in max_cest already)
        } else {
            FR_TRY(fetch_max_cest(c));
        }
        if (ctx->world > 1) {
            // every rank factored the same Schur complement and should hold the same estimates; the decision to repeat the
            // append is a collective one all the same (a repeat issues the all-gather again), so it is taken on the maximum
            // over the ranks, not on trust
            if (!ctx->agree_buf) FR_HIP(ctx, hipMalloc((void**)&ctx->agree_buf, sizeof(int64_t) * 65));
            double* ab = (double*)ctx->agree_buf;
            FR_HIP(ctx, hipMemcpyAsync(ab, &c->max_cest, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            FR_TRY(comm_allgather(ctx, ab, ab + 1, 1));
            if (ctx->world > 64) return set_err(ctx, FR_INVALID_ARGUMENT, "more than 64 ranks");
            double hm[64];
            FR_TRY(comm_d2h(ctx, hm, ab + 1, sizeof(double) * (size_t)ctx->world, ctx->stream, "add_rows: agreement on the conditioning estimate"));
            FR_TRY(comm_stream_sync(ctx, ctx->stream, "add_rows: agreement on the conditioning estimate"));
            for (int r = 0; r < ctx->world; ++r)
                if (hm[r] > c->max_cest) c->max_cest = hm[r];
        }
        if (!c->refine && c->max_cest > ctx->refine_threshold) {
            c->refine = true;  // an ill-conditioned appended block: once more with the refinement step behind every inverse product
            st = solve_retry(ctx, append, ctx->world > 1);
            if (st != FR_OK) {
                c->n = n_old;
                cut_caches_back();
            }
        }
    }
    return st;
}

int fr_chol_info(const fr_chol* c, int64_t* n, int64_t* capacity, int64_t* d, int64_t* n_subst, int64_t* fail_col)
{
    if (!c) return FR_INVALID_ARGUMENT;
    FR_LOCK(c->ctx);  // (a concurrent refactor / add_rows on the same handle is legal: read a consistent state)
    if (n) *n = c->n;
    if (capacity) *capacity = c->capacity;
    if (d) *d = c->d;
    if (n_subst) *n_subst = c->n_subst;
    if (fail_col) *fail_col = c->fail_col;
    return FR_OK;
}

int fr_chol_conditioning(const fr_chol* c, double* max_estimate, int* refined)
{
    if (!c) return FR_INVALID_ARGUMENT;
    FR_LOCK(c->ctx);
    if (max_estimate) *max_estimate = c->max_cest;
    if (refined) *refined = c->refine ? 1 : 0;
    return FR_OK;
}

int fr_chol_substitutions(const fr_chol* c, int64_t* idx, int64_t max_idx)
{
    if (!c || (!idx && max_idx > 0)) return FR_INVALID_ARGUMENT;
    FR_LOCK(c->ctx);
    for (int64_t i = 0; i < c->n_subst && i < max_idx; ++i) idx[i] = c->subst[(size_t)i];
    return FR_OK;
}

// In-place solves on a DEVICE operand: a persistent solve that gave up leaves B partly overwritten, so the operand is saved
// first whenever a persistent kernel may run (n x m doubles next to the 4 n^2 bytes of factor the solve streams) and put
// back before the repeat on the recursive path.
static int solve_in_place(fr_chol* c, double* B, int64_t m, int64_t ldb, bool both)
{
    fr_ctx* ctx = c->ctx;
    FR_HIP(ctx, hipSetDevice(ctx->device));
    // The path is decided HERE, once: when the 2048-row leaves are predicted their inverse caches are built now (n / 2048 x 32 MiB per
    // factor -- 512 MiB at N = 32768; option bigleaf_max = 0 switches the path off); if they cannot be, trsm_lower_fwd / bwd will fall
    // through to a persistent kernel and the backup is taken (round-4 advisor finding: the fall-through used to run without one).
    bool big = m > 1 && c->n > 0 && use_big_leaves(ctx, c, c->n, m, false) && !(ctx->narrow_batched_max > 0 && m <= ctx->narrow_batched_max);
    if (big && !(ensure_inv512(ctx, c, FR_PROF_GEMM_SOLVE) == FR_OK && ensure_invbig(ctx, c, FR_PROF_GEMM_SOLVE) == FR_OK)) {
        (void)hipGetLastError();
        big = false;
    }
    const bool persistent = ctx->trsv && !c->refine && c->n > 0 && m > 0 && (m == 1 || use_column_groups(ctx, c, c->n, m)) && !big;
    WsGuard bk(ctx);
    double* backup = nullptr;
    if (persistent && is_device_ptr(B) && ldb >= c->n) {
        backup = bk.get(sizeof(double) * (size_t)c->n * (size_t)m);
        if (!backup) return FR_OUT_OF_MEMORY;
        FR_TRY(launch_copy(ctx, B, ldb, backup, c->n, c->n, m));
    }
    bool first = true;
    return solve_retry(ctx, [&]() -> int {
        if (!first && backup) FR_TRY(launch_copy(ctx, backup, c->n, B, ldb, c->n, m));
        first = false;
        Staged b(ctx);
        FR_TRY(b.inout(B, c->n, m, ldb));
        FR_TRY(trsm_lower_fwd(ctx, c, c->n, b.dev, m, b.ld, FR_PROF_GEMM_SOLVE));
        if (both) FR_TRY(trsm_lower_bwd(ctx, c, c->n, b.dev, m, b.ld, FR_PROF_GEMM_SOLVE));
        return b.commit();  // (device operand: nothing to copy back, but commit reads the status of a persistent kernel)
    });
}

int fr_chol_solve(fr_chol* c, double* B, int64_t m, int64_t ldb)
{
    if (!c) return FR_INVALID_ARGUMENT;
    FR_LOCK(c->ctx);
    return solve_in_place(c, B, m, ldb, true);
}

static int check_zero_diag(fr_chol* c, const char* what)
{
    fr_ctx* ctx = c->ctx;
    if (c->diag_gen != c->gen) {  // once per factor (fr_internal.hpp: diag_gen)
        FR_HIP(ctx, hipMemsetAsync(c->info + 2, 0, sizeof(int64_t), ctx->stream));
        FR_TRY(launch_diag_check_zero(ctx, c->A, c->n, c->ld_a, c->info + 2));
        int64_t flag = 0;
        FR_TRY(comm_d2h(ctx, &flag, c->info + 2, sizeof(int64_t), ctx->stream, "the zero-diagonal check"));
        FR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        c->diag_zero = flag != 0;
        c->diag_gen = c->gen;
    }
    if (c->diag_zero) return set_err(ctx, FR_SINGULAR_SOLVE, "%s : solve failed", what);
    return FR_OK;
}

int fr_chol_solve_lower(fr_chol* c, double* B, int64_t m, int64_t ldb)
{
    if (!c) return FR_INVALID_ARGUMENT;
    FR_LOCK(c->ctx);
    FR_HIP(c->ctx, hipSetDevice(c->ctx->device));
    FR_TRY(check_zero_diag(c, "solve_lower_triangular"));
    return solve_in_place(c, B, m, ldb, false);
}

static int chol_inverse_impl(fr_chol* c, double* out, int64_t ldo);

int fr_chol_inverse(fr_chol* c, double* out, int64_t ldo)
{
    if (!c) return FR_INVALID_ARGUMENT;
    FR_LOCK(c->ctx);
    return solve_retry(c->ctx, [&]() -> int { return chol_inverse_impl(c, out, ldo); });
}

static int chol_inverse_impl(fr_chol* c, double* out, int64_t ldo)
{
    fr_ctx* ctx = c->ctx;
    FR_HIP(ctx, hipSetDevice(ctx->device));
    Staged o(ctx);
    FR_TRY(o.out(out, c->n, c->n, ldo));
    if (c->n > TRINV_BASE && !c->refine && ctx->tri_inverse) {
        // K^-1 = W^T W with W = L^-1 (2 n^3 / 3 flop, the structural zeros skipped: chol_tri_inverse) instead of the two
        // solves of the identity (2 n^3); the lower triangle is computed, the upper one mirrored
        const int64_t ldw = round_up(c->n, kAlign);
        WsGuard wg(ctx);
        double* W = wg.get(sizeof(double) * (size_t)ldw * (size_t)c->n);
        if (!W) return FR_OUT_OF_MEMORY;
        FR_TRY(chol_tri_inverse(ctx, c, W, ldw, o.dev, FR_PROF_GEMM_SOLVE));  // (the output buffer is the scratch first: (n / 2 + 512)^2 < n^2)
        GemmDesc g;
        g.M = c->n; g.N = c->n; g.K = c->n;
        g.A = W; g.lda = ldw; g.a_kmajor = true;
        g.B = W; g.ldb = ldw; g.b_kmajor = true;
        g.Cin = o.dev; g.ldcin = o.ld; g.D = o.dev; g.ldd = o.ld;
        g.alpha = 1.0; g.beta = 0.0; g.lower = true; g.prof_cls = FR_PROF_GEMM_SOLVE;
        g.tri = 1; g.dynamic = true;
        FR_TRY(launch_gemm(ctx, g));
        FR_TRY(launch_symmetrize(ctx, o.dev, c->n, o.ld));
        return o.commit();
    }
    FR_TRY(launch_set_identity(ctx, o.dev, c->n, o.ld));
    FR_TRY(trsm_lower_fwd(ctx, c, c->n, o.dev, c->n, o.ld, FR_PROF_GEMM_SOLVE));
    FR_TRY(trsm_lower_bwd(ctx, c, c->n, o.dev, c->n, o.ld, FR_PROF_GEMM_SOLVE));
    return o.commit();
}

int fr_chol_download_l(fr_chol* c, double* out, int64_t ldo, int upper_fill)
{
    if (!c) return FR_INVALID_ARGUMENT;
    FR_LOCK(c->ctx);
    fr_ctx* ctx = c->ctx;
    FR_HIP(ctx, hipSetDevice(ctx->device));
    Staged o(ctx);
    FR_TRY(o.out(out, c->n, c->n, ldo));
    FR_TRY(launch_copy(ctx, c->A, c->ld_a, o.dev, o.ld, c->n, c->n));
    FR_TRY(launch_tri_fill(ctx, o.dev, c->n, o.ld, upper_fill ? std::nan("") : 0.0));
    return o.commit();
}

int fr_chol_upload_l(fr_ctx* ctx, const double* L, int64_t n, int64_t ldl, const double* X, int64_t ldx, int64_t d,
                     int64_t capacity_hint, fr_chol** out)
{
    if (!ctx || !out) return FR_INVALID_ARGUMENT;
    FR_LOCK(ctx);
    *out = nullptr;
    FR_HIP(ctx, hipSetDevice(ctx->device));
    if (n < 0 || d < 0 || ldl < imax(n, 1) || (d > 0 && ldx < imax(n, 1))) return set_err(ctx, FR_SHAPE, "bad shape");
    fr_chol* c = nullptr;
    FR_TRY(chol_alloc(ctx, n, imax(capacity_hint, n), d, &c));
    int st = upload_rows(ctx, L, ldl, c->A, c->ld_a, n, n);
    if (st == FR_OK && d > 0) st = upload_rows(ctx, X, ldx, c->X, c->ld_x, n, d);
    if (st == FR_OK) {
        hipError_t e = hipMemsetAsync(c->info, 0, sizeof(int64_t) * 3, ctx->stream);
        if (e != hipSuccess) st = set_err(ctx, FR_HIP_ERROR, "memset failed");
    }
    if (st == FR_OK) {
        for (int64_t b = 0; st == FR_OK && b * IB < n; ++b) {
            const int64_t j = b * IB, sb = imin(IB, n - j);
            st = invert_block128(ctx, c->A + j + j * c->ld_a, c->ld_a, sb, c->dinv + b * INV_ELEMS, c->cest + b);
        }
        if (st == FR_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = FR_HIP_ERROR;
        // the uploaded factor's conditioning decides whether this handle's solves take the refinement step (fr_ctx::refine)
        if (st == FR_OK) st = fetch_max_cest(c);
        if (st == FR_OK) c->refine = ctx->refine == 1 || (ctx->refine == -1 && c->max_cest > ctx->refine_threshold);
    }
    if (st != FR_OK) {
        fr_chol_free(c);
        return st;
    }
    *out = c;
    return FR_OK;
}

void fr_chol_free(fr_chol* c)
{
    if (!c) return;
    if (c->ctx) {
        FR_LOCK(c->ctx);
        (void)hipSetDevice(c->ctx->device);
        (void)hipStreamSynchronize(c->ctx->stream);
        chol_release(c);
    } else {
        chol_release(c);
    }
    delete c;
}

}  // extern "C"
