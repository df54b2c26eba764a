// panel.hip -- the fused panel factorisation of the blocked Cholesky (K5 of DESIGN.md, round 2).
//
// Reference work being replaced: the column loop of nalgebra's Cholesky::new_internal on the kb columns of a panel
// (src/algebra/mod.rs:81-91).  Round 1 ran a panel as a recursion of ~11 dependent launches per 512 columns -- four
// diagonal-block kernels next to the trailing update's MFMA stream (~150 - 265 us each instead of 60) and seven
// latency-bound GEMMs of ~64 us -- and that chain, not the trailing SYRK, bounded every fit up to N = 16384.
//
// Now a panel is ONE launch of row-tile workgroups plus the diagonal-block server (potf2.hip) that lives on a CU of its
// own for the whole factorisation:
//   * workgroup t owns rows [k + 128 t, k + 128 t + 128) of the panel's kb columns (four 128-column sub-panels at kb = 512);
//   * right-looking inside the panel: as soon as the server has factored diagonal block s (done[s]: L_ss and its explicit
//     inverse are in memory) the tile solves its sub-panel s against the inverse (one 128^3 MFMA tile product, in place)
//     and applies it to its later sub-panels s' > s, which needs L[s', s] from the workgroup of diagonal row tile s'
//     (tdone[s'] >= s + 1);
//   * the workgroup of diagonal row tile t publishes ready[t] when block (t, t) carries every update; the server takes
//     it from there.
// The serial chain per 128 columns is therefore: server (one block) -> hand-off -> TWO tile products in the workgroup of
// the next diagonal row tile -> hand-off -> server.  Everything else proceeds in their shadow, and a fit issues
// 3 launches per panel (row tiles, look-ahead update, trailing update).  Hand-offs: handoff.hpp.
#include "fr_internal.hpp"
#include "gemm_tile.hpp"
#include "handoff.hpp"

namespace fr {

constexpr int PTB = 128;
constexpr int SLH = 32;             // rows of a diagonal-region slice
constexpr int NSL = PTB / SLH;      // slices per diagonal row tile (= flags per 128-block in ready[] / tdone[])
static_assert(NSL == 4, "the server (potf2.hip, SERVER_NSL) and the flag layout (chol.hip) assume four slices per block");

struct PanelArgs {
    double* A;  // the matrix being factored (n x n, lower triangle)
    int64_t lda, n, k, kb;
    const double* dinv;  // inverse of diagonal block g at dinv + g * 128 * 128
    int* ready;          // NSL flags per block
    int* done;           // 1 flag per block
    int* tdone;          // NSL flags per block
    unsigned* status;
    int jb;   // k / 128: first diagonal block of the panel
    int nsb;  // sub-panels
    int s_lo, s_hi;  // rest kernel: the sub-panels [s_lo, s_hi) of this launch
    // XCD reservation (gemm_tile.hpp, claim_item): place 0 = row tile blockIdx.x, 1 = row tiles claimed by the workgroups
    // that do not sit on the panel stream's XCD
    int place, nres;
    unsigned epoch;
    const unsigned* xcc_word;
    unsigned* claim;
    unsigned max_exit;
    int64_t ntiles;
};

__device__ __forceinline__ void tile_product(double* lds, int64_t M, int64_t N, int64_t K, const double* A, int64_t lda,
                                             const double* B, int64_t ldb, double* D, int64_t ldd, double alpha, double beta)
{
    GemmArgs g;
    g.M = M;
    g.N = N;
    g.K = K;
    g.A = A;
    g.lda = lda;
    g.B = B;
    g.ldb = ldb;
    g.Cin = D;
    g.ldcin = ldd;
    g.D = D;
    g.ldd = ldd;
    g.alpha = alpha;
    g.beta = beta;
    g.lower = 0;
    g.own_world = 1;
    g.yield_word = nullptr;
    gemm_f64_tile<false, false>(g, lds, 0, 0);
}

// ---- the diagonal region: 32-row slices ----------------------------------------------------------------------------------
// D (M <= 32 rows x N <= 128 columns) = beta * D + alpha * A (M x K) * B^T (B: N x K),  element (m, k) of A at A[m + k lda],
// (n, k) of B at B[n + k ldb], K <= 128.  Four waves, 32 columns each, MFMA fragments straight from memory (the operands
// are a few KiB in L2): the whole product is 32 MFMA steps deep per wave -- a quarter of the 128 x 128 tile's -- and four
// such workgroups (one per 32-row slice) work on a diagonal row tile at once.  D may alias A (same rows): the stores wait
// behind a workgroup barrier.
__device__ __forceinline__ void slice_product(int64_t M, int64_t N, int64_t K, const double* __restrict__ A, int64_t lda,
                                              const double* __restrict__ B, int64_t ldb, double* D, int64_t ldd, double alpha,
                                              double beta)
{
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    d4_t acc[2][2];  // [nt][mt]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};
    int mrow[2], ncol[2];
    bool mok[2], nok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        mrow[i] = i * 16 + l15;
        ncol[i] = 32 * w + i * 16 + l15;
        mok[i] = mrow[i] < M;
        nok[i] = ncol[i] < N;
        if (!mok[i]) mrow[i] = 0;
        if (!nok[i]) ncol[i] = 0;
    }
    // K in chunks of 32: the 32 fragment loads of a chunk are all in flight before its 32 MFMAs
#pragma unroll 1
    for (int64_t k0 = 0; k0 < K; k0 += 32) {
        double af[8][2], bf[8][2];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t kk = k0 + 4 * u + lq;
            const bool kok = kk < K;
            const int64_t kc = kok ? kk : 0;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const double av = A[mrow[i] + kc * lda], bv = B[ncol[i] + kc * ldb];
                af[u][i] = (kok && mok[i]) ? av : 0.0;
                bf[u][i] = (kok && nok[i]) ? bv : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[u][nt], af[u][mt], acc[nt][mt], 0, 0, 0);
    }
    __syncthreads();  // D may alias A: every wave has its operands
    // accumulator register r of tile (nt, mt) holds D[m = mt * 16 + l15][n = 32 w + nt * 16 + lq + 4 r]
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int64_t m = mt * 16 + l15, nn = 32 * w + nt * 16 + lq + 4 * r;
                if (m < M && nn < N) {
                    double* d = D + m + nn * ldd;
                    double v = alpha * acc[nt][mt][r];
                    if (beta != 0.0) v += beta * *d;
                    *d = v;
                }
            }
}

// Workgroup 0 hands the panel's first diagonal block to the server (it carries its updates by stream order); workgroup
// 1 + NSL (t - 1) + a owns rows [32 a, 32 a + 32) of diagonal row tile t = 1 .. nsb - 1.
__global__ __launch_bounds__(256, 2) void panel_diag_kernel(const PanelArgs a)
{
    if (blockIdx.x == 0) {
        if (threadIdx.x < NSL) {
            __hip_atomic_store((hgi32*)(a.ready + NSL * a.jb + threadIdx.x), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    const int t = 1 + (int)(blockIdx.x - 1) / NSL, sl = (int)(blockIdx.x - 1) % NSL;
    const int64_t row0 = a.k + (int64_t)PTB * t + SLH * sl;
    int64_t rows = a.n - row0;
    rows = rows < 0 ? 0 : (rows < SLH ? rows : SLH);  // 0: the slice lies behind the matrix, only its flags are needed
    int* my_tdone = a.tdone + NSL * (a.jb + t) + sl;
    for (int s = 0; s < t; ++s) {
        const int64_t c0 = a.k + (int64_t)PTB * s;
        const int64_t cs = (a.kb - (int64_t)PTB * s) < PTB ? (a.kb - (int64_t)PTB * s) : PTB;
        double* Lts = a.A + row0 + c0 * a.lda;
        if (!handoff_wait_ge(a.done + a.jb + s, 1, a.status)) return;
        // L[t, s] (this slice) = A[t, s] L_ss^-T, a product with the explicit inverse, in place
        if (rows > 0) slice_product(rows, cs, cs, Lts, a.lda, a.dinv + (int64_t)(a.jb + s) * (PTB * PTB), PTB, Lts, a.lda, 1.0, 0.0);
        handoff_publish(my_tdone, s + 1);
        for (int sp = s + 1; sp <= t; ++sp) {
            // A[t, sp] -= L[t, s] L[sp, s]^T needs every slice of L[sp, s]
            if (!handoff_wait_all_ge(a.tdone + NSL * (a.jb + sp), NSL, s + 1, a.status)) return;
            const int64_t cp0 = a.k + (int64_t)PTB * sp;
            int64_t cp = (a.kb - (int64_t)PTB * sp) < PTB ? (a.kb - (int64_t)PTB * sp) : PTB;
            if (sp == t && cp > (int64_t)SLH * (sl + 1)) cp = (int64_t)SLH * (sl + 1);  // own diagonal block: lower part only
            if (rows > 0)
                slice_product(rows, cp, cs, Lts, a.lda, a.A + cp0 + c0 * a.lda, a.lda, a.A + row0 + cp0 * a.lda, a.lda, -1.0, 1.0);
            __syncthreads();
        }
    }
    handoff_publish(a.ready + NSL * (a.jb + t) + sl, 1);  // this slice of block (t, t) is final: over to the server
}

// ---- the rows below the diagonal block: one workgroup per 128-row tile, launched behind the diagonal kernel (every
// diagonal block of the panel is factored, every inverse in memory: no waiting inside).  Left-looking, so that the
// updates of a sub-panel are ONE product of depth 128 s instead of s products of depth 128.
__global__ __launch_bounds__(256, 2) void panel_rest_kernel(const PanelArgs a)
{
    __shared__ double lds[4 * TILE_ELEMS];
    long long item = blockIdx.x;
    if (a.place) {
        item = claim_item(a.place, a.nres, a.epoch, a.xcc_word, a.claim, a.max_exit, a.ntiles);
        if (item < 0) return;
    }
    const int t = a.nsb + (int)item;
    const int64_t row0 = a.k + (int64_t)PTB * t;
    const int64_t rows = (a.n - row0) < PTB ? (a.n - row0) : PTB;
    // ops 2 s (update of sub-panel s by sub-panels < s) and 2 s + 1 (solve): ONE call site of the tile product
#pragma nounroll
    for (int op = (a.s_lo == 0 ? 1 : 2 * a.s_lo); op < 2 * a.s_hi; ++op) {
        const int s = op >> 1;
        const bool solve = (op & 1) != 0;
        const int64_t c0 = a.k + (int64_t)PTB * s;
        const int64_t cs = (a.kb - (int64_t)PTB * s) < PTB ? (a.kb - (int64_t)PTB * s) : PTB;
        double* Ats = a.A + row0 + c0 * a.lda;
        tile_product(lds, rows, cs, solve ? cs : (int64_t)PTB * s, solve ? Ats : a.A + row0 + a.k * a.lda, a.lda,
                     solve ? a.dinv + (int64_t)(a.jb + s) * (PTB * PTB) : a.A + c0 + a.k * a.lda, solve ? (int64_t)PTB : a.lda, Ats,
                     a.lda, solve ? 1.0 : -1.0, solve ? 0.0 : 1.0);
        __syncthreads();
    }
}

// The rows below the diagonal block alone (option panel_fused = 4: the diagonal block was factored by the per-block launches)
int launch_panel_rest(fr_ctx* ctx, double* A, int64_t lda, int64_t n, int64_t k, int64_t kb, const double* dinv)
{
    const int64_t T = (n - k + PTB - 1) / PTB;
    const int nsb = (int)((kb + PTB - 1) / PTB);
    if (T <= nsb) return FR_OK;
    PanelArgs a;
    a.A = A;
    a.lda = lda;
    a.n = n;
    a.k = k;
    a.kb = kb;
    a.dinv = dinv;
    a.ready = a.done = a.tdone = nullptr;
    a.status = ctx->dev_status;
    a.jb = (int)(k / PTB);
    a.nsb = nsb;
    a.s_lo = 0;
    a.s_hi = nsb;
    a.place = 0;
    const double rows = (double)(n - k - kb);
    ProfScope ps(ctx, FR_PROF_GEMM_PANEL, rows * (double)kb * (double)kb, 8.0 * rows * (double)kb * 2.0);
    hipLaunchKernelGGL(panel_rest_kernel, dim3((unsigned)(T - nsb)), dim3(256), 0, ctx->ls, a);
    FR_HIP(ctx, hipGetLastError());
    return FR_OK;
}

// Sub-panels [s_lo, s_hi) of the rows below the panel's diagonal block, for the critical-path schedule (chol.hip,
// factor_panel_cp): sub-panel s needs the diagonal blocks <= s factored and L(diagonal-block rows, sub-panels < s) final.
int launch_panel_rest_cols(fr_ctx* ctx, double* A, int64_t lda, int64_t n, int64_t k, int64_t kb, const double* dinv, int s_lo,
                           int s_hi)
{
    const int64_t T = (n - k + PTB - 1) / PTB;
    const int nsb = (int)((kb + PTB - 1) / PTB);
    if (T <= nsb || s_hi <= s_lo) return FR_OK;
    PanelArgs a;
    a.A = A;
    a.lda = lda;
    a.n = n;
    a.k = k;
    a.kb = kb;
    a.dinv = dinv;
    a.ready = a.done = a.tdone = nullptr;
    a.status = ctx->dev_status;
    a.jb = (int)(k / PTB);
    a.nsb = nsb;
    a.s_lo = s_lo;
    a.s_hi = s_hi;
    a.ntiles = T - nsb;
    int64_t grid = a.ntiles;
    a.place = claim_setup(ctx, 1, a.ntiles, &a.xcc_word, &a.claim, &a.max_exit, &grid);
    a.nres = ctx->reserve_now;
    a.epoch = ctx->panel_epoch;
    const double rows = (double)(n - k - kb);
    double fl = 0.0;
    for (int s = s_lo; s < s_hi; ++s) fl += 2.0 * rows * 128.0 * 128.0 * (double)(s + 1);
    ProfScope ps(ctx, FR_PROF_GEMM_PANEL, fl, 8.0 * rows * 128.0 * (double)(s_hi + 1));
    hipLaunchKernelGGL(panel_rest_kernel, dim3((unsigned)grid), dim3(256), 0, ctx->ls, a);
    FR_HIP(ctx, hipGetLastError());
    return FR_OK;
}

// ---- the chain step between two diagonal-block kernels (chol.hip, factor_panel_rl): ONE workgroup takes row tile j of the
// panel from "every update of the blocks before j - 1 applied" to "diagonal block (j, j) ready to be factored":
//   L[j, j-1] = A[j, j-1] W_{j-1}^T (in place),   A[j, j] -= L[j, j-1] L[j, j-1]^T.
// Two dependent 128^3 products on one CU (one call site: see panel_rest_kernel).
__global__ __launch_bounds__(256, 2) void panel_step_kernel(double* A, int64_t lda, int64_t r0, int64_t c0, int rows, int cols,
                                                            const double* __restrict__ W)
{
    __shared__ double lds[4 * TILE_ELEMS];
    double* L = A + r0 + c0 * lda;
#pragma nounroll
    for (int op = 0; op < 2; ++op) {
        const bool solve = op == 0;
        tile_product(lds, rows, solve ? cols : rows, cols, L, lda, solve ? W : L, solve ? (int64_t)PTB : lda,
                     solve ? L : A + r0 + r0 * lda, lda, solve ? 1.0 : -1.0, solve ? 0.0 : 1.0);
        __syncthreads();
    }
}

// The same step on FOUR workgroups (32-row slices of the row tile, MFMA fragments straight from L2: a slice product is a
// quarter as deep as the 128 x 128 tile's): solve, hand the slice over (the update needs every row of L[j, j-1]), update.
// Launched with 32 workgroups of which those with blockIdx.x % 8 == 0 work: the panel stream's own XCD (gemm_f64.hip).
__global__ __launch_bounds__(256, 2) void panel_step4_kernel(double* A, int64_t lda, int64_t r0, int64_t c0, int rows, int cols,
                                                             const double* __restrict__ W, int* flags, int target, unsigned* status)
{
    if (blockIdx.x & 7) return;
    const int sl = (int)(blockIdx.x >> 3);
    const int64_t row0 = r0 + (int64_t)SLH * sl;
    int m = rows - SLH * sl;
    m = m < 0 ? 0 : (m < SLH ? m : SLH);
    double* L = A + row0 + c0 * lda;
    if (m > 0) slice_product(m, cols, cols, L, lda, W, PTB, L, lda, 1.0, 0.0);
    handoff_publish(flags + sl, target);
    if (!handoff_wait_all_ge(flags, NSL, target, status)) return;
    if (m > 0) {
        int nc = SLH * (sl + 1);  // the lower part of the diagonal block only
        nc = nc < rows ? nc : rows;
        slice_product(m, nc, cols, L, lda, A + r0 + c0 * lda, lda, A + row0 + r0 * lda, lda, -1.0, 1.0);
    }
}

int launch_panel_step(fr_ctx* ctx, double* A, int64_t lda, int64_t n, int64_t r0, int64_t c0, int64_t cols, const double* W)
{
    const int64_t rows = (n - r0) < PTB ? (n - r0) : PTB;
    if (rows <= 0) return FR_OK;
    ProfScope ps(ctx, FR_PROF_GEMM_PANEL, 2.0 * (double)rows * (double)cols * (double)(cols + rows), 8.0 * 3.0 * 128.0 * 128.0);
    if (ctx->panel_rl == 2 && ctx->step_flags) {
        FR_TRY(ensure_status_word(ctx));
        hipLaunchKernelGGL(panel_step4_kernel, dim3(8 * NSL), dim3(256), 0, ctx->ls, A, lda, r0, c0, (int)rows, (int)cols, W,
                           ctx->step_flags, ++ctx->step_epoch, ctx->dev_status);
    } else {
        hipLaunchKernelGGL(panel_step_kernel, dim3(1), dim3(256), 0, ctx->ls, A, lda, r0, c0, (int)rows, (int)cols, W);
    }
    FR_HIP(ctx, hipGetLastError());
    return FR_OK;
}

int launch_panel_tiles(fr_ctx* ctx, double* A, int64_t lda, int64_t n, int64_t k, int64_t kb, const double* dinv, int* ready,
                       int* done, int* tdone)
{
    if (kb <= 0 || k >= n) return FR_OK;
    PanelArgs a;
    a.A = A;
    a.lda = lda;
    a.n = n;
    a.k = k;
    a.kb = kb;
    a.dinv = dinv;
    a.ready = ready;
    a.done = done;
    a.tdone = tdone;
    a.status = ctx->dev_status;
    a.jb = (int)(k / PTB);
    a.nsb = (int)((kb + PTB - 1) / PTB);
    a.s_lo = 0;
    a.s_hi = a.nsb;
    a.place = 0;
    const int64_t T = (n - k + PTB - 1) / PTB;
    const double rows = (double)(n - k);
    ProfScope ps(ctx, FR_PROF_GEMM_PANEL, rows * (double)kb * (double)kb, 8.0 * rows * (double)kb * 2.0);
    hipLaunchKernelGGL(panel_diag_kernel, dim3((unsigned)(1 + NSL * (a.nsb - 1))), dim3(256), 0, ctx->ls, a);
    if (T > a.nsb) hipLaunchKernelGGL(panel_rest_kernel, dim3((unsigned)(T - a.nsb)), dim3(256), 0, ctx->ls, a);
    FR_HIP(ctx, hipGetLastError());
    return FR_OK;
}

}  // namespace fr
