// gemm_f64.hip -- K5/K6: the FP64 matrix-core GEMM behind the blocked Cholesky (trailing SYRK update,
// TRSM recast as GEMM with inverted diagonal blocks), the triangular solves of predict / predict_variance /
// sample_at (Cholesky::solve_mut, solve_lower_triangular: mod.rs:235,260-263,298,342-345,379) and the
// gemm_tr contractions (mod.rs:348,383).  The reference runs these as nalgebra's unblocked axpy/dot loops.
//
//   D = alpha * op(A) * op(B) + beta * Cin          (all column-major f64, D may alias Cin)
//
// Design (gfx950): one workgroup = 4 wave64 in a 2x2 arrangement computes a 128x128 tile, each wave a
// 64x64 sub-tile as 4x4 v_mfma_f64_16x16x4_f64 accumulators (128 VGPRs).  K is consumed in steps of 16
// through double-buffered LDS; the next step's global loads are issued before the current step's MFMAs
// and written to the other LDS buffer after them (one barrier per step).  FP64 MFMA is slow enough
// (16x16x4 every 64 cycles per SIMD) that 8-byte global loads suffice, which removes every alignment
// requirement on sub-matrix base pointers (row offsets are arbitrary after add_samples).
//
// LDS layouts are chosen per operand so that global reads stay coalesced and the MFMA fragment reads
// (ds_read_b64, two 32-lane groups, 32 eight-byte slots) are conflict-free:
//   "m-major" operand (element (m,k) at P[m + k*ld]):  tile[k][m], row stride 144  (144 mod 32 == 16)
//   "k-major" operand (element (m,k) at P[k + m*ld]):  tile[m][k], row stride 18   (18*m mod 32 distinct, even)
//
// The MFMA is issued with the roles swapped (B-fragment as the A operand) so that the 16 lanes (lane & 15)
// of an accumulator register run along m, the contiguous axis of column-major D: every store / Cin load
// instruction moves four 128-byte segments.
#include "fr_internal.hpp"

namespace fr {

typedef double d4_t __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int S_MMAJ = 144;  // [BK][144]
constexpr int S_KMAJ = 18;   // [128][18]
constexpr int TILE_ELEMS = 2304;  // 16*144 == 128*18

struct GemmArgs {
    int64_t M, N, K;
    const double* A;
    int64_t lda;
    const double* B;
    int64_t ldb;
    const double* Cin;
    int64_t ldcin;
    double* D;
    int64_t ldd;
    double alpha, beta;
    int lower;
    int64_t tiles_m, tiles_n;
    int sw_log2;                       // super-tile = (64 >> sw_log2) x (1 << sw_log2) tiles
    int64_t super_m, nsuper, per_xcd;  // super-tiles along m, in total, per XCD
    // multi-GPU column ownership: a tile is computed only by the rank owning its block column
    // ((own_col0 + n0) / own_nb) % own_world == own_rank; own_world <= 1 disables the filter
    int own_world, own_rank;
    int64_t own_nb, own_col0;
    // batch: blockIdx.y selects a problem; operands advance by these strides (elements)
    int64_t batch_a, batch_b, batch_c, batch_d;
};

// element (x, k) of an operand tile; x is the m (or n) index inside the 128-wide tile
template <bool KMAJ>
__device__ __forceinline__ int lds_idx(int x, int k)
{
    return KMAJ ? x * S_KMAJ + k : k * S_MMAJ + x;
}

template <bool KMAJ>
__device__ __forceinline__ void load_tile(const double* __restrict__ P, int64_t ld, int64_t x0, int64_t X, int64_t k0,
                                          int64_t K, int t, double (&reg)[8])
{
    if (!KMAJ) {
        const int x = t & 127;
        const bool xok = (x0 + x) < X;
        const double* p = P + (x0 + x) + k0 * ld;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = (t >> 7) + 2 * i;
            reg[i] = (xok && (k0 + k) < K) ? p[(int64_t)k * ld] : 0.0;
        }
    } else {
        const int k = t & 15;
        const bool kok = (k0 + k) < K;
        const double* p = P + (k0 + k) + x0 * ld;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int x = (t >> 4) + 16 * i;
            reg[i] = (kok && (x0 + x) < X) ? p[(int64_t)x * ld] : 0.0;
        }
    }
}

// Branch-free variant for tiles that lie entirely inside the operand (every tile but the last row / column of
// tiles, every K-step but a ragged last one).  The predicated loader above compiles to one exec-masked branch per
// element -- ~300 scalar/branch instructions per K-step in front of the MFMAs -- so the hot path must avoid it.
template <bool KMAJ>
__device__ __forceinline__ void load_tile_fast(const double* __restrict__ p, int64_t ld, double (&reg)[8])
{
#pragma unroll
    for (int i = 0; i < 8; ++i) reg[i] = p[(int64_t)(KMAJ ? 16 * i : 2 * i) * ld];
}

template <bool KMAJ>
__device__ __forceinline__ const double* tile_thread_base(const double* P, int64_t ld, int64_t x0, int t)
{
    return KMAJ ? P + (t & 15) + (x0 + (t >> 4)) * ld : P + (x0 + (t & 127)) + (int64_t)(t >> 7) * ld;
}

template <bool KMAJ>
__device__ __forceinline__ void store_tile(double* __restrict__ S, int t, const double (&reg)[8])
{
    if (!KMAJ) {
        const int x = t & 127;
#pragma unroll
        for (int i = 0; i < 8; ++i) S[((t >> 7) + 2 * i) * S_MMAJ + x] = reg[i];
    } else {
        const int k = t & 15;
#pragma unroll
        for (int i = 0; i < 8; ++i) S[((t >> 4) + 16 * i) * S_KMAJ + k] = reg[i];
    }
}

template <bool A_KMAJ, bool B_KMAJ>
__device__ __forceinline__ void gemm_f64_body(const GemmArgs& g, double* lds)
{

    // Tile assignment.  Block b runs on XCD b % 8 (observed dispatch rule; used for speed only).  Each XCD gets a
    // contiguous run of 64-tile "super-tiles" (8 x 8 tiles): an XCD keeps ~64 workgroups resident (32 CUs x 2), i.e. about
    // one super-tile at a time, whose K-slices of 8 + 8 operand panels are each fetched into the XCD's L2 once and hit 7
    // times -- with a row-major tile order every B slice misses (L2 hit rate 46 %, measured) and the fabric traffic
    // (operand misses + the C read-modify-write) approaches what HBM/Infinity Cache can deliver.
    // Deep contractions (the recursive solves, K up to n/2) keep the plain order instead: over hundreds of K-steps the
    // workgroups of a super-tile drift apart and 16 live panels no longer fit the 4 MiB L2 (measured: -35 % on predict).
    const int64_t b = blockIdx.x;
    const int64_t xcd = b & 7;
    int64_t tm, tn;
    if (g.nsuper > 0) {
        // super-tiles are dealt round-robin to the XCDs (the triangular enumeration of the lower mode has cheaper
        // super-tiles on the diagonal: contiguous runs per XCD would leave a 6 % imbalance)
        const int64_t sidx = ((b >> 3) >> 6) * 8 + xcd;
        const int within = (int)((b >> 3) & 63);
        if (sidx >= g.nsuper) return;
        if (g.lower) {
            int64_t row = (int64_t)((sqrt(8.0 * (double)sidx + 1.0) - 1.0) * 0.5);
            while (row * (row + 1) / 2 > sidx) --row;
            while ((row + 1) * (row + 2) / 2 <= sidx) ++row;
            tm = row * 8 + (within & 7);
            tn = (sidx - row * (row + 1) / 2) * 8 + (within >> 3);
            if (tn > tm || tm >= g.tiles_m) return;
        } else {
            const int sh = 64 >> g.sw_log2;
            tm = (sidx % g.super_m) * sh + (within & (sh - 1));
            tn = (sidx / g.super_m) * (1 << g.sw_log2) + (within >> (6 - g.sw_log2));
            if (tm >= g.tiles_m || tn >= g.tiles_n) return;
        }
    } else {
        // plain order: each XCD gets a contiguous run of the tile list
        const int64_t nblk = gridDim.x;
        const int64_t q = nblk >> 3, r8 = nblk & 7;
        const int64_t tlin = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (b >> 3);
        if (g.lower == 2) {
            // lower triangle column by column (tm fastest), like the full mode: consecutive workgroups continue down the
            // same 128 columns of C (the next 1 KiB of every column) and share one B panel.  Column tn holds T - tn tiles.
            const int64_t T = g.tiles_m;
            const double tt = 2.0 * (double)T + 1.0;
            int64_t col = (int64_t)((tt - sqrt(tt * tt - 8.0 * (double)tlin)) * 0.5);
            if (col < 0) col = 0;
            if (col >= T) col = T - 1;
            while (col > 0 && col * T - col * (col - 1) / 2 > tlin) --col;
            while ((col + 1) * T - (col + 1) * col / 2 <= tlin) ++col;
            tn = col;
            tm = col + (tlin - (col * T - col * (col - 1) / 2));
        } else if (g.lower) {
            int64_t row = (int64_t)((sqrt(8.0 * (double)tlin + 1.0) - 1.0) * 0.5);
            while (row * (row + 1) / 2 > tlin) --row;
            while ((row + 1) * (row + 2) / 2 <= tlin) ++row;
            tm = row;
            tn = tlin - row * (row + 1) / 2;
        } else {
            tm = tlin % g.tiles_m;
            tn = tlin / g.tiles_m;
        }
    }
    const int64_t m0 = tm * BM, n0 = tn * BN;
    if (g.own_world > 1 && (int)(((g.own_col0 + n0) / g.own_nb) % g.own_world) != g.own_rank) return;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int l15 = lane & 15, lq = lane >> 4;

    d4_t acc[4][4];  // [nt][mt]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};

    const int64_t nk = (g.K + BK - 1) / BK;
    const int64_t nk_full = g.K / BK;  // K-steps that need no k predicate
    const bool a_fast = (m0 + BM) <= g.M, b_fast = (n0 + BN) <= g.N;
    // per-thread pointers of the branch-free loader, advanced by one K-step at a time
    const double* pa = tile_thread_base<A_KMAJ>(g.A, g.lda, m0, t);
    const double* pb = tile_thread_base<B_KMAJ>(g.B, g.ldb, n0, t);
    const int64_t step_a = A_KMAJ ? BK : BK * g.lda, step_b = B_KMAJ ? BK : BK * g.ldb;
    double ra[8], rb[8];
    if (nk > 0) {
        if (a_fast && nk_full > 0)
            load_tile_fast<A_KMAJ>(pa, g.lda, ra);
        else
            load_tile<A_KMAJ>(g.A, g.lda, m0, g.M, 0, g.K, t, ra);
        // op(B) element (k, n): B_KMAJ -> B[k + n*ldb] (k contiguous), else B[n + k*ldb]
        if (b_fast && nk_full > 0)
            load_tile_fast<B_KMAJ>(pb, g.ldb, rb);
        else
            load_tile<B_KMAJ>(g.B, g.ldb, n0, g.N, 0, g.K, t, rb);
        store_tile<A_KMAJ>(lds, t, ra);
        store_tile<B_KMAJ>(lds + TILE_ELEMS, t, rb);
    }
    __syncthreads();

    int cur = 0;
    if (a_fast && b_fast && nk_full == nk) {
        // Interior tile, K a multiple of 16: one branch-free basic block per K-step, with the instruction order
        // pinned by sched_group_barrier.  A wave issues its next MFMA only when the matrix pipe is free (64 cycles
        // each), so everything else -- the 16 global loads of the next K-slice, the LDS fragment reads of the next
        // k-substep, the 16 LDS stores of the prefetched slice -- is slotted BETWEEN MFMAs instead of in front of /
        // behind the 64-MFMA block, which leaves only the barrier and the first fragment read exposed per K-step.
        // The last K-step re-loads its own slice (pointer not advanced) into the unused buffer: harmless, and it
        // keeps the loop body free of branches.
        for (int64_t kt = 0; kt < nk; ++kt) {
            const bool more = (kt + 1) < nk;
            pa += more ? step_a : 0;
            pb += more ? step_b : 0;
            load_tile_fast<A_KMAJ>(pa, g.lda, ra);
            load_tile_fast<B_KMAJ>(pb, g.ldb, rb);
            const double* As = lds + cur * 2 * TILE_ELEMS;
            const double* Bs = As + TILE_ELEMS;
#pragma unroll
            for (int ks = 0; ks < BK / 4; ++ks) {
                const int kq = ks * 4 + lq;
                double af[4], bf[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    af[i] = As[lds_idx<A_KMAJ>(wm * 64 + i * 16 + l15, kq)];
                    bf[i] = Bs[lds_idx<B_KMAJ>(wn * 64 + i * 16 + l15, kq)];
                }
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        acc[nt][mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[nt], af[mt], acc[nt][mt], 0, 0, 0);
            }
            double* An = lds + (cur ^ 1) * 2 * TILE_ELEMS;
            store_tile<A_KMAJ>(An, t, ra);
            store_tile<B_KMAJ>(An + TILE_ELEMS, t, rb);
            // pipeline description (masks: 0x008 MFMA, 0x020 VMEM read, 0x100 DS read, 0x200 DS write)
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);  // fragments of k-substep 0
#pragma unroll
            for (int j = 0; j < 8; ++j) {  // k-substep 0: 16 MFMA + the 16 global loads (+ fragments of substep 1)
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
                if (j >= 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
#pragma unroll
            for (int ks = 1; ks < 3; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) {  // k-substeps 1, 2: 16 MFMA + the fragments of the next substep
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    if (j >= 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
#pragma unroll
            for (int j = 0; j < 8; ++j) {  // k-substep 3: 16 MFMA + the 16 LDS stores of the prefetched slice
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
            }
            __syncthreads();
            cur ^= 1;
        }
    } else {
        for (int64_t kt = 0; kt < nk; ++kt) {
            const bool more = (kt + 1) < nk;
            if (more) {
                pa += step_a;
                pb += step_b;
                const bool kfull = (kt + 1) < nk_full;
                if (a_fast && kfull)
                    load_tile_fast<A_KMAJ>(pa, g.lda, ra);
                else
                    load_tile<A_KMAJ>(g.A, g.lda, m0, g.M, (kt + 1) * BK, g.K, t, ra);
                if (b_fast && kfull)
                    load_tile_fast<B_KMAJ>(pb, g.ldb, rb);
                else
                    load_tile<B_KMAJ>(g.B, g.ldb, n0, g.N, (kt + 1) * BK, g.K, t, rb);
            }
            const double* As = lds + cur * 2 * TILE_ELEMS;
            const double* Bs = As + TILE_ELEMS;
#pragma unroll
            for (int ks = 0; ks < BK / 4; ++ks) {
                const int kq = ks * 4 + lq;
                double af[4], bf[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    af[i] = As[lds_idx<A_KMAJ>(wm * 64 + i * 16 + l15, kq)];
                    bf[i] = Bs[lds_idx<B_KMAJ>(wn * 64 + i * 16 + l15, kq)];
                }
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        acc[nt][mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[nt], af[mt], acc[nt][mt], 0, 0, 0);
            }
            if (more) {
                double* An = lds + (cur ^ 1) * 2 * TILE_ELEMS;
                store_tile<A_KMAJ>(An, t, ra);
                store_tile<B_KMAJ>(An + TILE_ELEMS, t, rb);
            }
            __syncthreads();
            cur ^= 1;
        }
    }

    // epilogue: accumulator register r of tile (nt, mt) holds D[m = .. + (lane&15)][n = .. + (lane>>4) + 4r]
    //
    // D may alias Cin, so the compiler must keep every Cin load behind the preceding D stores; a naive
    // load-modify-store loop therefore pays one full memory round trip per element (64 per lane).  The loads
    // of a whole 16-column strip are issued back to back into registers, one strip ahead of the stores.
    const bool use_c = g.beta != 0.0;
    const bool interior = a_fast && b_fast;  // whole 128 x 128 tile inside D: no per-element predicate
    auto c_index = [&](int nt, int r, int mt, int64_t& m, int64_t& n) {
        n = n0 + wn * 64 + nt * 16 + lq + 4 * r;
        m = m0 + wm * 64 + mt * 16 + l15;
    };
    if (interior && use_c) {
        // Interior tile: straight-line code, THREE of the four 16-column strips of C in flight at once (the registers of
        // the K-loop's prefetch / fragment buffers are dead here); the fourth is issued as soon as strip 0 is stored.
        // Measured before this change: the epilogue was 38 % of a K = 512 tile's lifetime (4 dependent round trips).
        const double* cbase = g.Cin + (m0 + wm * 64 + l15) + (n0 + wn * 64 + lq) * g.ldcin;
        double* dbase = g.D + (m0 + wm * 64 + l15) + (n0 + wn * 64 + lq) * g.ldd;
        double c0[16], c1[16], c2[16];
        auto ld_strip = [&](int nt, double (&c)[16]) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) c[r * 4 + mt] = cbase[mt * 16 + (int64_t)(nt * 16 + 4 * r) * g.ldcin];
        };
        auto st_strip = [&](int nt, const double (&c)[16]) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    dbase[mt * 16 + (int64_t)(nt * 16 + 4 * r) * g.ldd] = g.alpha * acc[nt][mt][r] + g.beta * c[r * 4 + mt];
        };
        ld_strip(0, c0);
        ld_strip(1, c1);
        ld_strip(2, c2);
        st_strip(0, c0);
        ld_strip(3, c0);
        st_strip(1, c1);
        st_strip(2, c2);
        st_strip(3, c0);
        return;
    }
    double cv[2][16];
    auto load_strip = [&](int nt, double (&c)[16]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                int64_t m, n;
                c_index(nt, r, mt, m, n);
                c[r * 4 + mt] = (n < g.N && m < g.M) ? g.Cin[m + n * g.ldcin] : 0.0;
            }
        }
    };
    if (use_c) load_strip(0, cv[0]);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        if (use_c && nt + 1 < 4) load_strip(nt + 1, cv[(nt + 1) & 1]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                int64_t m, n;
                c_index(nt, r, mt, m, n);
                double v = g.alpha * acc[nt][mt][r];
                if (use_c) v += g.beta * cv[nt & 1][r * 4 + mt];
                if (n < g.N && m < g.M) g.D[m + n * g.ldd] = v;
            }
        }
    }
}

// Two kernel symbols over the same body: the lower-mode launch is the trailing SYRK update of the factorisation (the
// dominant kernel of a fit); keeping it apart from the panel / solve GEMMs makes profiler per-kernel averages meaningful.
template <bool A_KMAJ, bool B_KMAJ>
__global__ __launch_bounds__(256, 2) void gemm_f64_kernel(const GemmArgs g0)
{
    __shared__ double lds[4 * TILE_ELEMS];  // [stage][A|B][TILE_ELEMS]
    GemmArgs g = g0;
    const int64_t bz = blockIdx.y;  // batch member (0 for a plain launch)
    g.A += bz * g.batch_a;
    g.B += bz * g.batch_b;
    g.Cin += bz * g.batch_c;
    g.D += bz * g.batch_d;
    gemm_f64_body<A_KMAJ, B_KMAJ>(g, lds);
}

__global__ __launch_bounds__(256, 2) void syrk_lower_f64_kernel(const GemmArgs g)
{
    __shared__ double lds[4 * TILE_ELEMS];
    gemm_f64_body<false, false>(g, lds);
}

// D = beta * Cin + sum over slices of the partial products (split-K), slices M x N with leading dimension M
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const double* __restrict__ part, int64_t M, int64_t N, int slices,
                                                            const double* __restrict__ cin, int64_t ldcin, double beta,
                                                            double* __restrict__ out, int64_t ldd, int lower)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t j = blockIdx.y;
    if (i >= M || j >= N) return;
    if (lower && (i >> 7) < (j >> 7)) return;  // tiles strictly above the diagonal are not part of a lower-mode result
    double acc = 0.0;
    const double* p = part + i + j * M;
    for (int s = 0; s < slices; ++s) acc += p[(int64_t)s * M * N];
    out[i + j * ldd] = (beta != 0.0 ? beta * cin[i + j * ldcin] : 0.0) + acc;
}

static int launch_gemm_plain(fr_ctx* ctx, const GemmDesc& d);

// Few result tiles and a deep contraction (a 512-row block against 8192 columns: add_rows, narrow predicts): one tile's
// K-loop is then the whole run time while most CUs idle.  The contraction is cut into slices computed as one batched
// launch into a workspace, and a second small kernel adds them up (fixed order).  Main stream only (the workspace pool
// relies on stream order), single GPU ownership only.
int launch_gemm(fr_ctx* ctx, const GemmDesc& d)
{
    if (d.M <= 0 || d.N <= 0) return FR_OK;
    const int64_t tiles = ((d.M + BM - 1) / BM) * ((d.N + BN - 1) / BN);
    if (d.batch <= 1 && d.own_world <= 1 && ctx->ls == ctx->stream && ctx->splitk != 0 && tiles <= 192 && d.K >= 2048 &&
        d.M <= 65535 * 256) {
        int64_t S = (384 + tiles - 1) / tiles;
        if (S > d.K / 256) S = d.K / 256;
        if (S > 32) S = 32;
        while (S > 1 && (d.K % S != 0 || (d.K / S) % BK != 0)) --S;
        if (S > 1) {
            WsGuard w(ctx);
            double* part = w.get(sizeof(double) * (size_t)S * (size_t)d.M * (size_t)d.N);
            if (!part) return FR_OUT_OF_MEMORY;
            const int64_t ks = d.K / S;
            GemmDesc p = d;
            p.K = ks;
            p.lower = false;
            p.Cin = part; p.ldcin = d.M; p.D = part; p.ldd = d.M;
            p.beta = 0.0;
            p.batch = S;
            p.batch_a = d.a_kmajor ? ks : ks * d.lda;
            p.batch_b = d.b_kmajor ? ks : ks * d.ldb;
            p.batch_c = p.batch_d = d.M * d.N;
            FR_TRY(launch_gemm_plain(ctx, p));
            const double* cin = d.Cin ? d.Cin : d.D;
            const int64_t ldcin = d.Cin ? d.ldcin : d.ldd;
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((d.M + 255) / 256), (unsigned)d.N), dim3(256), 0, ctx->ls,
                               part, d.M, d.N, (int)S, cin, ldcin, d.beta, d.D, d.ldd, d.lower ? 1 : 0);
            FR_HIP(ctx, hipGetLastError());
            return FR_OK;
        }
    }
    return launch_gemm_plain(ctx, d);
}

static int launch_gemm_plain(fr_ctx* ctx, const GemmDesc& d)
{
    if (d.M <= 0 || d.N <= 0) return FR_OK;
    GemmArgs g;
    g.M = d.M;
    g.N = d.N;
    g.K = d.K < 0 ? 0 : d.K;
    g.A = d.A;
    g.lda = d.lda;
    g.B = d.B;
    g.ldb = d.ldb;
    g.Cin = d.Cin ? d.Cin : d.D;
    g.ldcin = d.Cin ? d.ldcin : d.ldd;
    g.D = d.D;
    g.ldd = d.ldd;
    g.alpha = d.alpha;
    g.beta = d.beta;
    g.lower = d.lower ? ((ctx->gemm_tile == 5) ? 2 : 1) : 0;
    g.own_world = d.own_world;
    g.own_rank = d.own_rank;
    g.own_nb = d.own_nb > 0 ? d.own_nb : 1;
    g.own_col0 = d.own_col0;
    g.tiles_m = (d.M + BM - 1) / BM;
    g.tiles_n = (d.N + BN - 1) / BN;
    double flops;
    int64_t ntiles;
    if (d.lower) {
        if (d.M != d.N) return set_err(ctx, FR_INVALID_ARGUMENT, "lower-mode GEMM needs a square result");
        g.sw_log2 = 3;
        g.super_m = (g.tiles_m + 7) / 8;
        g.nsuper = g.super_m * (g.super_m + 1) / 2;
        ntiles = g.tiles_m * (g.tiles_m + 1) / 2;
        flops = (double)d.M * (double)(d.M + 1) * (double)g.K;  // 2 * M(M+1)/2 * K
    } else {
        int swl = 0;
        while ((1 << swl) < g.tiles_n && swl < 3) ++swl;
        g.sw_log2 = swl;
        const int64_t sh = 64 >> swl, sw = 1 << swl;
        g.super_m = (g.tiles_m + sh - 1) / sh;
        g.nsuper = g.super_m * ((g.tiles_n + sw - 1) / sw);
        ntiles = g.tiles_m * g.tiles_n;
        flops = 2.0 * (double)d.M * (double)d.N * (double)g.K;
    }
    g.per_xcd = (g.nsuper + 7) / 8;
    // measured inside the factorisation (scripts/order_ab.py): super-tiles pay for large shallow full products only; the
    // lower-mode SYRK next to the panel stream is ~4 % faster in plain order.  Option gemm_tile: 0 = this default,
    // 1 = never, 2 = both modes, 3 = lower mode only (A/B probes).
    bool use_super = g.K <= 2048 && g.nsuper >= 128;
    if (ctx->gemm_tile == 0 && d.lower) use_super = false;
    if (ctx->gemm_tile == 1) use_super = false;
    if (ctx->gemm_tile == 3 && !d.lower) use_super = false;
    if (use_super)
        ntiles = g.per_xcd * 8 * 64;
    else
        g.nsuper = 0;
    if (ntiles > 0x7fffffffLL) return set_err(ctx, FR_INVALID_ARGUMENT, "GEMM grid too large");
    double bytes = 8.0 * ((double)d.M * g.K + (double)d.N * g.K + (d.lower ? 1.0 : 2.0) * (double)d.M * d.N);
    if (d.own_world > 1) {
        // multi-GPU ownership filter: this rank computes only the tile columns it owns (same test as the kernel), so
        // the profile must credit only those -- otherwise every rank's TFLOP/s reads ~world times too high
        double own_flops = 0.0, own_c = 0.0;
        for (int64_t tn = 0; tn < g.tiles_n; ++tn) {
            const int64_t n0 = tn * BN;
            if ((int)(((d.own_col0 + n0) / g.own_nb) % d.own_world) != d.own_rank) continue;
            const double cols = (double)((n0 + BN <= d.N) ? BN : d.N - n0);
            const double rows = d.lower ? (double)(d.M - n0) : (double)d.M;  // lower mode: rows from the diagonal tile down
            own_flops += 2.0 * rows * cols * (double)g.K;
            own_c += rows * cols;
        }
        flops = own_flops;
        bytes = 8.0 * ((double)d.M * g.K + (double)d.N * g.K + 2.0 * own_c);
    }
    const double nbatch = d.batch > 1 ? (double)d.batch : 1.0;
    ProfScope ps(ctx, d.prof_cls, flops * nbatch, bytes * nbatch);
    g.batch_a = d.batch_a;
    g.batch_b = d.batch_b;
    g.batch_c = d.batch_c;
    g.batch_d = d.batch_d;
    if (d.batch > 1 && d.lower) return set_err(ctx, FR_INVALID_ARGUMENT, "batched GEMM is full-mode only");
    dim3 grid((unsigned)ntiles, (unsigned)(d.batch > 1 ? d.batch : 1)), block(256);
    if (d.lower && !d.a_kmajor && !d.b_kmajor)
        hipLaunchKernelGGL(syrk_lower_f64_kernel, grid, block, 0, ctx->ls, g);
    else if (!d.a_kmajor && !d.b_kmajor)
        hipLaunchKernelGGL((gemm_f64_kernel<false, false>), grid, block, 0, ctx->ls, g);
    else if (!d.a_kmajor && d.b_kmajor)
        hipLaunchKernelGGL((gemm_f64_kernel<false, true>), grid, block, 0, ctx->ls, g);
    else if (d.a_kmajor && d.b_kmajor)
        hipLaunchKernelGGL((gemm_f64_kernel<true, true>), grid, block, 0, ctx->ls, g);
    else
        hipLaunchKernelGGL((gemm_f64_kernel<true, false>), grid, block, 0, ctx->ls, g);
    FR_HIP(ctx, hipGetLastError());
    return FR_OK;
}

}  // namespace fr
