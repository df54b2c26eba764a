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
#include "gemm_tile.hpp"
#include <vector>

namespace fr {

template <bool A_KMAJ, bool B_KMAJ, int BMT = BM, bool MIRROR = false>
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
        const int sh = 64 >> g.sw_log2;
        if (g.lower) {
            // lower mode: the super-tiles of the lower triangle, row by row (8 x 8 tiles each; the ones on the diagonal hold 36)
            int64_t R = (int64_t)((sqrt(8.0 * (double)sidx + 1.0) - 1.0) * 0.5);
            while (R * (R + 1) / 2 > sidx) --R;
            while ((R + 1) * (R + 2) / 2 <= sidx) ++R;
            const int64_t C = sidx - R * (R + 1) / 2;
            tm = R * 8 + (within & 7);
            tn = C * 8 + (within >> 3);
            if (tm >= g.tiles_m || tn > tm) return;
        } else {
            tm = (sidx % g.super_m) * sh + (within & (sh - 1));
            tn = (sidx / g.super_m) * (1 << g.sw_log2) + (within >> (6 - g.sw_log2));
            if (tm >= g.tiles_m || tn >= g.tiles_n) return;
        }
    } else {
        // plain order: each XCD gets a contiguous run of the tile list
        int64_t tlin;
        if (g.place == 2) {
            // launch of the panel stream: workgroup b of a stream's launch is dealt to XCD (X + b) % 8, X being where its
            // single-workgroup launches (the diagonal-block kernel) run (scripts/xcc_single.hip) -- the workgroups with
            // b % 8 < nres land on the XCDs the trailing update keeps off.  The others retire at once.
            const int64_t x = b & 7;
            if (x >= g.nres) return;
            tlin = (b >> 3) * g.nres + x;
            if (tlin >= g.ntiles) return;
        } else if (g.place == 1) {
            tlin = claim_item(g.place, g.nres, g.epoch, g.xcc_word, g.claim, g.max_exit, g.ntiles);  // gemm_tile.hpp
            if (tlin < 0) return;
        } else {
            const int64_t nblk = gridDim.x;
            const int64_t q = nblk >> 3, r8 = nblk & 7;
            tlin = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (b >> 3);
        }
        if (g.lower) {
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
    const int64_t n0 = tn * BN;
    if (g.own_world > 1 && (int)(((g.own_col0 + n0) / g.own_nb) % g.own_world) != g.own_rank) return;
    // (MIRROR is a template parameter with kernel symbols of its own: as a run-time loop around the ONE call site of the tile
    // function it cost the trailing-update kernel 17 registers and 228 scratch instructions -- found as doubled WRITE_SIZE)
    constexpr int reps = MIRROR ? 2 : 1;
#pragma nounroll
    for (int rep = 0; rep < reps; ++rep) {
        const int64_t m0 = ((MIRROR && rep == 1) ? g.mirror_tiles - 1 - tm : tm) * BMT;
        GemmArgs gt = g;  // (ONE call site of the tile function: a second inlined copy would double the kernel)
        // (k0: where this launch's -- this batch member's -- part of the contraction starts in the whole one)
        const int64_t k0 = g.kslice * (int64_t)blockIdx.y;
        if (g.kslice > 0 && k0 + g.K > g.k_total) gt.K = g.k_total > k0 ? g.k_total - k0 : 0;  // the last, shorter slice
        if (g.tri) {
            // triangular operand(s): skip the part of the contraction that only multiplies structural zeros
            int64_t kbeg = k0, kend = k0 + gt.K;
            if ((g.tri & 1) && m0 > kbeg) kbeg = m0;
            if ((g.tri & 2) && n0 > kbeg) kbeg = n0;
            if ((g.tri & 4) && m0 + BMT < kend) kend = m0 + BMT;
            if (kbeg > kend) kbeg = kend;  // (an empty slice: the tile is written as beta * Cin)
            gt.A += (kbeg - k0) * (A_KMAJ ? 1 : g.lda);
            gt.B += (kbeg - k0) * (B_KMAJ ? 1 : g.ldb);
            gt.K = kend - kbeg;
        }
        if constexpr (BMT == BM)
            gemm_f64_tile<A_KMAJ, B_KMAJ>(gt, lds, m0, n0);
        else
            gemm_f64_tile_m32<A_KMAJ, B_KMAJ>(gt, lds, m0, n0);
        if (MIRROR && rep == 0) __syncthreads();  // (the LDS stages are reused by the mirrored tile)
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

// 32-row tiles: products with too few 128 x 128 tiles to fill the chip (gemm_tile.hpp: gemm_f64_tile_m32)
template <bool A_KMAJ, bool B_KMAJ>
__global__ __launch_bounds__(256, 2) void gemm_f64_m32_kernel(const GemmArgs g0)
{
    __shared__ double lds[2 * (TILE_A_S + TILE_ELEMS)];
    GemmArgs g = g0;
    const int64_t bz = blockIdx.y;
    g.A += bz * g.batch_a;
    g.B += bz * g.batch_b;
    g.Cin += bz * g.batch_c;
    g.D += bz * g.batch_d;
    gemm_f64_body<A_KMAJ, B_KMAJ, BMS>(g, lds);
}

// 32-row tiles in mirrored pairs (GemmArgs::mirror_tiles): the big solve leaves with their triangular left operand
template <bool A_KMAJ, bool B_KMAJ>
__global__ __launch_bounds__(256, 2) void gemm_f64_m32_mirror_kernel(const GemmArgs g0)
{
    __shared__ double lds[2 * (TILE_A_S + TILE_ELEMS)];
    GemmArgs g = g0;
    const int64_t bz = blockIdx.y;
    g.A += bz * g.batch_a;
    g.B += bz * g.batch_b;
    g.Cin += bz * g.batch_c;
    g.D += bz * g.batch_d;
    gemm_f64_body<A_KMAJ, B_KMAJ, BMS, true>(g, lds);
}

__global__ __launch_bounds__(256, 2) void syrk_lower_f64_kernel(const GemmArgs g)
{
    __shared__ double lds[4 * TILE_ELEMS];
    gemm_f64_body<false, false>(g, lds);
}

// ---- resident workgroups that claim tiles (GemmArgs::place == 3) ---------------------------------------------------------------
// While CUs are set aside for the panel chain (fr_ctx::cu_reserve), the main stream's products run as ONE round of resident
// workgroups: a workgroup that finds itself on a reserved CU -- one of the ncu_res lowest-ranked CUs of its shader engine, on
// every XCD -- leaves at once, everybody else takes tiles from the list until it is empty.  Nothing of the launch is dispatched
// after its first microseconds, so the reserved CUs stay empty for as long as it runs: the diagonal-block kernel and the panel
// stream's products find them without any placement logic of their own, and no launch of the panel stream carries idle
// workgroups (which have to wait for a slot on the busy XCDs before they can exit: scripts/dispatch_probe.hip).  Every tile is
// computed whole by one workgroup with the arithmetic of the one-tile-per-workgroup launch: the same bits.
// Own kernel symbols: the loop around the tile function costs registers the one-tile kernels must not pay (see MIRROR above).
template <bool A_KMAJ, bool B_KMAJ>
__device__ __forceinline__ void gemm_f64_persist_body(const GemmArgs& g, double* lds)
{
    __shared__ long long item;
    if (threadIdx.x == 0) {
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));  // cu_id[11:8] sh_id[12] se_id[15:13]
        const unsigned rank = g.cu_rank[(((xcc & 7u) * 8u + ((hwid >> 13) & 7u)) << 4) + ((hwid >> 8) & 15u)];
        bool leave = false;
        if (rank < (unsigned)g.ncu_res) leave = __hip_atomic_fetch_add(g.claim + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < g.max_exit;
        item = leave ? -2 : 0;
    }
    __syncthreads();
    if (item == -2) return;
    for (;;) {
        __syncthreads();  // (everybody has read `item`, the tile before is out of the LDS stages)
        if (threadIdx.x == 0) {
            const unsigned i = __hip_atomic_fetch_add(g.claim, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            item = (int64_t)i < g.ntiles ? (long long)i : -1;
        }
        __syncthreads();
        const long long tlin = item;
        if (tlin < 0) return;
        int64_t tm, tn;
        if (g.lower) {
            int64_t row = (int64_t)((sqrt(8.0 * (double)tlin + 1.0) - 1.0) * 0.5);
            while (row * (row + 1) / 2 > tlin) --row;
            while ((row + 1) * (row + 2) / 2 <= tlin) ++row;
            tm = row;
            tn = tlin - row * (row + 1) / 2;
        } else {
            tm = tlin % g.tiles_m;
            tn = tlin / g.tiles_m;
        }
        const int64_t n0 = tn * BN;
        if (g.own_world > 1 && (int)(((g.own_col0 + n0) / g.own_nb) % g.own_world) != g.own_rank) continue;
        // (the leading dimensions and base pointers pass through an empty asm in every round: what the tile function derives
        // from them per lane would otherwise be hoisted out of the loop and kept alive across it -- 255 registers and 260 B of
        // scratch per lane where the one-tile kernel needs 237 and none)
        GemmArgs gt = g;
        asm volatile("" : "+s"(gt.lda), "+s"(gt.ldb), "+s"(gt.ldd), "+s"(gt.ldcin));
        asm volatile("" : "+s"(gt.A), "+s"(gt.B), "+s"(gt.D), "+s"(gt.Cin));
        gemm_f64_tile<A_KMAJ, B_KMAJ>(gt, lds, tm * BM, n0);
    }
}

__global__ __launch_bounds__(256, 2) void syrk_lower_persist_f64_kernel(const GemmArgs g)
{
    __shared__ double lds[4 * TILE_ELEMS];
    gemm_f64_persist_body<false, false>(g, lds);
}

__global__ __launch_bounds__(256, 2) void gemm_f64_persist_kernel(const GemmArgs g)
{
    __shared__ double lds[4 * TILE_ELEMS];
    gemm_f64_persist_body<false, false>(g, lds);
}

// which (shader engine, CU id) pairs exist on each XCD: one workgroup per slot of the chip, each notes where it ran
__global__ __launch_bounds__(256) void cu_probe_kernel(unsigned* seen)
{
    if (threadIdx.x == 0) {
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        atomicAdd(seen + ((((xcc & 7u) * 8u + ((hwid >> 13) & 7u)) << 4) + ((hwid >> 8) & 15u)), 1u);
    }
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < 3000) __builtin_amdgcn_s_sleep(8);  // 30 us: the launch spreads over every CU
}

// ---- rows of a panel solved against its factored diagonal block, ONE launch ------------------------------------------------
// S (rows x kb) <- S L_kk^-T, left-looking over the 128-column sub-panels:  S_s <- (S_s - S_{<s} L[s, <s]^T) W_s^T.  The rows are
// independent, so a workgroup takes 32 of them through all 2 nblk - 1 tile products by itself -- no hand-off between
// workgroups, one launch instead of 2 nblk - 1 (each ~18 us of launch-bound time on the few tiles of a diagonal row tile or
// of a rank's slice: the sharded factorisation's R1 and bulk solves, 124 -> ~45 us for 512 rows).  Every element sees the
// arithmetic of the per-sub-panel launches (same tile function, same contraction order).
struct RowsSolveArgs {
    double* S;
    int64_t lds_, rows;
    const double* L;     // the factored kb x kb diagonal block (lower), leading dimension ldl
    int64_t ldl, kb;
    const double* dinv;  // its 128-block inverses, block s at dinv + s * 128 * 128 (ld 128)
};

__global__ __launch_bounds__(256, 2) void rows_solve_kernel(const RowsSolveArgs a)
{
    __shared__ double lds[2 * (TILE_A_S + TILE_ELEMS)];
    const int64_t m0 = (int64_t)blockIdx.x * BMS;
    const int64_t nblk = (a.kb + 127) / 128;
    GemmArgs g;
    g.M = a.rows;
    g.own_world = 1; g.own_rank = 0; g.own_nb = 1; g.own_col0 = 0;
    g.mirror_tiles = 0; g.kslice = 0; g.k_total = 0;
    g.lower = 0; g.tri = 0; g.place = 0; g.nres = 0; g.epoch = 0; g.ntiles = 0; g.xcc_word = nullptr; g.claim = nullptr; g.max_exit = 0;
    g.tiles_m = 1; g.tiles_n = 1; g.sw_log2 = 0; g.super_m = 1; g.nsuper = 0; g.per_xcd = 0;
    g.batch_a = g.batch_b = g.batch_c = g.batch_d = 0;
    // items: for every 128-column sub-panel s its update by the sub-panels in front of it (s > 0), then its solve -- ONE call site
    // of the tile function (a second inlined copy doubles the kernel and its register pressure)
#pragma nounroll
    for (int64_t it = 1; it < 2 * nblk; ++it) {
        const int64_t s = it >> 1;
        const bool update = (it & 1) == 0;  // it = 2 s: update of sub-panel s;  it = 2 s + 1: its solve
        const int64_t c0 = s * 128, cs = (a.kb - c0) < 128 ? (a.kb - c0) : 128;
        double* Ss = a.S + c0 * a.lds_;
        g.N = cs;
        g.Cin = Ss; g.ldcin = a.lds_; g.D = Ss; g.ldd = a.lds_;
        if (update) {
            // S_s -= S_{<s} L[s, <s]^T
            g.K = c0;
            g.A = a.S; g.lda = a.lds_;
            g.B = a.L + c0; g.ldb = a.ldl;
            g.alpha = -1.0; g.beta = 1.0;
        } else {
            // S_s <- S_s W_s^T  (in place: the tile's contraction has read its 32 rows before the epilogue writes them)
            g.K = cs;
            g.A = Ss; g.lda = a.lds_;
            g.B = a.dinv + s * (128 * 128); g.ldb = 128;
            g.alpha = 1.0; g.beta = 0.0;
        }
        gemm_f64_tile_m32<false, false>(g, lds, m0, 0);
        __syncthreads();  // (this workgroup's stores, drained, before its own loads of the same rows)
    }
}

// The same sweep with the workgroup's rows RESIDENT in LDS (round 5): 16 rows x kb <= 512 columns (64 KiB) are loaded once, every
// product of the sweep reads its left operand from there and its right operand -- a 128 x K block of the factor or an inverse
// block, shared by all workgroups and L2-resident -- as matrix-core fragments straight from memory, eight k-groups ahead; the
// strip goes back to memory once, at the end.  The generic kernel above pays a store -> drain -> reload round trip through L2
// between any two of its 2 nblk - 1 dependent products (~17 us each); here a product is its matrix-core time (64 instructions per
// wave and 128 of contraction: 1.7 us) plus one load latency.  Same arithmetic up to the order of summation (even and odd k-groups go to two accumulators: agreement
// with the generic kernel to round-off, 4e-16 against numpy in scripts/rows_solve_probe.py).  kb a multiple of 128, at most 512.
constexpr int RS16_STRIDE = 516;  // doubles per row of the strip (516 mod 32 = 4: rows 0..7 x four k's hit 32 distinct 8-byte banks)

template <int NW>  // waves per workgroup: 4 (two 16-column tiles of a sub-panel per wave) or 8 (one)
__global__ __launch_bounds__(64 * NW) void rows_solve16_kernel(const RowsSolveArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double strip[];  // [16][RS16_STRIDE]
    constexpr int NT = 64 * NW, CPT = NT / 16, TPW = 8 / NW, CW = 16 * TPW;  // threads, columns per trip of the loaders, tiles / columns per wave
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int64_t m0 = (int64_t)blockIdx.x * 16;
    const int nblk = (int)(a.kb / 128);
    // load: 16 rows x kb columns (a column is 128 contiguous bytes)
    {
        // (several loads in flight per thread: a loop of one load and one LDS store per trip is a chain of kb / CPT memory latencies)
        const int r = t & 15;
        const bool rok = m0 + r < a.rows;
        const double* src = a.S + (m0 + r) + (int64_t)(t >> 4) * a.lds_;
        double* dst = strip + r * RS16_STRIDE + (t >> 4);
        for (int64_t cb = 0; cb < a.kb; cb += 128) {
            double v[128 / CPT];
#pragma unroll
            for (int i = 0; i < 128 / CPT; ++i) v[i] = rok ? src[(cb + CPT * i) * a.lds_] : 0.0;
#pragma unroll
            for (int i = 0; i < 128 / CPT; ++i) dst[cb + CPT * i] = v[i];
        }
    }
    __syncthreads();
    const double* srow = strip + l15 * RS16_STRIDE + lq;  // this lane's left-operand element of k-group j: srow[4 j]
    // The sweep as ONE sequence of batches of D = 32 k-groups (128 of contraction): sub-panel s has s batches of its update
    // (acc = S[:, 0 .. c0) L[c0 + n, 0 .. c0)^T, then S_s -= acc) and one batch of its solve (acc = S_s W_s^T, then S_s = acc).
    // The right-operand fragments of batch i + 1 are requested slot by slot right behind the use of batch i's -- across the
    // boundaries between products too, where the strip is written back and the waves synchronise: the sweep is bound by the
    // latency of those loads (~2 us from a remote L2 / the Infinity Cache), not by the matrix core (17 us for the whole sweep).
    constexpr int D = 32;
    const int nbatch = nblk * (nblk + 1) / 2;
    auto batch_ptr = [&](int it, int& sub, int& upd_batch) -> const double* {  // -> this lane's first fragment element of batch `it`
        int sb = 0;
        while ((sb + 1) * (sb + 2) / 2 <= it) ++sb;
        const int r = it - sb * (sb + 1) / 2;
        sub = sb;
        upd_batch = r < sb ? r : -1;  // -1: the solve
        if (r < sb) return a.L + (128 * sb + CW * wave + l15) + (int64_t)(128 * r + lq) * a.ldl;
        return a.dinv + (int64_t)sb * (128 * 128) + (CW * wave + l15) + (int64_t)lq * 128;
    };
    double rb0[D], rb1[TPW == 2 ? D : 1];
    {
        int sb, ub;
        const double* p = batch_ptr(0, sb, ub);
#pragma unroll
        for (int u = 0; u < D; ++u) {
            rb0[u] = p[(int64_t)(4 * u) * 128];
            if constexpr (TPW == 2) rb1[u] = p[16 + (int64_t)(4 * u) * 128];
        }
    }
    // (even and odd k-groups go to accumulator chains of their own, added at the end of a product: a matrix-core instruction
    // that depends on the one before it on the same accumulator cannot issue back to back)
    d4_t acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0}, acc0b = {0.0, 0.0, 0.0, 0.0}, acc1b = {0.0, 0.0, 0.0, 0.0};
    for (int it = 0; it < nbatch; ++it) {
        int sb, ub, nsb = 0, nub = 0;
        (void)batch_ptr(it, sb, ub);
        const bool more = it + 1 < nbatch;
        const double* np = more ? batch_ptr(it + 1, nsb, nub) : a.dinv;
        const int64_t nld = (more && nub >= 0) ? a.ldl : 128;
        const int c0 = 128 * sb;
        const double* left = ub >= 0 ? srow + 128 * ub : srow + c0;
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const double av = left[4 * u];
            if (u & 1) {
                acc0b = __builtin_amdgcn_mfma_f64_16x16x4f64(rb0[u], av, acc0b, 0, 0, 0);
                if constexpr (TPW == 2) acc1b = __builtin_amdgcn_mfma_f64_16x16x4f64(rb1[u], av, acc1b, 0, 0, 0);
            } else {
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(rb0[u], av, acc0, 0, 0, 0);
                if constexpr (TPW == 2) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(rb1[u], av, acc1, 0, 0, 0);
            }
            if (more) {
                rb0[u] = np[(int64_t)(4 * u) * nld];
                if constexpr (TPW == 2) rb1[u] = np[16 + (int64_t)(4 * u) * nld];
            }
        }
        const bool upd_done = ub >= 0 && ub == sb - 1, solve_done = ub < 0;
        if (upd_done || solve_done) {
            __syncthreads();  // every wave has read what it needs of sub-panel sb before anybody overwrites it
            acc0 += acc0b;
            acc1 += acc1b;
            acc0b = d4_t{0.0, 0.0, 0.0, 0.0};
            acc1b = d4_t{0.0, 0.0, 0.0, 0.0};
            double* out = strip + l15 * RS16_STRIDE + c0 + CW * wave + lq;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (upd_done) {
                    out[4 * r] = out[4 * r] - acc0[r];
                    if constexpr (TPW == 2) out[16 + 4 * r] = out[16 + 4 * r] - acc1[r];
                } else {
                    out[4 * r] = acc0[r];
                    if constexpr (TPW == 2) out[16 + 4 * r] = acc1[r];
                }
            }
            acc0 = d4_t{0.0, 0.0, 0.0, 0.0};
            acc1 = d4_t{0.0, 0.0, 0.0, 0.0};
            __syncthreads();
        }
    }
    {
        const int r = t & 15;
        if (m0 + r < a.rows) {
            double* dst = a.S + (m0 + r) + (int64_t)(t >> 4) * a.lds_;
            const double* src = strip + r * RS16_STRIDE + (t >> 4);
            for (int64_t c = 0; c < a.kb; c += CPT) dst[c * a.lds_] = src[c];
        }
    }
}

int launch_rows_solve(fr_ctx* ctx, double* S, int64_t lds_, int64_t rows, const double* L, int64_t ldl, int64_t kb, const double* dinv)
{
    if (rows <= 0 || kb <= 0) return FR_OK;
    RowsSolveArgs a;
    a.S = S; a.lds_ = lds_; a.rows = rows; a.L = L; a.ldl = ldl; a.kb = kb; a.dinv = dinv;
    ProfScope ps(ctx, FR_PROF_GEMM_PANEL, (double)rows * (double)kb * (double)kb, 8.0 * 2.0 * (double)rows * (double)kb);
    static const int rs16 = getenv("FRIEDRICH_AMD_ROWS_SOLVE16") ? atoi(getenv("FRIEDRICH_AMD_ROWS_SOLVE16")) : 1;
    if (rs16 && kb % 128 == 0 && kb <= 512) {
        const size_t lds_bytes = sizeof(double) * 16 * RS16_STRIDE;
        if (!ctx->rs16_lds_set) {  // (66 KB: above the default dynamic-LDS limit)
            FR_TRY(set_dyn_lds(ctx, reinterpret_cast<const void*>(rows_solve16_kernel<4>), (int)lds_bytes));
            FR_TRY(set_dyn_lds(ctx, reinterpret_cast<const void*>(rows_solve16_kernel<8>), (int)lds_bytes));
            ctx->rs16_lds_set = true;
        }
        if (rs16 == 4)  // (A/B: four waves with two tiles each -- 45.5 us against 42.8 for eight waves with one)
            hipLaunchKernelGGL(rows_solve16_kernel<4>, dim3((unsigned)((rows + 15) / 16)), dim3(256), lds_bytes, ctx->ls, a);
        else
            hipLaunchKernelGGL(rows_solve16_kernel<8>, dim3((unsigned)((rows + 15) / 16)), dim3(512), lds_bytes, ctx->ls, a);
        FR_HIP(ctx, hipGetLastError());
        return FR_OK;
    }
    hipLaunchKernelGGL(rows_solve_kernel, dim3((unsigned)((rows + BMS - 1) / BMS)), dim3(256), 0, ctx->ls, a);
    FR_HIP(ctx, hipGetLastError());
    return FR_OK;
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

__global__ void release_xcds_kernel(unsigned* word, unsigned epoch)
{
    if (threadIdx.x == 0) __hip_atomic_store(word, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int launch_release_xcds(fr_ctx* ctx, unsigned epoch)
{
    hipLaunchKernelGGL(release_xcds_kernel, dim3(1), dim3(64), 0, ctx->ls, ctx->xcc_word + 1, epoch);
    FR_HIP(ctx, hipGetLastError());
    return FR_OK;
}

// The rank table of the CU-level reservation: [xcc][se][cu_id] -> position of that CU among the CUs of its shader engine that
// exist on this chip (harvested CUs leave holes in the ids, different ones per engine: scripts/hwid_probe.hip), 255 where there
// is none.  Built once per context from a launch that puts a workgroup on every CU; usable only if every one of 8 x 4 engines
// shows at least four CUs (otherwise the CU-level reservation stays off and the XCD-level one is used).
static int ensure_cu_table(fr_ctx* ctx)
{
    if (ctx->cu_rank_state != 0) return ctx->cu_rank_state;
    ctx->cu_rank_state = -1;
    unsigned* seen = nullptr;
    if (hipMalloc((void**)&seen, sizeof(unsigned) * 1024) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    std::vector<unsigned> h(1024, 0u);
    std::vector<unsigned char> rank(1024, 255);
    bool ok = hipMemsetAsync(seen, 0, sizeof(unsigned) * 1024, ctx->stream) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(cu_probe_kernel, dim3(4096), dim3(256), 0, ctx->stream, seen);
        ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(h.data(), seen, sizeof(unsigned) * 1024, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
             hipStreamSynchronize(ctx->stream) == hipSuccess;
    }
    (void)hipFree(seen);
    if (!ok) {
        (void)hipGetLastError();
        return -1;
    }
    int engines = 0;
    for (int e = 0; e < 64; ++e) {
        int r = 0;
        for (int c = 0; c < 16; ++c)
            if (h[(size_t)e * 16 + c]) rank[(size_t)e * 16 + c] = (unsigned char)r++;
        if (r >= 4) ++engines;
        else if (r > 0) return -1;  // an engine with fewer CUs than may be set aside
    }
    if (engines != 32) return -1;  // not the 8 XCDs x 4 engines this scheme is written for
    if (hipMalloc((void**)&ctx->cu_rank, 1024) != hipSuccess || hipMemcpy(ctx->cu_rank, rank.data(), 1024, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    ctx->cu_rank_state = 1;
    return 1;
}

bool cu_table_ready(fr_ctx* ctx) { return ctx->cu_reserve != 0 && ensure_cu_table(ctx) == 1; }
bool cu_reserve_active(fr_ctx* ctx) { return ctx->reserve_by_cu_now && ctx->cu_reserve != 0 && ctx->cu_rank_state == 1; }

// Host side of claim_item (gemm_tile.hpp): a counter pair from the factorisation's ring for one launch that keeps off the
// panel stream's XCDs.  Returns the placement to use (0: launch plainly) and the grid size.
static int claim_setup(fr_ctx* ctx, int64_t items, const unsigned** xcc_word, unsigned** claim, unsigned* max_exit, int64_t* grid)
{
    *xcc_word = nullptr;
    *claim = nullptr;
    *max_exit = 0;
    *grid = items;
    if (!ctx->reserve_now || ctx->ls == ctx->stream2 || !ctx->claim_ring || ctx->claim_next >= kClaimSlots) return 0;
    *xcc_word = ctx->xcc_word;
    *claim = ctx->claim_ring + 2 * ctx->claim_next++;
    const int64_t on = 8 - ctx->reserve_now;  // XCDs whose workgroups take items
    if (on <= 0) return 0;
    *max_exit = (unsigned)(items * (8 - on) / on + 16);
    *grid = items + *max_exit;
    return 1;
}

// Few result tiles and a deep contraction (a 512-row block against 8192 columns: add_rows, narrow predicts; the lower levels
// of the recursive wide solves): one tile's K-loop is then the whole run time while most CUs idle.  The contraction is cut into slices computed as one batched
// launch into a workspace, and a second small kernel adds them up (fixed order).  Main stream only (the workspace pool
// relies on stream order), single GPU ownership only.
// The rule: products of at most kSplitkTiles result tiles and a contraction of at least kSplitkMinK are cut into
// ~kSplitkTarget / tiles slices of at least kSplitkSlice.  Round 1 used 192 / 2048 / 384; measured in round 2 with
// 256 / 512 / 512 (forward solve of 512 / 1024 / 2048 / 4096 columns at n = 32768: 18.7 / 26.6 / 41.6 / 70.6 ->
// 14.0 / 22.1 / 37.6 / 68.3 ms, 1024 columns at n = 8192: 3.45 -> 2.48 ms, fits -1 %; slices of 128: another -8 ... -13 %)
constexpr int64_t kSplitkTiles = 256, kSplitkMinK = 512, kSplitkTarget = 512, kSplitkSlice = 128;

int launch_gemm(fr_ctx* ctx, const GemmDesc& d)
{
    if (d.M <= 0 || d.N <= 0) return FR_OK;
    const int64_t tiles = ((d.M + BM - 1) / BM) * ((d.N + BN - 1) / BN);
    // (the trailing updates of a factorisation -- profile class SYRK -- keep round 1's rule: their launches stay
    // syrk_lower_f64_kernel launches, which is what the profile class and the rocprofv3 summaries count)
    if (d.batch <= 1 && d.own_world <= 1 && (!d.tri || d.tri_splitk) && !d.dynamic && ctx->ls == ctx->stream && ctx->splitk != 0 &&
        tiles <= (d.prof_cls == FR_PROF_SYRK ? 192 : kSplitkTiles) && d.K >= (d.prof_cls == FR_PROF_SYRK ? 2048 : kSplitkMinK) &&
        d.M <= 65535 * 256) {
        // (triangular operands: about half of the slices only meet structural zeros and retire at once; twice the slices to make up
        // for them was measured and is slower -- 166 -> 200 us for a 2048 x 1024 x 2048 leaf: more partial sums to write and add)
        int64_t S = (kSplitkTarget + tiles - 1) / tiles;
        if (S > d.K / kSplitkSlice) S = d.K / kSplitkSlice;
        if (S > 32) S = 32;
        // slices of a whole number of K-steps; the last one takes what is left (shorter, possibly ragged: a contraction of
        // 32769 -- one row appended to a factor of 32768 -- used to find no divisor and ran as ONE 2048-step tile, 3.5 ms)
        int64_t ks = S > 1 ? round_up((d.K + S - 1) / S, BK) : d.K;
        if (S > 1) S = (d.K + ks - 1) / ks;
        if (S > 1) {
            WsGuard w(ctx);
            double* part = w.get(sizeof(double) * (size_t)S * (size_t)d.M * (size_t)d.N);
            if (!part) return FR_OUT_OF_MEMORY;
            GemmDesc p = d;
            p.K = ks;
            p.lower = false;
            p.Cin = part; p.ldcin = d.M; p.D = part; p.ldd = d.M;
            p.beta = 0.0;
            p.batch = S;
            p.batch_a = d.a_kmajor ? ks : ks * d.lda;
            p.batch_b = d.b_kmajor ? ks : ks * d.ldb;
            p.batch_c = p.batch_d = d.M * d.N;
            p.kslice = ks;  // (slices that only meet structural zeros of a triangular operand write zeros and retire)
            p.k_total = d.K;
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
    g.lower = d.lower ? 1 : 0;
    g.own_world = d.own_world;
    g.own_rank = d.own_rank;
    g.own_nb = d.own_nb > 0 ? d.own_nb : 1;
    g.own_col0 = d.own_col0;
    g.tiles_m = (d.M + BM - 1) / BM;
    g.tiles_n = (d.N + BN - 1) / BN;
    // Few tiles: 32-row tiles instead (four times the workgroups, a quarter of the matrix-core time each).  Measured on the
    // panel chain, whose products these are (in-process A/B, fits at N = 4096 / 8192 / 16384: 3.92 / 9.11 / 33.7 -> 3.41 / 8.52 /
    // 32.8 ms with the threshold at 64 tiles; at 160 or 320 tiles N = 16384 loses the gain again; predict_variance of 1024
    // points at N = 4096 / 8192: 0.86 / 2.15 -> 0.77 / 1.95 ms); the lower-mode and triangular-operand launches keep the large tile.
    // (a launch of the panel stream under the XCD reservation only has 32 CUs per reserved XCD: there four times the
    // workgroups pay while the large tiles would not fill those CUs either -- with 1 XCD and small tiles for up to 64 large
    // ones the fit at N = 8192 went from 9.1 to 10.2 ms)
    int64_t small_max = ctx->small_tiles;
    {
        // (by CUs the panel stream's workgroups go to the reserved CUs of ALL XCDs, two or three to a CU: 32-row tiles pay for up to
        // 64 large tiles per reserved unit -- N = 8192 6.56 -> 6.43 ms, in-process A/B of 32 / 64 / 128)
        static const int cu_cap = getenv("FRIEDRICH_AMD_CU_SMALL_CAP") ? atoi(getenv("FRIEDRICH_AMD_CU_SMALL_CAP")) : 64;
        const int64_t cap = (cu_reserve_active(ctx) ? (int64_t)cu_cap : 32) * ctx->reserve_now;
        if (ctx->reserve_now && d.batch <= 1 && !d.whole_chip && ctx->ls == ctx->stream2 && small_max > cap) small_max = cap;
    }
    const bool small = (small_max > 0 && !d.lower && !d.tri && d.M > BMS && d.D != d.B /* in place over op(B) needs ONE tile row */ &&
                        g.tiles_m * g.tiles_n * (d.batch > 1 ? d.batch : 1) <= small_max) ||
                       (d.force_small && !d.lower && d.M > BMS && d.D != d.B);
    if (small) g.tiles_m = (d.M + BMS - 1) / BMS;
    g.kslice = d.kslice;
    g.k_total = d.k_total;
    g.mirror_tiles = 0;
    if (d.mirror && small && d.b_kmajor && !d.lower && d.batch <= 1 && g.tiles_m >= 2 && g.tiles_m % 2 == 0) {
        g.mirror_tiles = g.tiles_m;
        g.tiles_m /= 2;
    }
    double flops;
    int64_t ntiles;
    g.sw_log2 = 3;
    g.super_m = 1;
    g.nsuper = 0;
    if (d.lower) {
        if (d.M != d.N) return set_err(ctx, FR_INVALID_ARGUMENT, "lower-mode GEMM needs a square result");
        ntiles = g.tiles_m * (g.tiles_m + 1) / 2;
        flops = (double)d.M * (double)(d.M + 1) * (double)g.K;  // 2 * M(M+1)/2 * K
        g.sw_log2 = 3;
        g.super_m = (g.tiles_m + 7) / 8;
        g.nsuper = g.super_m * (g.super_m + 1) / 2;
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
    // measured inside the factorisation (round 1, in-process A/B): super-tiles pay for large shallow products only
    // lower mode (the trailing update): 8 x 8 super-tiles of the lower triangle while nothing is reserved -- same time as the plain
    // order inside the factorisation (N = 32768: 194.5 vs 194.6 ms, in-process A/B with FRIEDRICH_AMD_SYRK_SUPER=0 / 1), 22 % less
    // L2-side fetch traffic per launch (8.15 -> 6.36 GB: rocprofv3 FETCH_SIZE; the workgroups of a super-tile drift apart over
    // a 1024-deep contraction, so the hit rate stays well below the 8 x 8 ideal)
    static const int syrk_super = getenv("FRIEDRICH_AMD_SYRK_SUPER") ? atoi(getenv("FRIEDRICH_AMD_SYRK_SUPER")) : 1;
    bool use_super = (d.lower ? (syrk_super != 0 && d.own_world <= 1) : !small) && g.K <= 2048 && g.nsuper >= 128;
    g.place = 0;
    g.ntiles = ntiles;
    g.xcc_word = nullptr;
    g.claim = nullptr;
    g.max_exit = 0;
    g.nres = ctx->reserve_now;
    g.epoch = ctx->panel_epoch;
    g.tri = d.tri;
    g.cu_rank = nullptr;
    g.ncu_res = 0;
    int64_t persist_grid = 0;
    const bool by_cu = ctx->reserve_now && cu_reserve_active(ctx);
    if (by_cu) {
        // CU-level reservation: the panel stream's launches go wherever there is room -- the reserved CUs, which everybody else
        // vacates; the other streams' products on 128 x 128 tiles run as resident workgroups that claim their tiles
        if (d.batch <= 1 && !d.whole_chip && ctx->ls != ctx->stream2 && !small && !d.a_kmajor && !d.b_kmajor && !d.tri && g.mirror_tiles == 0 &&
            g.kslice == 0 && ctx->claim_ring && ctx->claim_next < kClaimSlots) {
            g.place = 3;
            g.claim = ctx->claim_ring + 2 * ctx->claim_next++;
            g.cu_rank = ctx->cu_rank;
            g.ncu_res = ctx->reserve_now;
            // As many workgroups as are needed for `ntiles` of them to end up on CUs that are not set aside (the dispatcher spreads
            // a launch of up to 256 workgroups one to a CU: a product with fewer tiles than CUs keeps a CU's matrix cores to each of
            // its tiles), at most one per slot of the chip (two per CU).  A shortfall only means that some workgroups take a second tile.
            const int64_t rcus = (int64_t)ctx->reserve_now * 32;  // CUs set aside
            int64_t G = (ntiles * 256 + (256 - rcus) - 1) / (256 - rcus) + 8;
            if (G > 512) G = 512;
            int64_t me = G * rcus / 256 + G * rcus / 512 + 16;  // 1.5 x the expected share of the reserved CUs ...
            const int64_t want = ntiles < G ? ntiles : G;
            const int64_t keep = want / 2 > 1 ? want / 2 : 1;  // ... but this many workgroups stay whatever the dispatcher does
            if (me > G - keep) me = G - keep;
            g.max_exit = (unsigned)(me > 0 ? me : 0);
            persist_grid = G;
            use_super = false;
        }
    } else if (ctx->reserve_now && d.batch <= 1 && !d.whole_chip) {
        if (ctx->ls == ctx->stream2) {
            if (!d.lower) g.place = 2;
        } else {
            int64_t grid = 0;
            g.place = claim_setup(ctx, ntiles, &g.xcc_word, &g.claim, &g.max_exit, &grid);
        }
        if (g.place) use_super = false;
    }
    if (use_super)
        ntiles = g.per_xcd * 8 * 64;
    else
        g.nsuper = 0;
    if (d.dynamic && g.place == 0 && d.batch <= 1 && d.own_world <= 1 && ctx->dyn_ring && ctx->xcc_word) {
        // claimed order with nothing reserved (nres = 0): every workgroup takes the next tile of the list
        unsigned* ctr = ctx->dyn_ring + 2 * (ctx->dyn_next++ & 255);
        FR_HIP(ctx, hipMemsetAsync(ctr, 0, 2 * sizeof(unsigned), ctx->ls));
        g.place = 1;
        g.nres = 0;
        g.xcc_word = ctx->xcc_word;
        g.claim = ctr;
        g.max_exit = 0;
        g.nsuper = 0;
    }
    if (g.place == 2) ntiles = 8 * ((g.ntiles + g.nres - 1) / g.nres);
    if (g.place == 1) ntiles = g.ntiles + g.max_exit;
    if (g.place == 3) ntiles = persist_grid;
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
    if (d.tri) flops *= (d.tri == 1 && d.lower) ? (2.0 / 3.0) : 0.5;  // average length of the restricted contraction
    const double nbatch = d.batch > 1 ? (double)d.batch : 1.0;
    ProfScope ps(ctx, (g.place == 3 && d.prof_cls == FR_PROF_SYRK) ? (int)FR_PROF_SYRK_CHAIN : d.prof_cls, flops * nbatch, bytes * nbatch);
    g.batch_a = d.batch_a;
    g.batch_b = d.batch_b;
    g.batch_c = d.batch_c;
    g.batch_d = d.batch_d;
    if (d.batch > 1 && d.lower) return set_err(ctx, FR_INVALID_ARGUMENT, "batched GEMM is full-mode only");
    dim3 grid((unsigned)ntiles, (unsigned)(d.batch > 1 ? d.batch : 1)), block(256);
    if (g.mirror_tiles > 0 && !d.a_kmajor)
        hipLaunchKernelGGL((gemm_f64_m32_mirror_kernel<false, true>), grid, block, 0, ctx->ls, g);
    else if (g.mirror_tiles > 0)
        hipLaunchKernelGGL((gemm_f64_m32_mirror_kernel<true, true>), grid, block, 0, ctx->ls, g);
    else if (small && !d.a_kmajor && !d.b_kmajor)
        hipLaunchKernelGGL((gemm_f64_m32_kernel<false, false>), grid, block, 0, ctx->ls, g);
    else if (small && !d.a_kmajor && d.b_kmajor)
        hipLaunchKernelGGL((gemm_f64_m32_kernel<false, true>), grid, block, 0, ctx->ls, g);
    else if (small && d.a_kmajor && d.b_kmajor)
        hipLaunchKernelGGL((gemm_f64_m32_kernel<true, true>), grid, block, 0, ctx->ls, g);
    else if (small)
        hipLaunchKernelGGL((gemm_f64_m32_kernel<true, false>), grid, block, 0, ctx->ls, g);
    else if (g.place == 3 && d.lower)
        hipLaunchKernelGGL(syrk_lower_persist_f64_kernel, grid, block, 0, ctx->ls, g);
    else if (g.place == 3)
        hipLaunchKernelGGL(gemm_f64_persist_kernel, grid, block, 0, ctx->ls, g);
    else if (d.lower && !d.a_kmajor && !d.b_kmajor)
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

// The one-launch panel-row solve by itself (friedrich_amd.h: a diagnostic entry like fr_gemm; scripts/rows_solve_probe.py,
// scripts/dist_model.py, tests/test_gpu_dist.py).  S (rows x kb, device) <- S L^-T against a factored kb x kb block L and its 128-block inverses.
extern "C" int fr_panel_rows_solve(fr_ctx* ctx, double* S, int64_t lds_, int64_t rows, const double* L, int64_t ldl, int64_t kb, const double* dinv)
{
    if (!ctx) return FR_INVALID_ARGUMENT;
    FR_LOCK(ctx);
    FR_HIP(ctx, hipSetDevice(ctx->device));
    return fr::launch_rows_solve(ctx, S, lds_, rows, L, ldl, kb, dinv);
}
