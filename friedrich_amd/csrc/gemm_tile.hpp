// gemm_tile.hpp -- device code of the FP64 matrix-core GEMM tile (see gemm_f64.hip for the design notes).
#pragma once

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
    // XCD reservation (gemm_f64.hip): place 1 = keep off the XCD named by *xcc_word (tiles claimed from claim[0], at most
    // max_exit workgroups retire through claim[1]); place 2 = only the workgroups with blockIdx.x % 8 == 0 work
    int place;
    int nres;  // number of reserved XCDs: the one named by *xcc_word and the nres - 1 after it
    unsigned epoch;  // the reservation holds until xcc_word[1] (last panel whose chain is finished) reaches this panel number
    int64_t ntiles;
    const unsigned* xcc_word;  // 1 + XCC_ID of the XCD the diagonal-block kernels run on (0: not known yet)
    unsigned* claim;
    unsigned max_exit;
    // place 3 (gemm_f64.hip: gemm_f64_persist_body): resident workgroups claim tiles from claim[0] until none is left; a workgroup
    // that finds itself on one of the ncu_res lowest-ranked CUs of its shader engine (cu_rank: [xcc][se][cu_id] -> rank) leaves at
    // once, at most max_exit of them (claim[1])
    const unsigned char* cu_rank;
    int ncu_res;
    // triangular operands: the contraction of a tile runs over [kbeg, kend) only (multiples of 128, from the tile's offsets)
    //   1: kbeg = m0  (op(A)(m, k) = 0 for k < m: the transpose of a lower-triangular matrix)
    //   2: kbeg = n0  (op(B)(k, n) = 0 for k < n: a lower-triangular matrix as the right operand)
    //   4: kend = m0 + 128  (op(A)(m, k) = 0 for k > m: a lower-triangular matrix as the left operand)
    int tri;
    // mirrored row tiles (> 0: the number of row tiles of the product; tiles_m is then half of it): a workgroup computes row
    // tile tm and then row tile mirror_tiles - 1 - tm -- with a triangular left operand (tri 1 / 4) the two contractions add
    // up to the same length for every workgroup (the big solve leaves: chol.hip)
    int64_t mirror_tiles;
    // split-K launches: batch member z holds the slice [z * kslice, min((z + 1) * kslice, k_total)) of the contraction (K == kslice;
    // the last slice may be shorter and ragged); the triangular restriction above is applied in the coordinates of the whole
    // contraction (0: not sliced)
    int64_t kslice, k_total;
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

// ---- XCD reservation -----------------------------------------------------------------------------------------------------
// While the panel chain bounds a factorisation, the XCD on which the panel stream's diagonal-block kernel runs is left to
// it (a dependent f64 operation of the pivot chain costs 7 cycles on a CU of its own and 54 - 85 next to a GEMM workgroup's
// MFMAs).  Single-workgroup launches of one stream always land on the same XCD (scripts/xcc_single.hip); the
// diagonal-block kernel publishes which (xcc_word = 1 + XCC_ID); a launch of that stream deals workgroup b to XCD
// (that one + b) % 8, so `nres` XCDs can be set aside for the panel stream: its launches use the workgroups with b % 8 <
// nres only (place 2, a pure function of b), everybody else keeps off the nres XCDs from the published one on.  place 1: a workgroup dealt to that XCD retires at once
// (its slot is free again within a microsecond, so the dispatcher's round over the engines never waits there).
// Correct whatever the dispatcher does: work items are CLAIMED (one
// atomic on claim[0]) and at most max_exit workgroups may retire without one (claim[1]) -- the grid holds n + max_exit
// workgroups.  Returns the claimed item or -1.
__device__ __forceinline__ long long claim_item(int place, int nres, unsigned epoch, const unsigned* xcc_word, unsigned* claim,
                                                unsigned max_exit, int64_t n)
{
    __shared__ long long claimed;
    if (threadIdx.x == 0) {
        unsigned phys;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(phys));
        const unsigned word = __hip_atomic_load(xcc_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the panel whose chain the reservation protects is finished (its stream said so): the XCDs are everybody's again
        const unsigned released = __hip_atomic_load(xcc_word + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool held = (int)(released - epoch) < 0;
        const bool known = word != 0u;
        const unsigned dist = (phys - (word - 1u)) & 7u;  // 0: the XCD of the diagonal-block kernels
        (void)place;
        const bool want = !(held && known && dist < (unsigned)nres);  // trailing update: keep off the reserved XCDs
        bool retire = false;
        if (!want) retire = __hip_atomic_fetch_add(claim + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < max_exit;
        long long tl = -1;
        if (!retire) {
            const unsigned i = __hip_atomic_fetch_add(claim, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((int64_t)i < n) tl = (long long)i;
        }
        claimed = tl;
    }
    __syncthreads();
    return claimed;
}

// One 128 x 128 result tile at (m0, n0): the whole K-loop and the epilogue (one tile per workgroup).
template <bool A_KMAJ, bool B_KMAJ>
__device__ __forceinline__ void gemm_f64_tile(const GemmArgs& g, double* lds, const int64_t m0, const int64_t n0)
{

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

// ---- 32 x 128 result tile (the launches of the panel chain) -----------------------------------------------------------------
// A 128 x 128 tile is 13.6 us of matrix-core time per 128 of contraction depth on ONE CU, and the products between two
// diagonal-block kernels of a factorisation have a few dozen such tiles (rows below the block x 128 or 256 columns): three
// quarters of the chip idle while the chain waits for the one busy quarter.  The same product cut into 32-row tiles has four
// times the workgroups, each a quarter of the matrix-core work, and still ONE tile column for N <= 128 -- which the in-place
// "B <- B W^T" of the panel solves needs.  Four waves side by side along n, each 32 x 32 (2 x 2 accumulators); op(B) tiles in
// the layouts of the large kernel, the 32-wide op(A) tile as [k][48] (m-major: 2 * 48 = 32 mod 64 banks, conflict-free
// fragment reads) or [m][18] (k-major).  No pinned schedule: these launches are bound by latency, not by issue slots.
constexpr int BMS = 32;
constexpr int S_MMAJ_S = 48;
constexpr int TILE_A_S = 16 * S_MMAJ_S;  // 768 >= 32 * 18

template <bool KMAJ>
__device__ __forceinline__ int lds_idx_s(int x, int k)
{
    return KMAJ ? x * S_KMAJ + k : k * S_MMAJ_S + x;
}

template <bool KMAJ>
__device__ __forceinline__ void load_tile_s(const double* __restrict__ P, int64_t ld, int64_t x0, int64_t X, int64_t k0, int64_t K,
                                            int t, double (&reg)[2])
{
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int x = KMAJ ? (t >> 4) + 16 * i : (t & 31);
        const int k = KMAJ ? (t & 15) : (t >> 5) + 8 * i;
        const bool ok = (x0 + x) < X && (k0 + k) < K;
        const double* p = KMAJ ? P + (k0 + k) + (x0 + x) * ld : P + (x0 + x) + (k0 + k) * ld;
        reg[i] = ok ? *p : 0.0;
    }
}

template <bool KMAJ>
__device__ __forceinline__ void store_tile_s(double* __restrict__ S, int t, const double (&reg)[2])
{
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int x = KMAJ ? (t >> 4) + 16 * i : (t & 31);
        const int k = KMAJ ? (t & 15) : (t >> 5) + 8 * i;
        S[lds_idx_s<KMAJ>(x, k)] = reg[i];
    }
}

// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, i.e. it would wait for the global loads
// that were issued precisely in order to stay in flight across the barrier.
__device__ __forceinline__ void lds_only_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// K-steps the 32-row tile's loads run ahead of its MFMAs (register sets in rotation); 4, 6 and 8 measure the same on the fits
// (N = 2048 / 4096 / 8192: 1.39 / 2.84 / 7.71, 1.40 / 2.86 / 7.79, 1.41 / 2.87 / 7.74 ms), 2 was 7 / 7 / 3 % slower
#ifndef M32_PF
#define M32_PF 4
#endif
template <bool A_KMAJ, bool B_KMAJ>
__device__ __forceinline__ void gemm_f64_tile_m32(const GemmArgs& g, double* lds, const int64_t m0, const int64_t n0)
{
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wn = t >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    d4_t acc[2][2];  // [nt][mt]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};
    const int64_t nk = (g.K + BK - 1) / BK;
    const int64_t nk_full = g.K / BK;
    const bool b_fast = (n0 + BN) <= g.N;
    const double* pb = tile_thread_base<B_KMAJ>(g.B, g.ldb, n0, t);
    const int64_t step_b = B_KMAJ ? BK : BK * g.ldb;
    constexpr int STAGE = TILE_A_S + TILE_ELEMS;
    if ((m0 + BMS) <= g.M && b_fast && nk_full == nk && nk >= M32_PF) {
        // Interior tile, K a multiple of 16.  These launches have one workgroup per CU (or per slot of a reserved XCD) and 16
        // MFMAs per wave and K-step (0.4 us) against a memory round trip of 1 - 2 us: what a tile costs is how many round
        // trips it exposes.  So (round 3) the loads run FOUR K-steps ahead of the MFMAs (PF register sets in rotation: slice
        // kt + 1 waits to be stored to LDS, kt + 2 .. kt + PF are in flight) -- a K = 128 product, eight K-steps, has half its
        // operands requested before the first MFMA -- and the C tile is requested BEFORE the K-loop instead of behind it (it is
        // never an operand of the same product: callers guarantee D != B, and where D aliases A -- the in-place solves -- beta is
        // 0 and nothing is read).  Unpredicated loads off one pointer per operand.
        const double* pa = A_KMAJ ? g.A + (t & 15) + (m0 + (t >> 4)) * g.lda : g.A + (m0 + (t & 31)) + (int64_t)(t >> 5) * g.lda;
        const int64_t step_a = A_KMAJ ? BK : BK * g.lda;
        const int64_t ea = (A_KMAJ ? 16 : 8) * g.lda;  // second element of the thread's pair
        const double* pbn = pb;
        auto load4 = [&](double (&ra)[2], double (&rb)[8]) {
            ra[0] = pa[0];
            ra[1] = pa[ea];
            load_tile_fast<B_KMAJ>(pbn, g.ldb, rb);
            pa += step_a;
            pbn += step_b;
        };
        constexpr int PF = M32_PF;  // register sets = K-steps the loads run ahead
        double ra[PF][2], rb[PF][8];
#pragma unroll
        for (int j = 0; j < PF; ++j) load4(ra[j], rb[j]);
        // register r of tile (nt, mt) holds D[m0 + 16 mt + (lane & 15)][n0 + 32 wn + 16 nt + (lane >> 4) + 4 r]
        const bool use_c = g.beta != 0.0;
        const double* cbase = g.Cin + (m0 + l15) + (n0 + wn * 32 + lq) * g.ldcin;
        double cv[16];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) cv[(nt * 4 + r) * 2 + mt] = use_c ? cbase[mt * 16 + (int64_t)(nt * 16 + 4 * r) * g.ldcin] : 0.0;
        store_tile_s<A_KMAJ>(lds, t, ra[0]);
        store_tile<B_KMAJ>(lds + TILE_A_S, t, rb[0]);
        lds_only_barrier();  // (the slices and the C tile stay in flight across the barrier)
        int cur = 0;
        int64_t left = nk - PF;  // slices not requested yet
        // one K-step: request the next slice into the set slice kt came from, multiply slice kt (in LDS), store slice kt + 1
        auto kstep = [&](bool last, double (&la)[2], double (&lb)[8], const double (&sa)[2], const double (&sb)[8]) {
            if (left > 0) {
                load4(la, lb);
                --left;
            }
            const double* As = lds + cur * STAGE;
            const double* Bs = As + TILE_A_S;
#pragma unroll
            for (int ks = 0; ks < BK / 4; ++ks) {
                const int kq = ks * 4 + lq;
                double af[2], bf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    af[i] = As[lds_idx_s<A_KMAJ>(i * 16 + l15, kq)];
                    bf[i] = Bs[lds_idx<B_KMAJ>(wn * 32 + i * 16 + l15, kq)];
                }
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
                        acc[nt][mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[nt], af[mt], acc[nt][mt], 0, 0, 0);
            }
            if (!last) {
                double* An = lds + (cur ^ 1) * STAGE;
                store_tile_s<A_KMAJ>(An, t, sa);
                store_tile<B_KMAJ>(An + TILE_A_S, t, sb);
            }
            lds_only_barrier();  // (the slices in flight stay in flight)
            cur ^= 1;
        };
        for (int64_t kt0 = 0; kt0 < nk; kt0 += PF) {
#pragma unroll
            for (int j = 0; j < PF; ++j) {  // (unrolled: the set indices are compile-time constants)
                if (kt0 + j < nk) kstep(kt0 + j + 1 >= nk, ra[j], rb[j], ra[(j + 1) % PF], rb[(j + 1) % PF]);
            }
        }
        double* dbase = g.D + (m0 + l15) + (n0 + wn * 32 + lq) * g.ldd;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    double v = g.alpha * acc[nt][mt][r];
                    if (use_c) v += g.beta * cv[(nt * 4 + r) * 2 + mt];
                    dbase[mt * 16 + (int64_t)(nt * 16 + 4 * r) * g.ldd] = v;
                }
        return;
    }
    // The loads run TWO K-steps ahead of the MFMAs (registers: the slice for step kt + 1 waits to be stored, the one for
    // kt + 2 is in flight): these launches have one workgroup per CU and 16 MFMAs per wave and K-step, so a K-step costs a
    // memory round trip unless two are outstanding.
    auto load_slice = [&](int64_t kt, double (&ra)[2], double (&rb)[8]) {
        load_tile_s<A_KMAJ>(g.A, g.lda, m0, g.M, kt * BK, g.K, t, ra);
        if (b_fast && kt < nk_full)
            load_tile_fast<B_KMAJ>(pb + kt * step_b, g.ldb, rb);
        else
            load_tile<B_KMAJ>(g.B, g.ldb, n0, g.N, kt * BK, g.K, t, rb);
    };
    double ra1[2], rb1[8], ra2[2], rb2[8];
    if (nk > 0) {
        load_slice(0, ra1, rb1);
        store_tile_s<A_KMAJ>(lds, t, ra1);
        store_tile<B_KMAJ>(lds + TILE_A_S, t, rb1);
        if (nk > 1) load_slice(1, ra1, rb1);
    }
    __syncthreads();
    int cur = 0;
    for (int64_t kt = 0; kt < nk; ++kt) {
        if (kt + 2 < nk) load_slice(kt + 2, ra2, rb2);
        const double* As = lds + cur * STAGE;
        const double* Bs = As + TILE_A_S;
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
            const int kq = ks * 4 + lq;
            double af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = As[lds_idx_s<A_KMAJ>(i * 16 + l15, kq)];
                bf[i] = Bs[lds_idx<B_KMAJ>(wn * 32 + i * 16 + l15, kq)];
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[nt], af[mt], acc[nt][mt], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            double* An = lds + (cur ^ 1) * STAGE;
            store_tile_s<A_KMAJ>(An, t, ra1);
            store_tile<B_KMAJ>(An + TILE_A_S, t, rb1);
        }
        lds_only_barrier();  // (the slice in flight stays in flight)
        cur ^= 1;
#pragma unroll
        for (int i = 0; i < 2; ++i) ra1[i] = ra2[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) rb1[i] = rb2[i];
    }
    // epilogue: register r of tile (nt, mt) holds D[m0 + 16 mt + (lane & 15)][n0 + 32 wn + 16 nt + (lane >> 4) + 4 r];
    // all 16 values of C are loaded before the first store (D may alias Cin: one round trip, not sixteen)
    const bool use_c = g.beta != 0.0;
    double cv[16];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int64_t n = n0 + wn * 32 + nt * 16 + lq + 4 * r, m = m0 + mt * 16 + l15;
                cv[(nt * 4 + r) * 2 + mt] = (use_c && n < g.N && m < g.M) ? g.Cin[m + n * g.ldcin] : 0.0;
            }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int64_t n = n0 + wn * 32 + nt * 16 + lq + 4 * r, m = m0 + mt * 16 + l15;
                double v = g.alpha * acc[nt][mt][r];
                if (use_c) v += g.beta * cv[(nt * 4 + r) * 2 + mt];
                if (n < g.N && m < g.M) g.D[m + n * g.ldd] = v;
            }
}


}  // namespace fr
