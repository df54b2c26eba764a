// trsm_narrow.hip -- K9: triangular solves with 2 .. 16 right-hand sides (predict / predict_variance / sample_at of a
// handful of query points, mod.rs:235, 260-263, 342-345, 379) as ONE persistent launch per direction.
//
// Same organisation as the single-column solves of trsv.hip -- 128-row blocks dealt to the CUs, a block's owner streams its
// row of 128 x 128 tiles while the solution blocks it depends on become available, closes with the explicit inverse of the
// diagonal block, hands its solution block over -- with the products on the FP64 matrix core: a wave owns 16 rows of the
// block and all 16 (padded) right-hand sides, one v_mfma_f64_16x16x4 accumulator, tile fragments straight from memory
// (rows along the lanes: 128-byte segments), so there is no reduction across lanes or waves anywhere.
//
// The backward sweep runs the SAME kernel on a transposed copy of the factor: the strict upper triangle of the factor's
// buffer is free (the reference leaves NaN there, algebra/mod.rs:67), so the off-diagonal 128 x 128 blocks of L^T are kept
// in it (and the transposed inverse blocks next to the inverse blocks), rebuilt lazily after the factor changed
// (~2 ms at n = 32768).  With it "L^T x = b" reads rows-along-lanes exactly like "L x = b".
//
// Hand-off of a solution block (128 x 16 doubles).  Column-group kernel (m > 16): write-through (sc1) stores -> every wave
// drains -> barrier -> flag; the consumer polls the flag and reads the block with sc1 loads (cdna_hip_programming.md
// Guideline 16, R1).  Single-group kernel (m <= 16): the payload is its own flag (sentinel-filled buffer; see the kernel).
// Every spin is bounded (handoff.hpp); a timed-out launch is repeated on the recursive path by the entry point (solve_retry).
#include "fr_internal.hpp"
#include "handoff.hpp"

namespace fr {

typedef double d4n_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) double gdbl;
typedef double d2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) volatile d2_t gvd2;

constexpr int NB = 128;   // block
constexpr int MR = 16;    // right-hand sides per launch (padded)
constexpr int NTH = 512;  // 8 waves x 16 rows

struct TrsmnArgs {
    const double* A;    // factor buffer: lower triangle L; strict upper off-diagonal blocks hold L^T (backward)
    int64_t ld, n;
    const double* inv;  // inverse blocks (forward) or transposed inverse blocks (backward)
    const double* mchain;  // single-group kernel: the chain products M_b = inv_b * (tile next to the diagonal), block b at b * 128 * 128
    double* B;          // n x m right-hand sides in, solutions out
    int64_t ldb;
    int m;
    double* xg;         // hand-off payload: block g at xg + g * NB * MR, [col][q]
    int* flags;         // one per block
    unsigned* tickets;  // one block claim counter per column group (zeroed with the flags before every launch)
    unsigned* status;
    int nblk, bwd;
    int ngroups;        // column groups of 16 right-hand sides (blockIdx.y); hand-off state per group
};

// The next unclaimed block of this column group's sweep, or -1 (uniform).  Blocks are taken in order of arrival, so a block
// only waits for blocks held by workgroups that are already running: no co-residency needed (see trsv.hip).
__device__ __forceinline__ int claim_block(unsigned* ticket, int nblk, int* slot)
{
    if (threadIdx.x == 0) {
        const unsigned i = __hip_atomic_fetch_add((hgu32*)ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *slot = i < (unsigned)nblk ? (int)i : -1;
    }
    __syncthreads();
    return __builtin_amdgcn_readfirstlane(*slot);
}

struct Frag {
    const double* base;  // element (0, 0)
    int64_t stride;      // column (k) stride
    int mrows, kcols;    // valid extent
};

// 16 rows (of this wave) x 128 columns: the lane holds element (m = 16 w + l15, k = 4 u + lq), u = 0 .. 31
__device__ __forceinline__ void load_frag(const Frag& f, int w, int l15, int lq, double (&buf)[32])
{
    // address = wave-uniform part (block, column group: scalar registers) + one 32-bit lane offset shared by all 32 loads
    const unsigned lane_off = (unsigned)(16 * w + l15) + (unsigned)lq * (unsigned)f.stride;
#pragma unroll
    for (int u = 0; u < 32; ++u) buf[u] = (f.base + (int64_t)(4 * u) * f.stride)[lane_off];
    if (f.mrows < NB || f.kcols < NB) {  // last block only: zeros outside the valid extent (the addresses exist)
        const bool mok = 16 * w + l15 < f.mrows;
#pragma unroll
        for (int u = 0; u < 32; ++u) buf[u] = (mok && 4 * u + lq < f.kcols) ? buf[u] : 0.0;
    }
}

// 16 rows (of this wave) x 64 columns (half H of a 128-column tile): the lane holds element (m = 16 w + l15,
// k = 64 H + 4 u + lq), u = 0 .. 15.  Half tiles keep the two register buffers at 64 VGPRs together, so that two
// workgroups fit a CU: the chains of different column groups (and the tiles of one) then overlap on every CU.
__device__ __forceinline__ void load_half(const Frag& f, int H, int w, int l15, int lq, double (&buf)[16])
{
    // address = wave-uniform part (block, column group: scalar registers) + one 32-bit lane offset shared by all 16 loads
    const unsigned lane_off = (unsigned)(16 * w + l15) + (unsigned)lq * (unsigned)f.stride;
    const double* base = f.base + (int64_t)(64 * H) * f.stride;
#pragma unroll
    for (int u = 0; u < 16; ++u) buf[u] = (base + (int64_t)(4 * u) * f.stride)[lane_off];
    if (f.mrows < NB || f.kcols < NB) {  // last block only: zeros outside the valid extent (the addresses exist)
        const bool mok = 16 * w + l15 < f.mrows;
        const int klim = f.kcols - 64 * H;  // (uniform: one lane value, lq, against sixteen scalars)
#pragma unroll
        for (int u = 0; u < 16; ++u) buf[u] = (mok && lq < klim - 4 * u) ? buf[u] : 0.0;
    }
}

__device__ __forceinline__ Frag item_frag(const TrsmnArgs& a, int blk, int other, bool inverse)
{
    Frag f;
    const int64_t b0 = (int64_t)blk * NB;
    const int64_t left = a.n - b0;
    f.mrows = left < NB ? (int)left : NB;
    if (inverse) {
        f.base = a.inv + (int64_t)blk * (NB * NB);
        f.stride = NB;
        f.kcols = f.mrows;
    } else {
        const int64_t o0 = (int64_t)other * NB;
        f.base = a.A + b0 + o0 * a.ld;  // forward: tile (blk, other) of L; backward: block (blk, other) of the upper copy
        f.stride = a.ld;
        const int64_t oleft = a.n - o0;
        f.kcols = oleft < NB ? (int)oleft : NB;
    }
    return f;
}

// The kernel for TWO OR MORE column groups (m > 16): half-tile register buffers, one LDS buffer -- 128 VGPRs and 16 KiB, so two
// workgroups share a CU and the chains of different groups (each a sequence of dependent hand-offs, ~10 us apiece) run side
// by side.  Measured against the kernel below on the same solves: 8 x add_samples(512) at 4096 .. 8192 rows 18.3 -> 14.9 ms,
// sample_at(256) 4.0 -> 3.3 ms; with ONE group the deeper prefetch of the kernel below wins (predict of 16 points at
// N = 32768: 4.7 vs 5.5 ms).  NQ = 2: a group is two MFMA tiles of 16 right-hand sides fed by the same factor fragments (half the
// passes over the factor, half the workgroups); LDS rows are then padded by 16 doubles, so that the four k-rows a fragment
// read touches start 32 banks apart.
template <int NQ>
__global__ __launch_bounds__(NTH, 4) void trsm_narrow_half_kernel(const TrsmnArgs a0)
{
    constexpr int MRT = 16 * NQ, S = MRT + (NQ > 1 ? 16 : 0);
    __shared__ double xs[NB * S];  // a solution block while it is multiplied; then t = b - sum for the closing product
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int last = a0.nblk - 1;
    // column group: 16 NQ right-hand sides with their own solution blocks and flags -- the groups are independent chains that
    // stream the same tiles (served from L2 / Infinity Cache after the first reader)
    TrsmnArgs a = a0;
    {
        const int grp = blockIdx.y;
        a.B += (int64_t)grp * MRT * a.ldb;
        a.m = a.m - grp * MRT < MRT ? a.m - grp * MRT : MRT;
        a.xg += (int64_t)grp * a.nblk * (NB * MRT);
        a.flags += (int64_t)grp * a.nblk;
        a.tickets += grp;
    }
    __shared__ int claim_slot;
#pragma nounroll
    for (;;) {
        const int bi = claim_block(a.tickets, a.nblk, &claim_slot);
        if (bi < 0) return;
        const int blk = a.bwd ? last - bi : bi;
        const int cnt = a.bwd ? last - blk : blk;  // tiles; item q uses the solution block dep(q); item cnt: the inverse block
        const int64_t b0 = (int64_t)blk * NB;
        const int64_t row = b0 + 16 * w + l15;
        // the accumulators start at -b (this lane's right-hand side entries (row, q = 16 j + lq + 4 i)): after the tiles they
        // hold -(b - sum L x), and no copy of b lives across the loop
        d4n_t acc[NQ];
#pragma unroll
        for (int j = 0; j < NQ; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = 16 * j + lq + 4 * i;
                acc[j][i] = (row < a.n && q < a.m) ? -a.B[row + (int64_t)q * a.ldb] : 0.0;
            }
        double H0[16], H1[16];  // first / second half of the current item; the next half is always in flight
        load_half(item_frag(a, blk, a.bwd ? last : 0, cnt == 0), 0, w, l15, lq, H0);
#pragma nounroll
        for (int q = 0; q <= cnt; ++q) {
            const Frag f = item_frag(a, blk, a.bwd ? last - q : q, q >= cnt);
            load_half(f, 1, w, l15, lq, H1);
            if (q < cnt) {
                // solution block dep(q) -> xs; the barrier inside the wait also tells that every wave is done with the block before
                const int dep = a.bwd ? last - q : q;
                if (!handoff_wait_ge<false>(a.flags + dep, 1, a.status)) return;  // no acquire fence: sc1 stores, sc1 loads
                const double* src = a.xg + (int64_t)dep * (NB * MRT);
#pragma unroll
                for (int i = 0; i < (NB * MRT) / NTH; ++i) {
                    const int e = t + NTH * i;
                    xs[(e / MRT) * S + (e % MRT)] = __hip_atomic_load((gdbl*)(src + e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __syncthreads();
            } else {
                // t = b - sum = -acc, the [col][q] operand of the closing product with the inverse block (same buffer: every
                // wave is done with the last solution block first)
                __syncthreads();
#pragma unroll
                for (int j = 0; j < NQ; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        xs[(16 * w + l15) * S + 16 * j + lq + 4 * i] = -acc[j][i];
                        acc[j][i] = 0.0;
                    }
                __syncthreads();
            }
#pragma unroll
            for (int H = 0; H < 2; ++H) {
                if (H == 1 && q < cnt) load_half(item_frag(a, blk, a.bwd ? last - q - 1 : q + 1, q + 1 >= cnt), 0, w, l15, lq, H0);
#pragma unroll
                for (int u = 0; u < 16; ++u) {
#pragma unroll
                    for (int j = 0; j < NQ; ++j) {
                        const double xf = xs[(64 * H + 4 * u + lq) * S + 16 * j + l15];  // element (k = 64 H + 4 u + lq, q = 16 j + l15)
                        acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(xf, H == 0 ? H0[u] : H1[u], acc[j], 0, 0, 0);
                    }
                }
            }
        }
        // publish (write-through), then the caller's copy
        double* dst = a.xg + (int64_t)blk * (NB * MRT);
#pragma unroll
        for (int j = 0; j < NQ; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __hip_atomic_store((gdbl*)(dst + (16 * w + l15) * MRT + 16 * j + lq + 4 * i), acc[j][i], __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) __hip_atomic_store((hgi32*)(a.flags + blk), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int j = 0; j < NQ; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = 16 * j + lq + 4 * i;
                if (row < a.n && q < a.m) a.B[row + (int64_t)q * a.ldb] = acc[j][i];
            }
        __syncthreads();  // xs is reused by the next block of this workgroup
    }
}

// (Round 4 built the 64-column generalisation of this kernel -- trsm_narrow_wide_kernel<4>: four accumulator tiles per wave on the
// same factor fragments, a swizzled 64 KiB operand block, the two nearest dependencies as cached banded products M, M2 -- and the
// 32-column pair variant of round 3 was still here.  Both measured slower than groups of 16 wherever column groups are used at all
// (the chain step of a 64-column group carries a 128 x 128 x 64 product, 6.8 us of a CU's matrix
// cores) and were removed; the code is in the history at 6fd4683.  Solves with more right-hand sides take the 2048-row leaves of
// chol.hip.)

// ---- one column group (m <= 16) ------------------------------------------------------------------------------------------
// The solve is a chain of n / 128 hand-offs; what one step costs is what the whole solve costs.  Round 2's step was: flag poll,
// payload fetch, the product with the neighbouring tile (1.7 us: 64 MFMAs per SIMD), a transposition through LDS, the closing
// product with the inverse block (1.7 us), payload stores, their acknowledgement, the flag -- ~9 us.  Now (timed with
// s_memrealtime stamps per stage and per wave; the numbers are in DESIGN.md section 5):
//   * the last dependency of block r is its NEIGHBOUR x_{r-1}; everything that does not need it is done before it arrives:
//       t' = b_r - sum_{q < r-1} L[r, q] x_q,   y' = W_r t',   and   M_r = W_r L[r, r-1],
//     the chain product: it depends on the factor only, so it is formed once per factor (ensure_chain_products: one batched
//     128^3 product, cached like the transposed copy) and sits in LDS while the block is worked on;
//   * when x_{r-1} arrives ONE product is left on the chain:  x_r = y' - M_r x_{r-1};
//   * EVERY solution block is awaited on the payload itself: the hand-off buffer is filled with a NaN bit pattern no
//     computation produces before every launch, a lane polls its two 16-byte pieces until no double is the pattern (aligned
//     8-byte halves are single-copy atomic, each element is written once) -- no flag, no acknowledgement wait on the
//     producer's side, and the block of the NEXT dependency is fetched while the current one is multiplied (a flag would have
//     to be seen first: two dependent round trips per dependency, which is what bounded every block's progress before);
//   * the two waits on the chain (x_{r-2}: the last tile and the closing product follow it; x_{r-1}) poll with several samples
//     in flight;
//   * a block travels as 128 full cache lines (see below).
constexpr unsigned SENT32 = 0x7FF7A5A5u;  // both halves: hipMemsetD32 fills the payload with the (NaN) double 0x7FF7A5A57FF7A5A5
__device__ __forceinline__ bool is_sentinel(double v) { return (unsigned long long)__double_as_longlong(v) == 0x7FF7A5A57FF7A5A5ull; }

// ---- the payload of a solution block, awaited on the data itself ------------------------------------------------------------
// A block travels as 128 full cache lines ([col k][16 right-hand sides], 16 KiB): the producer's waves transpose their 16 rows
// through LDS and write them with 16-byte stores (two instructions of 1 KiB per wave; the accumulator layout would scatter 512
// quarter-lines per workgroup, and the consumer's barrier waits for the LAST of them); a consumer lane owns two 16-byte pieces
// (doubles 2 t, 2 t + 1 and 1024 + 2 t, 1024 + 2 t + 1).  Volatile accesses: system-scope (sc0 sc1) loads and write-through
// stores, which need no fence (Guideline 16, R1); every double is written once and none can be the sentinel.
struct Piece {
    d2_t lo, hi;
};
// blockbase: the block's 16 KiB (wave-uniform: a scalar base, the lane's offset is one 32-bit register)
__device__ __forceinline__ void issue_payload(const double* blockbase, int t, Piece& v)
{
    const gvd2* p = (const gvd2*)blockbase;
    v.lo = p[t];
    v.hi = p[NTH + t];
}
__device__ __forceinline__ bool payload_complete(const Piece& v)
{
    return !(is_sentinel(v.lo.x) || is_sentinel(v.lo.y) || is_sentinel(v.hi.x) || is_sentinel(v.hi.y));
}
__device__ __forceinline__ void payload_to_lds(double* xb, int t, const Piece& v)
{
    *reinterpret_cast<d2_t*>(xb + 2 * t) = v.lo;
    *reinterpret_cast<d2_t*>(xb + 2 * NTH + 2 * t) = v.hi;
}
__device__ __forceinline__ bool handoff_dead(const TrsmnArgs& a, unsigned long long t0)
{
    const bool dead = __hip_atomic_load((hgu32*)a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
    if (dead || wall_clock64() - t0 > HANDOFF_TIMEOUT_TICKS) {
        __hip_atomic_store((hgu32*)a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return true;
    }
    return false;
}
// v: a sample already issued (possibly incomplete).  Returns 0 after a timeout (per lane: combine with a barrier vote).
__device__ __forceinline__ int await_payload(const TrsmnArgs& a, const double* src, int t, Piece& v)
{
    if (payload_complete(v)) return 1;
    const unsigned long long t0 = wall_clock64();
    unsigned spins = 0;
    for (;;) {
        issue_payload(src, t, v);
        if (payload_complete(v)) return 1;
        if ((++spins & 127u) == 0 && handoff_dead(a, t0)) return 0;
    }
}
// The same with DEPTH samples in flight: the memory is sampled DEPTH times per round trip (a poll's round trip is 1.5 - 2 us
// while the chip streams the factor -- scripts/handoff_probe.hip -- and a serial poll loop adds on average a whole round trip to
// the hand-off: half of one until the next sample is taken, half for its way back; and the barrier behind the wait makes that
// the LATEST of eight waves).  For the two hand-offs on the chain.
template <int DEPTH>
__device__ __forceinline__ int await_payload_pipelined(const TrsmnArgs& a, const double* src, int t, Piece& v)
{
    if (payload_complete(v)) return 1;
    Piece ring[DEPTH];
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) issue_payload(src, t, ring[i]);
    const unsigned long long t0 = wall_clock64();
    unsigned spins = 0;
    for (;;) {
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) {
            if (payload_complete(ring[i])) {
                v = ring[i];
                return 1;
            }
            issue_payload(src, t, ring[i]);
        }
        if ((++spins & 63u) == 0 && handoff_dead(a, t0)) return 0;
    }
}

// LDS (dynamic, 160 KiB = all of a CU's): two solution-block buffers and the block's chain product M, stored fragment-major
// (wave w, slot u, lane: the lane's MFMA operand (m = 16 w + l15, k = 4 u + lq) -- consecutive lanes, consecutive words).  M in
// LDS rather than in 64 registers per lane: the registers hold tile quarters in flight instead (the solve is latency-bound
// on exactly those), and the end of the row can hold the last tile AND the inverse block before the solution block it waits
// for arrives.
constexpr size_t TRSMN_LDS = sizeof(double) * (2 * NB * MR + NB * NB);

__global__ __launch_bounds__(NTH, 2) void trsm_narrow_kernel(const TrsmnArgs a0)
{
    extern __shared__ __attribute__((aligned(16))) double k9_lds[];
    double* const xs0 = k9_lds;
    double* const xs1 = k9_lds + NB * MR;
    double* const Ml = k9_lds + 2 * NB * MR;
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int last = a0.nblk - 1;
    TrsmnArgs a = a0;  // (one column group: blockIdx.y == 0)
    // A timed-out wait raises the host-visible status word and goes on with whatever it has: every later wait then gives up
    // within a few polls, the launch drains, and the host repeats the solve on the recursive path (solve_retry) -- no vote,
    // no early exit.
#pragma nounroll
    for (;;) {
        // claim (the slot aliases the first solution-block buffer: free between blocks; the second barrier keeps a fast wave's
        // next write to the buffer behind everybody's read of the slot)
        if (t == 0) {
            const unsigned i = __hip_atomic_fetch_add((hgu32*)a.tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *reinterpret_cast<int*>(xs0) = i < (unsigned)a.nblk ? (int)i : -1;
        }
        __syncthreads();
        const int bi = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(xs0));
        __syncthreads();
        if (bi < 0) return;
        const int blk = a.bwd ? last - bi : bi;
        const int cnt = a.bwd ? last - blk : blk;  // dependencies: blocks dep(0) .. dep(cnt - 1), the neighbour last
        const int64_t b0 = (int64_t)blk * NB;
        const int64_t row = b0 + 16 * w + l15;
        // the accumulator starts at -b (this lane's right-hand side entries (row, q = lq + 4 i)): after the tiles it holds
        // -(b - sum L x) = -t'
        d4n_t acc;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = (row < a.n && lq + 4 * i < a.m) ? -a.B[row + (int64_t)(lq + 4 * i) * a.ldb] : 0.0;
        // ---- M = W L[blk, neighbour]: the chain product, one per block row of the factor (ensure_chain_products) -> LDS
        if (cnt > 0) {
            Frag fm;
            fm.base = a.mchain + (int64_t)blk * (NB * NB);
            fm.stride = NB;
            fm.mrows = fm.kcols = NB;  // (rows / columns outside a last, partial block: see ensure_chain_products)
            double Mf[32];
            load_frag(fm, w, l15, lq, Mf);
#pragma unroll
            for (int u = 0; u < 32; ++u) Ml[(w * 32 + u) * 64 + lane] = Mf[u];  // (read back by the same wave only)
        }
        // ---- the items in front of the neighbour: the tiles L[blk, dep(it)], it < ndep, then the inverse block (operand t').
        //      QUARTER-tile register buffers (16 rows x 32 columns, 8 doubles per lane), loaded three quarters ahead (48 KiB per
        //      workgroup in flight: what the CU's share of the HBM rate needs at 2 us of latency); the solution block of the
        //      NEXT item is already being fetched while this one is multiplied -- every block is awaited on its payload
        //      (sentinel), so fetching ahead of time needs no flag
        const int ndep = cnt > 0 ? cnt - 1 : 0;
        auto dep_frag = [&](int it) { return item_frag(a, blk, a.bwd ? last - it : it, false); };
        auto payload_of = [&](int it) { return a.xg + (int64_t)(a.bwd ? last - it : it) * (NB * MR); };
        auto load_quarter = [&](const Frag& f, int Hq, double (&buf)[8]) {
            const unsigned lane_off = (unsigned)(16 * w + l15) + (unsigned)lq * (unsigned)f.stride;
            const double* base = f.base + (int64_t)(32 * Hq) * f.stride;
#pragma unroll
            for (int u = 0; u < 8; ++u) buf[u] = (base + (int64_t)(4 * u) * f.stride)[lane_off];
            if (f.mrows < NB || f.kcols < NB) {  // last block only: zeros outside the valid extent (the addresses exist)
                const bool mok = 16 * w + l15 < f.mrows;
#pragma unroll
                for (int u = 0; u < 8; ++u) buf[u] = (mok && 32 * Hq + 4 * u + lq < f.kcols) ? buf[u] : 0.0;
            }
        };
        auto mma_quarter = [&](const double (&buf)[8], int Hq, const double* xb) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xb[(32 * Hq + 4 * u + lq) * MR + l15], buf[u], acc, 0, 0, 0);
        };
        {
            double Q0[8], Q1[8], Q2[8], Q3[8];
            Piece xv;
            Frag f = dep_frag(0);
            if (ndep > 0) {
                issue_payload(payload_of(0), t, xv);
                load_quarter(f, 0, Q0);
                load_quarter(f, 1, Q1);
                load_quarter(f, 2, Q2);
            }
            // steady part: every dependency but the last
#pragma nounroll
            for (int it = 0; it + 1 < ndep; ++it) {
                double* xb = (it & 1) ? xs1 : xs0;  // (two buffers: the barrier of item it - 1 lies between the reads of item it - 2 and these writes)
                await_payload(a, payload_of(it), t, xv);
                payload_to_lds(xb, t, xv);
                issue_payload(payload_of(it + 1), t, xv);
                __syncthreads();
                const Frag fn = dep_frag(it + 1);
                load_quarter(f, 3, Q3);
                mma_quarter(Q0, 0, xb);
                load_quarter(fn, 0, Q0);
                mma_quarter(Q1, 1, xb);
                load_quarter(fn, 1, Q1);
                mma_quarter(Q2, 2, xb);
                load_quarter(fn, 2, Q2);
                mma_quarter(Q3, 3, xb);
                f = fn;
            }
            // the end of the row is on the chain (the last dependency arrives one step before the neighbour): its last quarter and
            // the whole inverse block are in registers before it arrives
            double W0[8], W1[8], W2[8], W3[8];
            const Frag fw = item_frag(a, blk, 0, true);
            if (ndep > 0) load_quarter(f, 3, Q3);
            load_quarter(fw, 0, W0);
            load_quarter(fw, 1, W1);
            load_quarter(fw, 2, W2);
            load_quarter(fw, 3, W3);
            if (ndep > 0) {
                double* xb = ((ndep - 1) & 1) ? xs1 : xs0;
                await_payload_pipelined<2>(a, payload_of(ndep - 1), t, xv);
                payload_to_lds(xb, t, xv);
                __syncthreads();
                mma_quarter(Q0, 0, xb);
                mma_quarter(Q1, 1, xb);
                mma_quarter(Q2, 2, xb);
                mma_quarter(Q3, 3, xb);
            }
            // closing product y' = W t':  t' = -acc in the [col][q] layout of a solution block
            double* xb = (ndep & 1) ? xs1 : xs0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                xb[(16 * w + l15) * MR + lq + 4 * i] = -acc[i];
                acc[i] = 0.0;
            }
            __syncthreads();
            mma_quarter(W0, 0, xb);
            mma_quarter(W1, 1, xb);
            mma_quarter(W2, 2, xb);
            mma_quarter(W3, 3, xb);
        }
        d4n_t x = acc;  // y'
        // ---- the neighbour; ONE product on the chain:  x = y' - M x_neighbour
        if (cnt > 0) {
            double* xb = ((ndep + 1) & 1) ? xs1 : xs0;
            const double* src = a.xg + (int64_t)(a.bwd ? blk + 1 : blk - 1) * (NB * MR);
            Piece xv;
            issue_payload(src, t, xv);
            await_payload_pipelined<3>(a, src, t, xv);
            payload_to_lds(xb, t, xv);
            __syncthreads();
            d4n_t z = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int u = 0; u < 32; ++u)
                z = __builtin_amdgcn_mfma_f64_16x16x4f64(xb[(4 * u + lq) * MR + l15], Ml[(w * 32 + u) * 64 + lane], z, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) x[i] -= z[i];
        }
        // publish: this wave's 16 rows through LDS (the closing operand's buffer: every wave read it before the neighbour's
        // barrier; a wave reads back only what it wrote itself, and a wave's LDS operations complete in order) -> two 16-byte
        // write-through stores per lane, full cache lines.  No flag: every consumer of this kernel awaits the payload.
        if (cnt == 0) __syncthreads();  // (no neighbour, no barrier behind the closing product: the other waves may still read its operand)
        {
            double* xo = ((ndep & 1) ? xs1 : xs0) + 256 * w;
#pragma unroll
            for (int i = 0; i < 4; ++i) xo[l15 * MR + lq + 4 * i] = x[i];
            const d2_t p0 = *reinterpret_cast<const d2_t*>(xo + 2 * lane);
            const d2_t p1 = *reinterpret_cast<const d2_t*>(xo + 128 + 2 * lane);
            gvd2* dst = (gvd2*)(a.xg + (int64_t)blk * (NB * MR) + 256 * w);
            dst[lane] = p0;
            dst[64 + lane] = p1;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (row < a.n && lq + 4 * i < a.m) a.B[row + (int64_t)(lq + 4 * i) * a.ldb] = x[i];
        __syncthreads();  // the buffers are reused by the next block of this workgroup
    }
}


// ---- the transposed copy ---------------------------------------------------------------------------------------------
// upper off-diagonal blocks := transposes of the lower ones (the diagonal 128-blocks keep their zeroed upper triangle: the
// refined solve leaves use them as plain operands); 64 x 64 LDS tiles, coalesced on both sides
__global__ __launch_bounds__(256) void transpose_offdiag_kernel(double* A, int64_t n, int64_t ld)
{
    __shared__ double tile[64][65];
    const int64_t bi = blockIdx.x, bj = blockIdx.y;  // 64-blocks: source (bi, bj), bi > bj
    if (bj >= bi || (bi >> 1) == (bj >> 1)) return;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int c = ty; c < 64; c += 4) {
        const int64_t r = bi * 64 + tx, cc = bj * 64 + c;
        tile[c][tx] = (r < n && cc < n) ? A[r + cc * ld] : 0.0;
    }
    __syncthreads();
    for (int c = ty; c < 64; c += 4) {
        const int64_t r = bj * 64 + tx, cc = bi * 64 + c;  // element (r, cc) of the upper part = lower (cc, r)
        if (r < n && cc < n) A[r + cc * ld] = tile[tx][c];
    }
}

__global__ __launch_bounds__(256) void transpose_inv_kernel(const double* __restrict__ inv, double* __restrict__ invt)
{
    __shared__ double tile[64][65];
    const double* src = inv + (int64_t)blockIdx.z * (NB * NB);
    double* dst = invt + (int64_t)blockIdx.z * (NB * NB);
    const int bi = blockIdx.x, bj = blockIdx.y;  // 64-sub-blocks of the 128 x 128 block
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int c = ty; c < 64; c += 4) tile[c][tx] = src[(bi * 64 + tx) + (bj * 64 + c) * NB];
    __syncthreads();
    for (int c = ty; c < 64; c += 4) dst[(bj * 64 + tx) + (bi * 64 + c) * NB] = tile[tx][c];
}

static int ensure_transposed(fr_ctx* ctx, fr_chol* c)
{
    if (c->ut_gen == c->gen && c->ut_n == c->n && c->dinvt) return FR_OK;
    const int64_t nblk = (c->n + NB - 1) / NB;
    const int64_t cap_blk = (c->capacity + NB - 1) / NB;
    if (c->dinvt_cap < cap_blk) {
        if (c->dinvt) {
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipFree(c->dinvt);
            c->dinvt = nullptr;
            c->dinvt_cap = 0;
        }
        FR_HIP(ctx, dev_malloc(ctx, (void**)&c->dinvt, sizeof(double) * (size_t)cap_blk * NB * NB));
        c->dinvt_cap = cap_blk;
    }
    const int64_t nb64 = (c->n + 63) / 64;
    if (nb64 > 65535) return set_err(ctx, FR_INVALID_ARGUMENT, "matrix too large for the transposed copy");
    hipLaunchKernelGGL(transpose_offdiag_kernel, dim3((unsigned)nb64, (unsigned)nb64), dim3(256), 0, ctx->ls, c->A, c->n, c->ld_a);
    hipLaunchKernelGGL(transpose_inv_kernel, dim3(2, 2, (unsigned)nblk), dim3(256), 0, ctx->ls, c->dinv, c->dinvt);
    FR_HIP(ctx, hipGetLastError());
    c->ut_gen = c->gen;
    c->ut_n = c->n;
    return FR_OK;
}

// The chain products: M_b = W_b L[b, b - k] (forward) / W_b^T L[b + k, b]^T (backward; both operands from the transposed
// copy), k = the distance of the tile from the diagonal (1: the neighbour, what the single-group kernel uses; the routine is
// general in k: round 4's wide kernel also banded k = 2), b = every block with such a dependency --
// ONE batched product of 128^3 per block row (n = 32768: 1 GFLOP, 32 MiB per direction and distance), cached per factor like
// the transposed copy: valid for the factor's generation AND its row count (add_rows solves against the old rows under the
// new generation before the factor grows).  (Round 3 first formed M inside the solve, 31 us at the start of every block's
// life and 64 more live registers next to it.)
static int ensure_chain_products(fr_ctx* ctx, fr_chol* c, bool fwd, int k, int prof_cls)
{
    const int slot = (fwd ? 0 : 1) + 2 * (k - 1);
    const int64_t nblk = (c->n + NB - 1) / NB;
    if (nblk < k + 1) return FR_OK;
    if (c->mchain_gen[slot] == c->gen && c->mchain_n[slot] == c->n && c->mchain[slot]) return FR_OK;
    const int64_t cap_blk = (c->capacity + NB - 1) / NB;
    if (c->mchain_cap[slot] < cap_blk) {
        if (c->mchain[slot]) {
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipFree(c->mchain[slot]);
            c->mchain[slot] = nullptr;
            c->mchain_cap[slot] = 0;
        }
        FR_HIP(ctx, dev_malloc(ctx, (void**)&c->mchain[slot], sizeof(double) * (size_t)cap_blk * NB * NB));
        c->mchain_cap[slot] = cap_blk;
    }
    // full blocks as ONE batched product; a last, partial block (rem rows) separately, restricted to what is valid: forward, its
    // tile L[last, last - k] has rem rows (K = rem; rows >= rem of M only reach solution entries nobody reads); backward, the
    // upper-copy block (last - k, last) has rem columns (N = rem, the other columns of M zero: they meet the zero entries of
    // the last solution block)
    const int64_t ld = c->ld_a;
    const int64_t nfull = c->n / NB, rem = c->n - nfull * NB;
    GemmDesc g;
    g.M = NB; g.N = NB; g.K = NB;
    g.lda = NB; g.a_kmajor = false;
    g.ldb = ld; g.b_kmajor = true;
    g.alpha = 1.0; g.beta = 0.0; g.lower = false; g.prof_cls = prof_cls;
    g.batch_a = NB * NB;
    g.batch_b = NB + NB * ld;
    g.batch_c = g.batch_d = NB * NB;
    g.ldcin = g.ldd = NB;
    double* Mc = c->mchain[slot];
    if (fwd) {  // blocks k .. nblk - 1: W_b x L[b, b - k]
        g.A = c->dinv + (int64_t)k * NB * NB;
        g.B = c->A + (int64_t)k * NB;
        g.D = Mc + (int64_t)k * NB * NB;
        g.Cin = g.D;
        g.batch = nfull - k;
        if (g.batch >= 1) FR_TRY(launch_gemm(ctx, g));
        if (rem > 0 && nfull >= k) {
            g.batch = 1;
            g.K = rem;
            g.A = c->dinv + nfull * (NB * NB);
            g.B = c->A + nfull * NB + (nfull - k) * NB * ld;
            g.D = Mc + nfull * (NB * NB);
            g.Cin = g.D;
            FR_TRY(launch_gemm(ctx, g));
        }
    } else {  // blocks 0 .. nblk - 1 - k: W_b^T x (block (b, b + k) of the upper copy)
        g.A = c->dinvt;
        g.B = c->A + (int64_t)k * NB * ld;
        g.D = Mc;
        g.Cin = g.D;
        g.batch = nfull - k;
        if (g.batch >= 1) FR_TRY(launch_gemm(ctx, g));
        if (rem > 0 && nfull >= k) {
            const int64_t b = nfull - k;
            FR_HIP(ctx, hipMemsetAsync(Mc + b * (NB * NB), 0, sizeof(double) * NB * NB, ctx->ls));
            g.batch = 1;
            g.N = rem;
            g.A = c->dinvt + b * (NB * NB);
            g.B = c->A + b * NB + (b + k) * NB * ld;
            g.D = Mc + b * (NB * NB);
            g.Cin = g.D;
            FR_TRY(launch_gemm(ctx, g));
        }
    }
    c->mchain_gen[slot] = c->gen;
    c->mchain_n[slot] = c->n;
    return FR_OK;
}

// B (n x m, device, m >= 2: column groups of 16) <- L^-1 B (fwd) or L^-T B.  One launch (+ a memset of the flags; the backward sweep also
// builds the transposed copy when the factor changed since it was last built).
int launch_trsm_narrow(fr_ctx* ctx, const fr_chol* cc, double* B, int64_t m, int64_t ldb, bool fwd, int prof_cls)
{
    fr_chol* c = const_cast<fr_chol*>(cc);  // the transposed copy is a cache: logically const
    const int64_t n = c->n;
    if (n <= 0 || m <= 0) return FR_OK;
    // 16 right-hand sides per column group
    const int nq = 1;
    const int MRT = 16 * nq;
    const int ngroups = (int)((m + MRT - 1) / MRT);
    if (ngroups > 65535) return set_err(ctx, FR_INVALID_ARGUMENT, "narrow solve: too many right-hand sides");
    const int nblk = (int)((n + NB - 1) / NB);
    FR_TRY(ensure_status_word(ctx));
    if (!fwd) FR_TRY(ensure_transposed(ctx, c));
    const size_t bytes = (sizeof(double) * (size_t)nblk * NB * MRT + sizeof(int) * (size_t)(nblk + 1)) * (size_t)ngroups + 64;
    if (ctx->trsmn_buf_cap < bytes) {
        if (ctx->trsmn_buf) {
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipFree(ctx->trsmn_buf);
            ctx->trsmn_buf = nullptr;
            ctx->trsmn_buf_cap = 0;
        }
        FR_HIP(ctx, dev_malloc(ctx, &ctx->trsmn_buf, bytes));
        ctx->trsmn_buf_cap = bytes;
    }
    TrsmnArgs a;
    a.A = c->A;
    a.ld = c->ld_a;
    a.n = n;
    a.inv = fwd ? c->dinv : c->dinvt;
    a.mchain = nullptr;
    if (ngroups == 1) {
        FR_TRY(ensure_chain_products(ctx, c, fwd, 1, prof_cls));
        a.mchain = c->mchain[fwd ? 0 : 1];
    }
    a.B = B;
    a.ldb = ldb;
    a.m = (int)m;
    a.xg = (double*)ctx->trsmn_buf;
    a.flags = (int*)((char*)ctx->trsmn_buf + sizeof(double) * (size_t)nblk * NB * MRT * (size_t)ngroups);
    a.tickets = (unsigned*)(a.flags + (size_t)nblk * (size_t)ngroups);
    a.ngroups = ngroups;
    a.status = ctx->dev_status;
    a.nblk = nblk;
    int G = nblk < ctx->num_cus ? nblk : ctx->num_cus;  // (two workgroups fit a CU: two column groups' chains side by side)
    if (ctx->test_max_wgs > 0 && G > ctx->test_max_wgs) G = ctx->test_max_wgs;
    a.bwd = fwd ? 0 : 1;
    FR_HIP(ctx, hipMemsetAsync(a.flags, 0, sizeof(int) * (size_t)(nblk + 1) * (size_t)ngroups, ctx->ls));
    if (ngroups == 1)  // the single-group kernel awaits the neighbour's block on the payload itself: fill it with the sentinel
        FR_HIP(ctx, hipMemsetD32Async((hipDeviceptr_t)a.xg, (int)SENT32, (size_t)nblk * NB * MR * 2, ctx->ls));
    ProfScope ps(ctx, prof_cls, (double)n * (double)n * (double)m, 4.0 * (double)n * (double)n);
    if (ngroups >= 2)
        hipLaunchKernelGGL(trsm_narrow_half_kernel<1>, dim3((unsigned)G, (unsigned)ngroups), dim3(NTH), 0, ctx->ls, a);
    else
    {
        if (!ctx->trsmn_lds_set) {  // per context (= per device): > 64 KiB of dynamic LDS needs the attribute
            FR_TRY(set_dyn_lds(ctx, (const void*)trsm_narrow_kernel, (int)TRSMN_LDS));
            ctx->trsmn_lds_set = true;
        }
        hipLaunchKernelGGL(trsm_narrow_kernel, dim3((unsigned)G, (unsigned)ngroups), dim3(NTH), TRSMN_LDS, ctx->ls, a);
    }
    FR_HIP(ctx, hipGetLastError());
    ctx->persistent_pending = true;
    if (ctx->test_force_timeout) {  // test hook: behave as if a hand-off of this launch had timed out
        FR_HIP(ctx, hipStreamSynchronize(ctx->ls));
        ((volatile unsigned*)ctx->host_status)[0] = 1u;
    }
    return FR_OK;
}

}  // namespace fr
