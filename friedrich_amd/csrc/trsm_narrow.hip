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
// Hand-off of a solution block (128 x 16 doubles): write-through (sc1) stores -> every wave drains -> barrier -> flag
// (tagged 8-byte granules as in trsv.hip, measured on this payload of 16 KiB: 2.43 -> 3.83 ms per direction at n = 32768);
// the consumer polls the flag and reads the block with sc1 loads (cdna_hip_programming.md Guideline 16, R1).  Every
// spin is bounded (handoff.hpp).
#include "fr_internal.hpp"
#include "handoff.hpp"

namespace fr {

typedef double d4n_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) double gdbl;

constexpr int NB = 128;   // block
constexpr int MR = 16;    // right-hand sides per launch (padded)
constexpr int NTH = 512;  // 8 waves x 16 rows

struct TrsmnArgs {
    const double* A;    // factor buffer: lower triangle L; strict upper off-diagonal blocks hold L^T (backward)
    int64_t ld, n;
    const double* inv;  // inverse blocks (forward) or transposed inverse blocks (backward)
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
#pragma unroll
        for (int u = 0; u < 16; ++u) buf[u] = (mok && 64 * H + 4 * u + lq < f.kcols) ? buf[u] : 0.0;
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

// solution block `blk` -> xs ([col][q]); false after a timeout.  ONE buffer: the barrier inside the wait also tells that
// every wave is done with the block before.
__device__ __forceinline__ bool fetch_block(const TrsmnArgs& a, int blk, double* xs, int t)
{
    if (!handoff_wait_ge<false>(a.flags + blk, 1, a.status)) return false;  // no acquire fence: sc1 stores, sc1 loads
    const double* src = a.xg + (int64_t)blk * (NB * MR);
#pragma unroll
    for (int i = 0; i < (NB * MR) / NTH; ++i)
        xs[t + NTH * i] = __hip_atomic_load((gdbl*)(src + t + NTH * i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    return true;
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

// ---- one column group (m <= 16) ------------------------------------------------------------------------------------------
// The solve is a chain of n / 128 hand-offs; what one step costs is what the whole solve costs.  Round 2's step was: flag poll,
// payload fetch, the product with the neighbouring tile (1.7 us: 64 MFMAs per SIMD), a transposition through LDS, the closing
// product with the inverse block (1.7 us), payload stores, their acknowledgement, the flag -- ~9 us.  Now:
//   * the last dependency of block r is its NEIGHBOUR x_{r-1}; everything that does not need it is done before it arrives:
//       t' = b_r - sum_{q < r-1} L[r, q] x_q  (as before, behind flags),   y' = W_r t',   M = W_r L[r, r-1]  (128^3 on the matrix
//     core, while the workgroup would otherwise wait: the tile goes through LDS in eight 16-column chunks, and the accumulator
//     layout of v_mfma_f64_16x16x4 is exactly the operand layout of the next product, so M never leaves the registers);
//   * when x_{r-1} arrives ONE product is left on the chain:  x_r = y' - M x_{r-1};
//   * x_{r-1} is awaited on the payload itself: the hand-off buffer is filled with a NaN bit pattern no computation produces
//     before every launch, a lane polls its four doubles until none is the pattern (8-byte stores are single-copy atomic, each
//     element is written once) -- no flag round trip, no acknowledgement wait on the producer's side.  Only the ONE workgroup
//     whose last dependency is pending polls this way; the others wait for flags as before (the flag is still published,
//     behind the acknowledged stores, off the critical path).
constexpr unsigned SENT32 = 0x7FF7A5A5u;  // both halves: hipMemsetD32 fills the payload with the (NaN) double 0x7FF7A5A57FF7A5A5
__device__ __forceinline__ bool is_sentinel(double v) { return (unsigned long long)__double_as_longlong(v) == 0x7FF7A5A57FF7A5A5ull; }

__device__ __forceinline__ void mma_block(const double (&buf)[32], const double* xs, int l15, int lq, d4n_t& acc)
{
#pragma unroll
    for (int u = 0; u < 32; ++u) {
        const double xf = xs[(4 * u + lq) * MR + l15];  // element (k = 4 u + lq, q = l15)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xf, buf[u], acc, 0, 0, 0);
    }
}

// the payload of block `blk`, awaited on the data itself -> xs ([col][q]); false after a timeout
__device__ __forceinline__ bool poll_payload(const TrsmnArgs& a, int blk, double* xs, int t)
{
    const double* src = a.xg + (int64_t)blk * (NB * MR);
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    unsigned pending = 0xFu;
    int ok = 1;
    unsigned long long t0 = 0;
    unsigned spins = 0;
    for (;;) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (pending & (1u << i)) {
                v[i] = __hip_atomic_load((gdbl*)(src + t + NTH * i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!is_sentinel(v[i])) pending &= ~(1u << i);
            }
        if (!pending) break;
        __builtin_amdgcn_s_sleep(1);
        if (spins == 0) t0 = wall_clock64();
        if ((++spins & 255u) == 0) {
            const bool dead = __hip_atomic_load((hgu32*)a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
            if (dead || wall_clock64() - t0 > HANDOFF_TIMEOUT_TICKS) {
                __hip_atomic_store((hgu32*)a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                ok = 0;
                break;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) xs[t + NTH * i] = v[i];
    return __builtin_amdgcn_readfirstlane(__syncthreads_and(ok)) != 0;
}

__global__ __launch_bounds__(NTH, 2) void trsm_narrow_kernel(const TrsmnArgs a0)
{
    __shared__ double xs[2][NB * MR];
    __shared__ double tv[NB * MR];
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int last = a0.nblk - 1;
    TrsmnArgs a = a0;  // (one column group: blockIdx.y == 0)
    __shared__ int claim_slot;
#pragma nounroll
    for (;;) {
        const int bi = claim_block(a.tickets, a.nblk, &claim_slot);
        if (bi < 0) return;
        const int blk = a.bwd ? last - bi : bi;
        const int cnt = a.bwd ? last - blk : blk;  // dependencies: blocks dep(0) .. dep(cnt - 1), the neighbour last
        const int64_t b0 = (int64_t)blk * NB;
        const int64_t row = b0 + 16 * w + l15;
        double bv[4];  // this lane's right-hand side entries: (row, q = lq + 4 i)
#pragma unroll
        for (int i = 0; i < 4; ++i) bv[i] = (row < a.n && lq + 4 * i < a.m) ? a.B[row + (int64_t)(lq + 4 * i) * a.ldb] : 0.0;
        // the inverse block's fragments stay in registers for the whole block
        double Wf[32];
        load_frag(item_frag(a, blk, 0, true), w, l15, lq, Wf);
        // ---- M = W L[blk, neighbour], off the chain: the tile through LDS in 16-column chunks [k][16]
        double Mf[32];
        if (cnt > 0) {
            const Frag f = item_frag(a, blk, a.bwd ? blk + 1 : blk - 1, false);
            const int kk = t & 127, cc = t >> 7;
            // address = one per-lane pointer + wave-uniform column offsets; the loads are unconditional (the addresses exist: see
            // load_frag), what lies outside the valid extent of a last, partial block is replaced by zeros afterwards
            const double* pl = f.base + kk + (int64_t)cc * f.stride;
            auto load_chunk = [&](int j, double (&reg)[4]) {
#pragma unroll
                for (int i = 0; i < 4; ++i) reg[i] = pl[(int64_t)(16 * j + 4 * i) * f.stride];
                if (f.mrows < NB || f.kcols < NB) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) reg[i] = (kk < f.mrows && 16 * j + cc + 4 * i < f.kcols) ? reg[i] : 0.0;
                }
            };
            double cr[4];
            load_chunk(0, cr);
#pragma unroll
            for (int i = 0; i < 4; ++i) xs[0][kk * MR + cc + 4 * i] = cr[i];
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j + 1 < 8) load_chunk(j + 1, cr);
                d4n_t am = {0.0, 0.0, 0.0, 0.0};
                mma_block(Wf, xs[j & 1], l15, lq, am);
#pragma unroll
                for (int i = 0; i < 4; ++i) Mf[4 * j + i] = am[i];  // accumulator (m = l15, n = 16 j + lq + 4 i) == fragment slot u = 4 j + i
                if (j + 1 < 8) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) xs[(j + 1) & 1][kk * MR + cc + 4 * i] = cr[i];
                }
                __syncthreads();
            }
        }
        // ---- the dependencies in front of the neighbour: tiles in QUARTER-tile register buffers (16 rows x 32 columns, 8 doubles
        //      per lane), the next quarter always in flight -- W and M hold 128 of the 256 registers
        d4n_t acc = {0.0, 0.0, 0.0, 0.0};
        if (cnt > 1) {
            double Q0[8], Q1[8];
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
            auto mma_quarter = [&](const double (&buf)[8], int Hq) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xs[0][(32 * Hq + 4 * u + lq) * MR + l15], buf[u], acc, 0, 0, 0);
            };
            load_quarter(item_frag(a, blk, a.bwd ? last : 0, false), 0, Q0);
#pragma nounroll
            for (int q = 0; q < cnt - 1; ++q) {
                const int dep = a.bwd ? last - q : q;
                const Frag f = item_frag(a, blk, dep, false);
                load_quarter(f, 1, Q1);
                // (the barrier inside the wait: every wave is done with the block before; the dependency next to the neighbour is
                // awaited on its payload as well -- its flag follows the payload by an acknowledgement round trip, and this
                // block has y' = W t' to form between the two arrivals)
                if (q == cnt - 2) {
                    __syncthreads();
                    if (!poll_payload(a, dep, xs[0], t)) return;
                } else if (!fetch_block(a, dep, xs[0], t))
                    return;
                mma_quarter(Q0, 0);
                load_quarter(f, 2, Q0);
                mma_quarter(Q1, 1);
                load_quarter(f, 3, Q1);
                mma_quarter(Q0, 2);
                if (q + 1 < cnt - 1) load_quarter(item_frag(a, blk, a.bwd ? last - q - 1 : q + 1, false), 0, Q0);
                mma_quarter(Q1, 3);
            }
        }
        // ---- y' = W (b - acc): everything that does not need the neighbour
#pragma unroll
        for (int i = 0; i < 4; ++i) tv[(16 * w + l15) * MR + lq + 4 * i] = bv[i] - acc[i];
        __syncthreads();
        d4n_t x = {0.0, 0.0, 0.0, 0.0};
        mma_block(Wf, tv, l15, lq, x);
        // ---- the neighbour, awaited on its payload; ONE product on the chain
        if (cnt > 0) {
            if (!poll_payload(a, a.bwd ? blk + 1 : blk - 1, xs[1], t)) return;
            d4n_t z = {0.0, 0.0, 0.0, 0.0};
            mma_block(Mf, xs[1], l15, lq, z);
#pragma unroll
            for (int i = 0; i < 4; ++i) x[i] -= z[i];
        }
        // publish (write-through: the next owner polls these very words), then the caller's copy; the flag for the workgroups
        // that take this block as an earlier dependency follows the acknowledged stores, off the chain
        double* dst = a.xg + (int64_t)blk * (NB * MR);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __hip_atomic_store((gdbl*)(dst + (16 * w + l15) * MR + lq + 4 * i), x[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (row < a.n && lq + 4 * i < a.m) a.B[row + (int64_t)(lq + 4 * i) * a.ldb] = x[i];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) __hip_atomic_store((hgi32*)(a.flags + blk), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();  // tv / xs are reused by the next block of this workgroup
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
    if (c->ut_gen == c->gen && c->dinvt) return FR_OK;
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
    return FR_OK;
}

// B (n x m, device, m >= 2: column groups of 16) <- L^-1 B (fwd) or L^-T B.  One launch (+ a memset of the flags; the backward sweep also
// builds the transposed copy when the factor changed since it was last built).
int launch_trsm_narrow(fr_ctx* ctx, const fr_chol* cc, double* B, int64_t m, int64_t ldb, bool fwd, int prof_cls)
{
    fr_chol* c = const_cast<fr_chol*>(cc);  // the transposed copy is a cache: logically const
    const int64_t n = c->n;
    if (n <= 0 || m <= 0) return FR_OK;
    // right-hand sides per column group: 16; 32 (the half-tile kernel with two MFMA tiles per workgroup) from narrow_pair_min
    // right-hand sides on
    // (measured, scripts/narrow_pair_ab.py: pairs pay from 128 right-hand sides on against a factor of 16384+ rows -- n = 32768:
    // m = 128 / 256 / 512 forward solve 8.1 / 14.7 / 27.8 -> 7.7 / 12.7 / 22.9 ms -- and cost up to 60 % on smaller solves)
    const int64_t pair_min = ctx->narrow_pair_min >= 0 ? ctx->narrow_pair_min : (n >= 12288 ? 128 : 0);
    const int nq = (pair_min > 0 && m >= pair_min) ? 2 : 1;
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
    if (nq == 1 && ngroups == 1)  // the single-group kernel awaits the neighbour's block on the payload itself: fill it with the sentinel
        FR_HIP(ctx, hipMemsetD32Async((hipDeviceptr_t)a.xg, (int)SENT32, (size_t)nblk * NB * MR * 2, ctx->ls));
    ProfScope ps(ctx, prof_cls, (double)n * (double)n * (double)m, 4.0 * (double)n * (double)n);
    if (nq == 2)
        hipLaunchKernelGGL(trsm_narrow_half_kernel<2>, dim3((unsigned)G, (unsigned)ngroups), dim3(NTH), 0, ctx->ls, a);
    else if (ngroups >= 2)
        hipLaunchKernelGGL(trsm_narrow_half_kernel<1>, dim3((unsigned)G, (unsigned)ngroups), dim3(NTH), 0, ctx->ls, a);
    else
        hipLaunchKernelGGL(trsm_narrow_kernel, dim3((unsigned)G, (unsigned)ngroups), dim3(NTH), 0, ctx->ls, a);
    FR_HIP(ctx, hipGetLastError());
    ctx->persistent_pending = true;
    if (ctx->test_force_timeout) {  // test hook: behave as if a hand-off of this launch had timed out
        FR_HIP(ctx, hipStreamSynchronize(ctx->ls));
        ((volatile unsigned*)ctx->host_status)[0] = 1u;
    }
    return FR_OK;
}

}  // namespace fr
