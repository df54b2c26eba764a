// trsv.hip -- K8: triangular solves with ONE right-hand side (L x = b, L^T x = b) as one persistent launch per direction.
//
// Replaces nalgebra's solve_lower_triangular / ad_solve_lower_triangular on a single column: K^-1 y inside predict (with
// predict_assoc = 1) and the cached alpha, likelihood (mod.rs:203), and every predict / predict_variance of ONE query
// point (mod.rs:235, 260-263) -- the Bayesian-optimisation inner loop of readme.md:7.
//
// The work is HBM-bound: the lower triangle of L is read exactly once (8 n^2 / 2 bytes: 4.3 GB at n = 32768).  Round 1
// issued it as a recursion of ~127 dependent launches per direction (10 ms at n = 32768 against a floor of 0.7 ms).  Here
// the solve is ONE launch: the 128-row blocks are dealt to the workgroups (one per CU); block r's owner streams its row of
// tiles L[r, c] (forward) or its column of tiles L[i, r] (backward) while the solution blocks x_c it needs become
// available, finishes with the explicit inverse of the diagonal block (fr_chol::dinv) and publishes x_r.  The only
// serial part is the hand-off of x_r from owner to owner, which uses the data-is-the-flag form of
// cdna_hip_programming.md Guideline 16 (R2): every double travels as two 8-byte {epoch, half} granules written by
// write-through (sc1) stores and polled with relaxed agent-scope loads -- no fence, no separate flag, placement
// independent.  The tile that a block needs LAST (the one next to the diagonal) and its inverse block are already in
// registers when the hand-off arrives, so a step of the chain costs one poll + 2 x 64 FMAs per lane + two small reductions.
//
// Deterministic: every sum has a fixed order (no atomics on data).
//
// Forward progress.  A workgroup CLAIMS its block with an atomic ticket when it starts (and again after every block it
// finishes): block index = order of arrival (forward: ticket i -> block i, backward: ticket i -> block nblk - 1 - i).  A block
// only ever waits for blocks with LOWER tickets, i.e. for workgroups that are already running -- whatever part of the grid
// the dispatcher has not placed yet (other kernels holding CUs, a second context, a CU mask) cannot be waited on, so the
// launch needs no co-residency guarantee.  (Round 2 dealt block r to workgroup r % G: with more blocks than resident
// workgroups a workgroup's second block waited for first blocks of workgroups that were never dispatched.)
// Every spin is still bounded (wall-clock timeout -> status word in host-visible memory); the entry point then repeats the
// operation on the recursive GEMM path (solve_retry in fr_internal.hpp) instead of failing.
#include "fr_internal.hpp"

namespace fr {

constexpr int TB = 128;  // block = the inverse-block size of the factor
constexpr int NT = 512;  // threads: 8 waves, two per SIMD -- VALU code can only name 256 VGPRs, and a lane keeps two tiles in flight
constexpr int NW = NT / 64;
constexpr int NC = TB / NW;  // columns of a tile per wave (16): a lane holds rows 2 rp, 2 rp + 1 of them = 32 doubles

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) unsigned gu32;

struct TrsvArgs {
    const double* L;
    int64_t ld, n;
    const double* dinv;  // block b at dinv + b * TB * TB, ld TB
    double* b;           // right-hand side in, solution out
    u64* gran;           // 2 granules per entry of x: [2 i] = {epoch, low word}, [2 i + 1] = {epoch, high word}
    unsigned* ticket;    // block claim counter, zeroed with the granules before every launch
    unsigned* status;    // host-visible: [0] != 0 after a timed-out wait
    int nblk;
    unsigned epoch;
};

// The next unclaimed block of the sweep, or -1 when none is left (uniform; the barrier also separates the LDS use of two
// consecutive blocks of this workgroup).
__device__ __forceinline__ int claim_block(unsigned* ticket, int nblk, int* slot)
{
    if (threadIdx.x == 0) {
        const unsigned i = __hip_atomic_fetch_add((gu32*)ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *slot = i < (unsigned)nblk ? (int)i : -1;
    }
    __syncthreads();
    const int r = __builtin_amdgcn_readfirstlane(*slot);
    return r;
}

__device__ __forceinline__ u64 gran_load(const u64* p)
{
    return __hip_atomic_load((gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void gran_store(u64* p, unsigned epoch, unsigned v)
{
    __hip_atomic_store((gu64*)p, ((u64)epoch << 32) | (u64)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Block `blk` of the solution: TB doubles = 2 TB granules, one per thread of the first four waves.  The granule of the NEXT
// dependency is requested while the current one is multiplied (gran_request; nothing to see first: the data is the flag), so
// a block that is behind the chain pays no round trip per dependency; a granule that has not arrived is polled with two
// samples in flight (half a round trip between samples: scripts/handoff_probe.hip).  A wait that times out raises the
// host-visible status word and goes on with whatever it has -- every later wait then gives up within a few polls, the
// launch drains and the entry point repeats the solve on the recursive path (solve_retry): no vote, no early exit.
__device__ __forceinline__ u64 gran_request(const TrsvArgs& a, int blk, int t)
{
    return t < 2 * TB ? gran_load(a.gran + (int64_t)blk * (2 * TB) + t) : 0;
}
__device__ __forceinline__ void wait_block(const TrsvArgs& a, int blk, double* xs, int t, u64 g)
{
    if (t < 2 * TB) {
        if ((unsigned)(g >> 32) != a.epoch) {
            const u64* p = a.gran + (int64_t)blk * (2 * TB) + t;
            const u64 t0 = wall_clock64();  // 100 MHz
            unsigned spins = 0;
            u64 g1 = gran_load(p);
            for (;;) {
                g = gran_load(p);
                if ((unsigned)(g1 >> 32) == a.epoch) {
                    g = g1;
                    break;
                }
                g1 = gran_load(p);
                if ((unsigned)(g >> 32) == a.epoch) break;
                if ((++spins & 127u) == 0) {
                    const bool dead = __hip_atomic_load((gu32*)a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
                    if (dead || wall_clock64() - t0 > 300000000ull) {  // 3 s
                        __hip_atomic_store((gu32*)a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                }
            }
        }
        // lanes 2 i / 2 i + 1 hold the low / high word of double i
        const unsigned mine = (unsigned)g;
        const unsigned other = (unsigned)__shfl_xor((int)mine, 1, 64);
        if ((t & 1) == 0) xs[t >> 1] = __hiloint2double((int)other, (int)mine);
    }
    __syncthreads();
}

__device__ __forceinline__ void publish_entry(const TrsvArgs& a, int64_t i, double v)
{
    gran_store(a.gran + 2 * i, a.epoch, (unsigned)__double2loint(v));
    gran_store(a.gran + 2 * i + 1, a.epoch, (unsigned)__double2hiint(v));
}

// One 128 x 128 operand block in registers: a lane holds rows 2 rp, 2 rp + 1 of the NC columns of its wave's column group
// (2 NC doubles), loaded 16 bytes wide and coalesced along the rows (1 KiB per wave-instruction).  The same routine serves
// the tiles of L (column stride ld) and the inverse of a diagonal block (column stride TB): one load site per register
// buffer keeps the address arithmetic -- and with it the register pressure -- of the unrolled loads in one place.
struct Operand {
    const double* base;  // element (0, 0) of the block
    int64_t stride;      // column stride
    int rows, cols;      // valid extent (entries outside read as zero); 128 x 128 for every block but the last
};

__device__ __forceinline__ Operand tile_operand(const TrsvArgs& a, int rb, int cb)
{
    Operand o;
    o.base = a.L + (int64_t)rb * TB + (int64_t)cb * TB * a.ld;
    o.stride = a.ld;
    const int64_t rr = a.n - (int64_t)rb * TB;
    o.rows = rr < TB ? (int)rr : TB;
    o.cols = TB;  // cb < rb: the column block is complete
    return o;
}

__device__ __forceinline__ Operand inv_operand(const TrsvArgs& a, int b)
{
    Operand o;
    o.base = a.dinv + (int64_t)b * (TB * TB);
    o.stride = TB;
    const int64_t rr = a.n - (int64_t)b * TB;
    o.rows = o.cols = rr < TB ? (int)rr : TB;
    return o;
}

__device__ __forceinline__ void load_operand(const Operand& o, int rp, int cg, double2 (&buf)[NC])
{
    // The loads are unconditional: the factor's leading dimension is a multiple of 128 (chol_alloc_buffers) and an inverse
    // block is a full 128 x 128 slot, so every address of a partial block exists; what lies outside its valid extent is
    // replaced by zeros afterwards (selects, no branches, no second set of addresses).
    // address = wave-uniform part (block, column: scalar registers) + the lane's row offset (one 32-bit VGPR shared by all
    // NC loads): no per-load 64-bit address registers
    const double* ub = o.base + (int64_t)(cg * NC) * o.stride;
    const unsigned lane_off = 2u * (unsigned)rp;
#pragma unroll
    for (int k = 0; k < NC; ++k) buf[k] = *reinterpret_cast<const double2*>(ub + (int64_t)k * o.stride + lane_off);
    if (o.rows < TB || o.cols < TB) {  // uniform per block: only the last block of a factor whose size is not a multiple of 128
        const bool okx = 2 * rp < o.rows, oky = 2 * rp + 1 < o.rows;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const bool cok = cg * NC + k < o.cols;
            buf[k].x = (cok && okx) ? buf[k].x : 0.0;
            buf[k].y = (cok && oky) ? buf[k].y : 0.0;
        }
    }
}

// ---- forward: L x = b ---------------------------------------------------------------------------------------------
// acc(rows 2 rp, 2 rp + 1) += tile[:, cols of this wave] . x[cols of this wave]
__device__ __forceinline__ void fwd_fma(const double2 (&buf)[NC], const double* xs, int cg, double& acc0, double& acc1)
{
    const double2* xv = reinterpret_cast<const double2*>(xs + cg * NC);
#pragma unroll
    for (int k = 0; k < NC / 2; ++k) {
        const double2 x = xv[k];  // same address in every lane: LDS broadcast
        acc0 = __builtin_fma(buf[2 * k].x, x.x, acc0);
        acc1 = __builtin_fma(buf[2 * k].y, x.x, acc1);
        acc0 = __builtin_fma(buf[2 * k + 1].x, x.y, acc0);
        acc1 = __builtin_fma(buf[2 * k + 1].y, x.y, acc1);
    }
}

// The inverse of the block's diagonal block waits in LDS (128 KiB, dynamic) from the start of the block: it is staged once,
// off the chain, through the tile registers, and the closing product reads it back in the register layout of a tile.
struct Shared {
    double xs[2][TB];
    double part[NW][TB];
    double tv[TB];
};
constexpr size_t TRSV_LDS = sizeof(double) * TB * TB;

__device__ __forceinline__ void stage_inverse(const TrsvArgs& a, int b, double* wl, int rp, int cg, double2 (&buf)[NC])
{
    load_operand(inv_operand(a, b), rp, cg, buf);
#pragma unroll
    for (int k = 0; k < NC; ++k) *reinterpret_cast<double2*>(wl + (cg * NC + k) * TB + 2 * rp) = buf[k];
}
__device__ __forceinline__ void fetch_inverse(const double* wl, int rp, int cg, double2 (&buf)[NC])
{
#pragma unroll
    for (int k = 0; k < NC; ++k) buf[k] = *reinterpret_cast<const double2*>(wl + (cg * NC + k) * TB + 2 * rp);
}

__device__ __forceinline__ double sum_parts(const Shared& s, int t)
{
    return ((s.part[0][t] + s.part[1][t]) + (s.part[2][t] + s.part[3][t])) +
           ((s.part[4][t] + s.part[5][t]) + (s.part[6][t] + s.part[7][t]));
}

__global__ __launch_bounds__(NT, 2) void trsv_fwd_kernel(const TrsvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double wl[];  // TB x TB inverse block
    __shared__ Shared s;
    __shared__ int claim_slot;
    const int t = threadIdx.x, rp = t & 63, cg = __builtin_amdgcn_readfirstlane(t >> 6);  // cg: wave-uniform
#pragma nounroll
    for (;;) {
        const int r = claim_block(a.ticket, a.nblk, &claim_slot);
        if (r < 0) return;
        const int64_t r0 = (int64_t)r * TB;
        double acc0 = 0.0, acc1 = 0.0;
        double2 A[NC], B[NC];
        stage_inverse(a, r, wl, rp, cg, A);
        // tiles L[r, q], q = 0 .. r - 1, through two register buffers: while tile q is consumed, tile q + 1 is in flight --
        // the loads do not depend on the hand-offs, only the FMAs do
        u64 g = 0;
        if (r > 0) {
            g = gran_request(a, 0, t);
            load_operand(tile_operand(a, r, 0), rp, cg, A);
        }
#pragma nounroll
        for (int q = 0; q < r; q += 2) {
            if (q + 1 < r) load_operand(tile_operand(a, r, q + 1), rp, cg, B);
            wait_block(a, q, s.xs[0], t, g);
            if (q + 1 < r) g = gran_request(a, q + 1, t);
            fwd_fma(A, s.xs[0], cg, acc0, acc1);
            if (q + 1 >= r) break;
            if (q + 2 < r) load_operand(tile_operand(a, r, q + 2), rp, cg, A);
            wait_block(a, q + 1, s.xs[1], t, g);
            if (q + 2 < r) g = gran_request(a, q + 2, t);
            fwd_fma(B, s.xs[1], cg, acc0, acc1);
        }
        // x_r = W_r (b_r - acc)
        s.part[cg][2 * rp] = acc0;
        s.part[cg][2 * rp + 1] = acc1;
        __syncthreads();  // also: the staged inverse is complete
        if (t < TB) s.tv[t] = (r0 + t < a.n) ? a.b[r0 + t] - sum_parts(s, t) : 0.0;
        fetch_inverse(wl, rp, cg, A);
        __syncthreads();
        double q0 = 0.0, q1 = 0.0;
        fwd_fma(A, s.tv, cg, q0, q1);
        s.part[cg][2 * rp] = q0;
        s.part[cg][2 * rp + 1] = q1;
        __syncthreads();
        if (t < TB) {
            const double x = sum_parts(s, t);
            publish_entry(a, r0 + t, x);  // first: the next owner is waiting for it
            if (r0 + t < a.n) a.b[r0 + t] = x;
        }
        __syncthreads();  // wl / part / tv are reused by the next block of this workgroup
    }
}

// ---- backward: L^T x = b --------------------------------------------------------------------------------------------
// Block j needs every x_i below it: its owner streams the tiles L[i, j], i = nblk - 1 .. j + 1 (rows along the lanes, so the
// loads stay coalesced) and keeps per-lane partial sums for its 16 columns; the 64-lane reduction happens once per block.
__device__ __forceinline__ void bwd_fma(const double2 (&buf)[NC], double x0, double x1, double (&p)[NC])
{
#pragma unroll
    for (int k = 0; k < NC; ++k) p[k] = __builtin_fma(buf[k].y, x1, __builtin_fma(buf[k].x, x0, p[k]));
}

// ---- the backward kernel's LDS (dynamic) --------------------------------------------------------------------------------
// (1) W_j^T, packed: the rows of W_j in groups of 16 (one group per wave of the closing product), group g holding its
//     16 (g + 1) possibly non-zero columns: element (row 16 g + k, col) at WT_OFF(g) + k * 16 (g + 1) + col;  72 KiB.
// (2) the transposition scratch of the column sums: per wave 16 x RS doubles;  66 KiB.
constexpr int RS = 66;                                  // row stride of the scratch (doubles): 2-way conflicts at worst
constexpr int WT_ELEMS = 128 * (NW * (NW + 1) / 2) * 2;  // sum over g of 16 * 16 (g + 1) = 9216
constexpr size_t TRSV_BWD_LDS = sizeof(double) * (WT_ELEMS + NW * NC * RS);
__device__ __forceinline__ int wt_off(int g) { return 128 * g * (g + 1); }

__device__ __forceinline__ double dpp_quad_xor(double v, const int ctrl_is_xor2)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    int rl, rh;
    if (ctrl_is_xor2) {
        rl = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, true);  // quad_perm [2, 3, 0, 1]
        rh = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, true);
    } else {
        rl = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true);  // quad_perm [1, 0, 3, 2]
        rh = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true);
    }
    return __hiloint2double(rh, rl);
}

// Sum p[k] over the 64 lanes of the wave, for the wave's 16 columns k at once, through the wave's scratch: every lane
// writes its 16 partials (lanes along the fast axis), then lane l adds a quarter (l & 3) of column l >> 2, and two DPP
// exchanges inside the quad finish.  Returns, in every lane, the total of column (lane >> 2).  (The shuffle-based halving
// this replaces took 3.4 us per call -- 34 ds_bpermute per lane from eight waves at once -- two thirds of a chain step.)
__device__ __forceinline__ double wave_reduce_cols(const double (&p)[NC], int lane, double* red)
{
#pragma unroll
    for (int k = 0; k < NC; ++k) red[k * RS + lane] = p[k];
    // same wave writes and reads: the LDS operations of one wave complete in order
    const double2* src = reinterpret_cast<const double2*>(red + (lane >> 2) * RS + 16 * (lane & 3));
    double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const double2 v = src[i];
        acc0 += v.x;
        acc1 += v.y;
    }
    double v = acc0 + acc1;
    v += dpp_quad_xor(v, 0);
    v += dpp_quad_xor(v, 1);
    return v;
}

__global__ __launch_bounds__(NT, 2) void trsv_bwd_kernel(const TrsvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double dyn[];
    __shared__ Shared s;
    double* wt = dyn;
    const int t = threadIdx.x, rp = t & 63, cg = __builtin_amdgcn_readfirstlane(t >> 6);  // cg: wave-uniform
    double* red = dyn + WT_ELEMS + cg * (NC * RS);
    __shared__ int claim_slot;
#pragma nounroll
    for (;;) {
        const int jj = claim_block(a.ticket, a.nblk, &claim_slot);
        if (jj < 0) return;
        const int j = a.nblk - 1 - jj;
        const int last = a.nblk - 1;
        const int cnt = last - j;  // tiles L[last - q, j], q = 0 .. cnt - 1
        const int64_t j0 = (int64_t)j * TB;
        const int col = cg * NC + (rp >> 2);  // the column whose sum this lane ends up with
        const double bj = (j0 + col < a.n) ? a.b[j0 + col] : 0.0;  // early: not on the chain
        double p[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) p[k] = 0.0;
        double2 A[NC], B[NC];
        // stage W_j^T (packed) once per block, off the chain: the lane holds W[rows 2 rp, 2 rp + 1][cols 16 cg + k]
        load_operand(inv_operand(a, j), rp, cg, A);
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int c = cg * NC + k;
            const int r0 = 2 * rp, r1 = 2 * rp + 1;
            // (every slot of the packed groups is written: above the diagonal the inverse block holds explicit zeros)
            if (c < 16 * ((r0 >> 4) + 1)) wt[wt_off(r0 >> 4) + (r0 & 15) * 16 * ((r0 >> 4) + 1) + c] = A[k].x;
            if (c < 16 * ((r1 >> 4) + 1)) wt[wt_off(r1 >> 4) + (r1 & 15) * 16 * ((r1 >> 4) + 1) + c] = A[k].y;
        }
        u64 g = 0;
        if (cnt > 0) {
            g = gran_request(a, last, t);
            load_operand(tile_operand(a, last, j), rp, cg, A);
        }
#pragma nounroll
        for (int q = 0; q < cnt; q += 2) {
            if (q + 1 < cnt) load_operand(tile_operand(a, last - q - 1, j), rp, cg, B);
            wait_block(a, last - q, s.xs[0], t, g);
            if (q + 1 < cnt) g = gran_request(a, last - q - 1, t);
            {
                const double2 x = *reinterpret_cast<const double2*>(&s.xs[0][2 * rp]);
                bwd_fma(A, x.x, x.y, p);
            }
            if (q + 1 >= cnt) break;
            if (q + 2 < cnt) load_operand(tile_operand(a, last - q - 2, j), rp, cg, A);
            wait_block(a, last - q - 1, s.xs[1], t, g);
            if (q + 2 < cnt) g = gran_request(a, last - q - 2, t);
            {
                const double2 x = *reinterpret_cast<const double2*>(&s.xs[1][2 * rp]);
                bwd_fma(B, x.x, x.y, p);
            }
        }
        // t_j = b_j - u, u = column sums of the partial products
        const double u = wave_reduce_cols(p, rp, red);
        if ((rp & 3) == 0) s.tv[col] = bj - u;
        __syncthreads();  // tv complete (and the staged inverse)
        // x_j = W_j^T t_j: the lane owns the outputs 2 rp, 2 rp + 1, its wave the rows 16 cg .. 16 cg + 15 of W_j
        double x0 = 0.0, x1 = 0.0;
        if (2 * rp < 16 * (cg + 1)) {
            const double* wrow = wt + wt_off(cg) + 2 * rp;
            const double2* tv2 = reinterpret_cast<const double2*>(s.tv + cg * NC);
#pragma unroll
            for (int k = 0; k < NC / 2; ++k) {
                const double2 tt = tv2[k];
                const double2 w0 = *reinterpret_cast<const double2*>(wrow + (2 * k) * 16 * (cg + 1));
                const double2 w1 = *reinterpret_cast<const double2*>(wrow + (2 * k + 1) * 16 * (cg + 1));
                x0 = __builtin_fma(w0.x, tt.x, x0);
                x1 = __builtin_fma(w0.y, tt.x, x1);
                x0 = __builtin_fma(w1.x, tt.y, x0);
                x1 = __builtin_fma(w1.y, tt.y, x1);
            }
        }
        s.part[cg][2 * rp] = x0;
        s.part[cg][2 * rp + 1] = x1;
        __syncthreads();
        if (t < TB) {
            const double x = sum_parts(s, t);
            publish_entry(a, j0 + t, x);
            if (j0 + t < a.n) a.b[j0 + t] = x;
        }
        __syncthreads();  // wt / part / tv are reused by the next block of this workgroup
    }
}

// b (n entries, device) <- L^-1 b or L^-T b with the factor's 128-block inverses.  One launch (+ one memset of the
// hand-off granules).  Enqueued on ctx->ls.
int launch_trsv(fr_ctx* ctx, const fr_chol* c, double* b, bool fwd, int prof_cls)
{
    const int64_t n = c->n;
    if (n <= 0) return FR_OK;
    const int nblk = (int)((n + TB - 1) / TB);
    FR_TRY(ensure_status_word(ctx));
    const size_t gran_bytes = sizeof(u64) * 2 * (size_t)nblk * TB + 64;  // + the ticket word
    if (ctx->trsv_gran_cap < gran_bytes) {
        if (ctx->trsv_gran) {
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipFree(ctx->trsv_gran);
            ctx->trsv_gran = nullptr;
            ctx->trsv_gran_cap = 0;
        }
        FR_HIP(ctx, dev_malloc(ctx, &ctx->trsv_gran, gran_bytes));
        ctx->trsv_gran_cap = gran_bytes;
    }
    // tags are compared with a per-call epoch; zeroing the granules before EVERY launch keeps the protocol independent of
    // whatever a previous (possibly aborted) call left behind (Guideline 16, "re-initialise every call")
    FR_HIP(ctx, hipMemsetAsync(ctx->trsv_gran, 0, gran_bytes, ctx->ls));
    TrsvArgs a;
    a.L = c->A;
    a.ld = c->ld_a;
    a.n = n;
    a.dinv = c->dinv;
    a.b = b;
    a.gran = (u64*)ctx->trsv_gran;
    a.ticket = (unsigned*)((char*)ctx->trsv_gran + gran_bytes - 64);
    a.status = ctx->dev_status;
    a.nblk = nblk;
    a.epoch = 1u;
    // one workgroup per CU (> 64 KiB of LDS); blocks are claimed, so the grid is only a degree of parallelism
    int G = nblk < ctx->num_cus ? nblk : ctx->num_cus;
    if (ctx->test_max_wgs > 0 && G > ctx->test_max_wgs) G = ctx->test_max_wgs;
    if (!ctx->trsv_lds_set) {  // per context (= per device): > 64 KiB of dynamic LDS needs the attribute
        FR_TRY(set_dyn_lds(ctx, reinterpret_cast<const void*>(trsv_fwd_kernel), (int)TRSV_LDS));
        FR_TRY(set_dyn_lds(ctx, reinterpret_cast<const void*>(trsv_bwd_kernel), (int)TRSV_BWD_LDS));
        ctx->trsv_lds_set = true;
    }
    ProfScope ps(ctx, prof_cls, (double)n * (double)n, 4.0 * (double)n * (double)n);
    if (fwd)
        hipLaunchKernelGGL(trsv_fwd_kernel, dim3((unsigned)G), dim3(NT), TRSV_LDS, ctx->ls, a);
    else
        hipLaunchKernelGGL(trsv_bwd_kernel, dim3((unsigned)G), dim3(NT), TRSV_BWD_LDS, ctx->ls, a);
    FR_HIP(ctx, hipGetLastError());
    ctx->persistent_pending = true;
    if (ctx->test_force_timeout) {  // test hook: behave as if a hand-off of this launch had timed out
        FR_HIP(ctx, hipStreamSynchronize(ctx->ls));
        ((volatile unsigned*)ctx->host_status)[0] = 1u;
    }
    return FR_OK;
}

}  // namespace fr
