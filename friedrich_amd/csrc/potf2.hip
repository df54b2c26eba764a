// potf2.hip -- K4: Cholesky of one diagonal block (<= 128 x 128) with friedrich's pivot rules, fused with the
// explicit inverse of the factored block (consumed by the GEMM-recast triangular solves).
//
// Pivot rule = nalgebra 0.31.4 Cholesky::new_internal (called from src/algebra/mod.rs:83,:90 and
// src/gaussian_process/multivariate_normal.rs:57):
//   d > 0            -> sqrt(d)
//   otherwise (0, negative, NaN):
//       mode 1 (cholesky_epsilon = Some(sub), sub > 0) -> sqrt(sub), column index appended to the log
//       else                                           -> failure, first failing column recorded
//   mode 2 (add_rows / Cholesky::insert_column, algebra/mod.rs:124; Appendix A.3): plain sqrt(d), NaN and
//   division by zero propagate exactly as in the reference, nothing is recorded.
//   mode 3: the block already holds a factor; only its inverse is produced (serde upload, re-alignment).
// Column scaling (`col /= denom`) is a reciprocal multiply with one residual correction.
//
// The factorisation of a diagonal block is a chain of n dependent pivots: what matters is the latency of ONE step, and
// this kernel sits on the critical path of the look-ahead pipeline (N such steps per fit, next to a chip full of GEMM
// workgroups).  So the serial part runs inside ONE wavefront with no barrier and no LDS round trip:
//
//   * the 128 x 128 block is cut into 32 x 32 sub-blocks kept in LDS (lower triangle: 10 slots, 80 KiB, which still
//     fits beside one resident GEMM workgroup);
//   * F_b: wave 0 factors diagonal sub-block b wave-synchronously.  Lane i (< 32) owns row i of the sub-block in 32
//     registers; step j broadcasts the pivot and L(c, j) with v_readlane (scalar operands of the FMAs) -- ~120
//     instructions per step, fully unrolled.  The inverse rides in the dead columns (slot c < j holds X(i, c)), and
//     lanes 32..63 carry the 32 rows below the sub-block, whose triangular solve is the same instruction stream;
//   * the block row/column updates between two F steps are small 32^3 products spread over all 8 waves (LDS-broadcast
//     operands):  T: L_ib = A_ib X_bb^T,  U: A_ik -= L_ib L_kb^T,  and the inverse W = L^-1 by block elimination
//     (W_bc = X_bb W_bc;  W_ic -= L_ib W_bc;  W_ib = -L_ib X_bb), which overwrites the L sub-blocks once they are dead.
//
// Factor columns are stored to global memory as they are produced (fire and forget: barriers order LDS only).
#include "fr_internal.hpp"

namespace fr {

constexpr int PB = 128;  // largest block
constexpr int SB = 32;   // sub-block
constexpr int SBE = SB * SB;
constexpr int PT = 512;  // 8 waves; <= 128 VGPRs so that they fit beside ONE resident GEMM workgroup
constexpr int NSLOT = 10;
constexpr size_t POTF2_LDS = (size_t)NSLOT * SBE * sizeof(double);

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it would wait for every
// outstanding factor-column store to be acknowledged by memory (microseconds under GEMM load).
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// sqrt(d) and 1/sqrt(d) from ONE v_rsq_f64 seed and Newton steps (~15 dependent instructions).  The step's critical
// path is sqrt -> reciprocal -> column scaling; libm's sqrt + two IEEE divisions are ~100 dependent f64 instructions,
// most of a step.  Results agree with sqrt()/division to the last bit or 1 ulp (the oracle comparison bounds it).
__device__ __forceinline__ void sqrt_rsqrt(double d, double& p, double& ip)
{
    double r = __builtin_amdgcn_rsq(d);
    const double h = 0.5 * d;
    r = r * __builtin_fma(-h * r, r, 1.5);
    r = r * __builtin_fma(-h * r, r, 1.5);
    double q = d * r;
    q = __builtin_fma(0.5 * r, __builtin_fma(-q, q, d), q);  // sqrt(d), corrected
    r = r * __builtin_fma(-q, r, 2.0);                        // 1 / q
    const bool zero = (d == 0.0);  // plain-sqrt mode: sqrt(0) = 0, then the reference divides by zero
    p = zero ? 0.0 : q;
    ip = zero ? __builtin_inf() : r;
}

__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ int slot_of(int i, int k)
{
    return (i * (i + 1) / 2 + k) * SBE;
}

// 32^3 products of the update waves.  Lane (r, h) accumulates NC consecutive result columns of row r; the row operand
// A(r, k) is read from LDS as it is needed (a rolled k loop: with the row held in registers and the loop unrolled the
// instruction selector hoists every LDS read to the top and spills hundreds of registers).
//   acc[c] = sum_k A[r + 32 k] * B[(c0 + c) + 32 k]   ("A B^T": B is indexed [column of the result, k])
template <int NC>
__device__ __forceinline__ void prod_nt(const double* Arow, const double* B, int c0, double (&acc)[NC])
{
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = 0.0;
    const double* bp = B + c0;
#pragma unroll 2
    for (int k = 0; k < SB; ++k) {
        const double a = Arow[SB * k];
        const double2* b2 = reinterpret_cast<const double2*>(bp + SB * k);
#pragma unroll
        for (int c = 0; c < NC; c += 2) {
            const double2 v = b2[c >> 1];
            acc[c] = __builtin_fma(a, v.x, acc[c]);
            acc[c + 1] = __builtin_fma(a, v.y, acc[c + 1]);
        }
    }
}

//   acc[c] = sum_k A[r + 32 k] * B[k + 32 (c0 + c)]   ("A B")
template <int NC>
__device__ __forceinline__ void prod_nn(const double* Arow, const double* B, int c0, double (&acc)[NC])
{
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = 0.0;
    const double* bp = B + SB * c0;
#pragma unroll 2
    for (int k = 0; k < SB; k += 2) {
        const double a0 = Arow[SB * k], a1 = Arow[SB * (k + 1)];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const double2 v = *reinterpret_cast<const double2*>(bp + k + SB * c);
            acc[c] = __builtin_fma(a0, v.x, acc[c]);
            acc[c] = __builtin_fma(a1, v.y, acc[c]);
        }
    }
}

// ---- F_b: wave-synchronous factorisation of diagonal sub-block b (executed by wave 0 only) --------------------------
// lanes 0..31 : row (lane) of sub-block (b, b);  lanes 32..63 : row (lane - 32) of sub-block (b + 1, b) (ride-along solve)
struct FState {
    double* gptr;    // &A[row of this lane, current column]
    double* rslot;   // ride-along L sub-block in LDS
    int64_t lda, colbase;  // colbase: global column index of the sub-block's first column
    double p_exc, ip_exc;  // replacement pivot of the exception rule (NaN = failure)
    int lane, r, mode, ncols_ok;  // ncols_ok: columns j < ncols_ok lie inside the matrix
    bool lo, ride, grow_ok;
};

// The pivot of step J + 1 (sqrt and reciprocal of the next diagonal value) is a chain of ~13 dependent f64 operations.
// It is cut into stages that the step interleaves with its 30 independent rank-1 updates (two updates per stage,
// pinned with sched_barrier) so that the chain's latency disappears behind the update stream.
// Optimisation barrier: the value must be materialised in a VGPR at this point of the program.  Without it the
// instruction selector's scheduler sinks every rank-1 FMA down to the step that next reads the slot (a legal but
// pathological order: all multipliers and broadcast scalars of all steps stay live, hundreds of spills).
__device__ __forceinline__ void pin(double& x)
{
    asm volatile("" : "+v"(x));
}

struct PivotChain {
    double d, r, h, q, t;
};

template <bool M3, int K>
__device__ __forceinline__ void chain_stage(PivotChain& c)
{
    if constexpr (M3) {  // the block already holds the factor: p = d, ip = 1 / d (reciprocal seed + two Newton steps)
        if constexpr (K == 0) c.r = __builtin_amdgcn_rcp(c.d);
        if constexpr (K == 1) c.t = __builtin_fma(-c.d, c.r, 1.0);
        if constexpr (K == 2) c.r = __builtin_fma(c.r, c.t, c.r);
        if constexpr (K == 3) c.t = __builtin_fma(-c.d, c.r, 1.0);
        if constexpr (K == 4) c.r = __builtin_fma(c.r, c.t, c.r);
        if constexpr (K == 5) c.q = c.d;
        if constexpr (K == 0 || K == 2 || K == 4) pin(c.r);
        if constexpr (K == 1 || K == 3) pin(c.t);
    } else {  // same arithmetic as sqrt_rsqrt()
        if constexpr (K == 0) {
            c.r = __builtin_amdgcn_rsq(c.d);
            c.h = -0.5 * c.d;
        }
        if constexpr (K == 1) c.t = c.h * c.r;
        if constexpr (K == 2) c.t = __builtin_fma(c.t, c.r, 1.5);
        if constexpr (K == 3) c.r = c.r * c.t;
        if constexpr (K == 4) c.t = c.h * c.r;
        if constexpr (K == 5) c.t = __builtin_fma(c.t, c.r, 1.5);
        if constexpr (K == 6) c.r = c.r * c.t;
        if constexpr (K == 7) {
            c.q = c.d * c.r;
            c.h = 0.5 * c.r;
        }
        if constexpr (K == 8) c.t = __builtin_fma(-c.q, c.q, c.d);
        if constexpr (K == 9) c.q = __builtin_fma(c.h, c.t, c.q);
        if constexpr (K == 10) c.t = __builtin_fma(-c.q, c.r, 2.0);
        if constexpr (K == 11) c.r = c.r * c.t;
        if constexpr (K == 12) {
            const bool zero = (c.d == 0.0);  // plain-sqrt mode: sqrt(0) = 0, then the reference divides by zero
            c.q = zero ? 0.0 : c.q;
            c.r = zero ? __builtin_inf() : c.r;
        }
        if constexpr (K == 0 || K == 3 || K == 6 || K == 11 || K == 12) pin(c.r);
        if constexpr (K == 1 || K == 2 || K == 4 || K == 5 || K == 8 || K == 10) pin(c.t);
        if constexpr (K == 7 || K == 9 || K == 12) pin(c.q);
        if constexpr (K == 0 || K == 7) pin(c.h);
    }
}

// Pivot rule for a diagonal value that is 0, negative or NaN (any mode but plain-sqrt): the replacement pivot
// (sqrt(sub) and its reciprocal, or NaN = failure) is prepared once per launch and selected without a branch; the
// column is only noted in a bit mask.  The log in global memory is written after the unrolled steps: a memory access on
// a rare path inside them would make every step wait for the outstanding factor-column stores at the join.
__device__ __forceinline__ void pivot_select(const FState& st, double d, int j, double& p, double& ip, unsigned& excmask)
{
    const bool bad = !(st.mode == 2 || d > 0.0);
    p = bad ? st.p_exc : p;
    ip = bad ? st.ip_exc : ip;
    excmask |= bad ? (1u << j) : 0u;
}

// the u-th independent update of step J: slots J + 2 .. 31 (factor: A(i, c) -= L(i, J) L(c, J)), then slots 0 .. J - 1
// (inverse, riding in the dead columns: X(i, c) -= L(i, J) / p * X'(J, c))
template <bool M3, int J, int U>
__device__ __forceinline__ void step_update(double (&a)[SB], double l, double m)
{
    constexpr int NF = M3 ? 0 : ((SB - J - 2) > 0 ? (SB - J - 2) : 0);
    if constexpr (U < NF) {
        constexpr int c = J + 2 + U;
        const double s = readlane_f64(l, c);  // L(c, J)
        a[c] = __builtin_fma(-l, s, a[c]);
        pin(a[c]);
    } else if constexpr (U - NF < J) {
        constexpr int c = U - NF;
        const double s = readlane_f64(a[c], J);  // X'(J, c)
        a[c] = __builtin_fma(-m, s, a[c]);
        pin(a[c]);
    }
}

template <bool M3, int J, int U, int K>
__device__ __forceinline__ void step_interleave(double (&a)[SB], double l, double m, PivotChain& ch)
{
    constexpr int NUPD = (M3 ? 0 : ((SB - J - 2) > 0 ? (SB - J - 2) : 0)) + J;
    constexpr int NST = M3 ? 6 : 13;
    if constexpr (U < NUPD || K < NST) {
        if constexpr (K < NST && J + 1 < SB) chain_stage<M3, K>(ch);
        step_update<M3, J, U>(a, l, m);
        step_update<M3, J, U + 1>(a, l, m);
        __builtin_amdgcn_sched_barrier(0);
        step_interleave<M3, J, U + 2, K + 1>(a, l, m, ch);
    }
}

// one elimination step; J is a compile-time constant so that every register index and lane select is static
// (template recursion instead of `#pragma unroll`: the body is beyond clang's pragma-unroll budget)
template <bool M3, int J>
__device__ __forceinline__ void f_step(double (&a)[SB], FState& st, double p, double ip, unsigned& excmask)
{
    const int lane = st.lane;
    // column J: col /= denom as reciprocal multiply + one residual correction
    const double v = a[J];
    double q = v * ip;
    q = __builtin_fma(__builtin_fma(-q, p, v), ip, q);
    if constexpr (M3) q = v;
    // L(i, J); the ride-along lanes are all below.  Padding rows / columns (block smaller than 128) are forced to
    // stay the identity: 0 * inf = NaN would otherwise leak from an overflowing substituted factor into the log
    const double l = (lane > J && st.grow_ok && J < st.ncols_ok) ? q : 0.0;
    if constexpr (!M3) {
        if (lane >= J && st.grow_ok && J < st.ncols_ok) *st.gptr = (lane == J) ? p : q;
        st.gptr += st.lda;
        if (!st.lo && st.ride) st.rslot[st.r + SB * J] = q;
    }
    // inverse multiplier: the pivot row itself is scaled by 1/p, written as a - (1 - 1/p) a so that it is the same FMA
    const double m = (lane == J) ? (1.0 - ip) : (st.lo ? l * ip : 0.0);
    PivotChain ch;
    if constexpr (J + 1 < SB) {
        if constexpr (!M3) {
            const double s = readlane_f64(l, J + 1);
            a[J + 1] = __builtin_fma(-l, s, a[J + 1]);
            pin(a[J + 1]);
        }
        ch.d = readlane_f64(a[J + 1], J + 1);  // next pivot candidate (uniform)
    }
    __builtin_amdgcn_sched_barrier(0);
    step_interleave<M3, J, 0, 0>(a, l, m, ch);
    a[J] = (lane == J) ? ip : -m;  // X(i, J) = -L(i, J) / p below the diagonal, 0 above
    if constexpr (J + 1 < SB) {
        double pn = ch.q, ipn = ch.r;
        if constexpr (!M3) pivot_select(st, ch.d, J + 1, pn, ipn, excmask);
        f_step<M3, J + 1>(a, st, pn, ipn, excmask);
    }
}

template <bool M3>
__device__ __forceinline__ void factor_subblock(double* lds, int b, int nblk, int lane, double* __restrict__ A, int64_t lda,
                                                int n, int64_t col0, int mode, double sub, int64_t* __restrict__ info,
                                                bool want_inv)
{
    FState st;
    st.lane = lane;
    st.lo = lane < SB;
    st.r = lane & (SB - 1);
    st.ride = (b + 1 < nblk);
    double* dslot = lds + slot_of(b, b);
    st.rslot = lds + slot_of(st.ride ? b + 1 : b, b);
    st.gptr = A + (SB * b + lane) + (int64_t)(SB * b) * lda;
    st.grow_ok = SB * b + lane < n;
    st.ncols_ok = n - SB * b;
    st.lda = lda;
    st.colbase = col0 + SB * b;
    st.mode = mode;
    st.p_exc = __builtin_nan("");
    st.ip_exc = st.p_exc;
    const bool substitute = (mode == 1 && sub > 0.0);
    if (substitute) sqrt_rsqrt(sub, st.p_exc, st.ip_exc);
    double a[SB];
    {
        const double* src = st.lo ? dslot : st.rslot;
        const bool ok = st.lo || st.ride;
#pragma unroll
        for (int c = 0; c < SB; ++c) {
            const double v = src[st.r + SB * c];
            a[c] = ok ? v : 0.0;
        }
    }
    // first pivot
    unsigned excmask = 0;
    PivotChain ch;
    ch.d = readlane_f64(a[0], 0);
    double p, ip;
    if constexpr (M3) {
        p = ch.d;
        ip = 1.0 / ch.d;
    } else {
        sqrt_rsqrt(ch.d, p, ip);
        pivot_select(st, ch.d, 0, p, ip, excmask);
    }
    f_step<M3, 0>(a, st, p, ip, excmask);
    if (st.ncols_ok < SB) excmask &= (1u << (st.ncols_ok > 0 ? st.ncols_ok : 0)) - 1u;
    if (excmask != 0 && lane == 0) {  // the log: substituted columns in order, or the first failing column
        if (substitute) {
            int64_t q = info[1];
            for (int j = 0; j < SB; ++j)
                if (excmask & (1u << j)) info[3 + q++] = st.colbase + j;
            info[1] = q;
        } else if (info[0] == 0) {
            info[0] = 1 + st.colbase + (__builtin_ffs((int)excmask) - 1);
        }
    }
    if (want_inv && st.lo) {
#pragma unroll
        for (int c = 0; c < SB; ++c) dslot[st.r + SB * c] = (c <= st.r) ? a[c] : 0.0;
    }
}

__global__ __launch_bounds__(PT, 4) void potf2_kernel(double* __restrict__ A, int64_t lda, int n, int64_t col0, int mode,
                                                   double sub, double* __restrict__ inv, int64_t ldinv,
                                                   int64_t* __restrict__ info)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int w = t >> 6;  // wave, uniform
    const int r = lane & (SB - 1);
    const int h = lane >> 5;
    const int nblk = (n + SB - 1) / SB;
    const bool want_inv = inv != nullptr;
    const bool m3 = (mode == 3);

    // ---- load the lower triangle into the LDS slots; rows / columns >= n are padded with the identity
    {
        const int cg = t >> 5;  // 16 column groups, 2 columns each per slot
        for (int i = 0; i < nblk; ++i)
            for (int k = 0; k <= i; ++k) {
                double* s = lds + slot_of(i, k);
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    const int c = cg * 2 + cc;
                    const int gr = SB * i + r, gc = SB * k + c;
                    double v = (gr == gc) ? 1.0 : 0.0;
                    if (gr < n && gc < n) v = (gr >= gc) ? A[gr + (int64_t)gc * lda] : 0.0;
                    s[r + SB * c] = v;
                }
            }
    }
    lds_barrier();

    // Wave 0 is the panel wave (the serial pivot chain), waves 1..7 are the update waves (the 32^3 products); both sides
    // execute the same barriers per stage: A (F done) | B (T rows loaded) | C (T, (1) done) | D (U, (2) done) |
    // E1 ((3) rows loaded) | E2 ((3) done) -- the last two only when the inverse is wanted.
    if (w == 0) {
        for (int b = 0; b < nblk; ++b) {
            if (m3)
                factor_subblock<true>(lds, b, nblk, lane, A, lda, n, col0, mode, sub, info, want_inv);
            else
                factor_subblock<false>(lds, b, nblk, lane, A, lda, n, col0, mode, sub, info, want_inv);
            lds_barrier();  // A
            lds_barrier();  // B
            lds_barrier();  // C
            lds_barrier();  // D
            if (want_inv) {
                lds_barrier();  // E1
                lds_barrier();  // E2
            }
        }
    } else {
        const int u = w - 1;  // update wave 0..6
        for (int b = 0; b < nblk; ++b) {
            lds_barrier();  // A
            const double* Xbb = lds + slot_of(b, b);

            // ---- P1: T: L_ib = A_ib X_bb^T for i >= b + 2 (in place: every task finishes reading before anyone
            //          stores);  (1): W_bc = X_bb W_bc for c < b (in place too, but column-local to one task)
            {
                const int i = b + 2 + (u >> 1);
                const bool tact = !m3 && u < 4 && i < nblk;
                double* Tib = lds + slot_of(tact ? i : b, tact ? b : 0);
                const int tc0 = ((u & 1) * 2 + h) * 8;
                double tacc[8];
                if (tact) prod_nt<8>(Tib + r, Xbb, tc0, tacc);
                if (want_inv) {
                    // tasks of (1) go to the waves without a T task first
                    for (int task = (u + 3) % 7; task < 4 * b; task += 7) {
                        double* Wbc = lds + slot_of(b, task >> 2);
                        const int c0 = ((task & 3) * 2 + h) * 4;
                        double acc[4];
                        prod_nn<4>(Xbb + r, Wbc, c0, acc);
#pragma unroll
                        for (int c = 0; c < 4; ++c) Wbc[r + SB * (c0 + c)] = acc[c];
                    }
                }
                lds_barrier();  // B
                if (tact) {
                    const int gr = SB * i + r;
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        Tib[r + SB * (tc0 + c)] = tacc[c];
                        const int gc = SB * b + tc0 + c;
                        if (gr < n && gc < n) A[gr + (int64_t)gc * lda] = tacc[c];
                    }
                }
            }
            lds_barrier();  // C

            // ---- P2: U: A_ik -= L_ib L_kb^T (b < k <= i);  (2): W_ic -= L_ib W_bc (i > b, c < b)
            {
                const int rem = nblk - b - 1;
                const int nU = m3 ? 0 : rem * (rem + 1) / 2;
                const int n2 = want_inv ? rem * b : 0;
                for (int task = u; task < 4 * (nU + n2); task += 7) {
                    const int pidx = task >> 2;
                    const int c0 = ((task & 3) * 2 + h) * 4;
                    double acc[4];
                    double* C;
                    if (pidx < nU) {
                        // pairs in the order (1,1) (2,1) (2,2) (3,1) (3,2) (3,3), relative to b
                        const int ri = (pidx >= 3) ? 3 : ((pidx >= 1) ? 2 : 1);
                        const int rk = pidx - ri * (ri - 1) / 2 + 1;
                        const int i = b + ri, k = b + rk;
                        prod_nt<4>(lds + slot_of(i, b) + r, lds + slot_of(k, b), c0, acc);
                        C = lds + slot_of(i, k);
                    } else {
                        const int q = pidx - nU;
                        const int i = b + 1 + q / b, cb = q % b;
                        prod_nn<4>(lds + slot_of(i, b) + r, lds + slot_of(b, cb), c0, acc);
                        C = lds + slot_of(i, cb);
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) C[r + SB * (c0 + c)] -= acc[c];
                }
            }
            lds_barrier();  // D

            // ---- P3: (3): W_ib = -L_ib X_bb, in place over the now dead L_ib (all reads, a barrier, then the stores)
            if (want_inv) {
                const int i = b + 1 + (u >> 1);
                const bool act = (u < 6) && i < nblk;
                double* Wib = lds + slot_of(act ? i : b, act ? b : 0);
                const int c0 = ((u & 1) * 2 + h) * 8;
                double acc[8];
                if (act) prod_nn<8>(Wib + r, Xbb, c0, acc);
                lds_barrier();  // E1
                if (act) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) Wib[r + SB * (c0 + c)] = -acc[c];
                }
                lds_barrier();  // E2
            }
        }
    }

    // ---- store the inverse (lower blocks from LDS, zeros above)
    if (want_inv) {
        const int cg = t >> 5;
        for (int i = 0; i < nblk; ++i)
            for (int k = 0; k < nblk; ++k) {
                const double* s = lds + slot_of(i, k <= i ? k : 0);
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    const int c = cg * 2 + cc;
                    const int gr = SB * i + r, gc = SB * k + c;
                    if (gr < n && gc < n) inv[gr + (int64_t)gc * ldinv] = (k <= i) ? s[r + SB * c] : 0.0;
                }
            }
    }
}

int launch_potf2(fr_ctx* ctx, double* A, int64_t lda, int64_t nbk, int64_t col0, int mode, double sub, double* inv,
                 int64_t ldinv, int64_t* info)
{
    if (nbk <= 0) return FR_OK;
    if (nbk > PB) return set_err(ctx, FR_INVALID_ARGUMENT, "potf2 block too large");
    static bool attr_set = false;
    if (!attr_set) {
        FR_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(potf2_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)POTF2_LDS));
        attr_set = true;
    }
    ProfScope ps(ctx, FR_PROF_POTF2, (double)nbk * nbk * nbk * (2.0 / 3.0), (double)nbk * nbk * 8.0 * 3.0);
    hipLaunchKernelGGL(potf2_kernel, dim3(1), dim3(PT), POTF2_LDS, ctx->ls, A, lda, (int)nbk, col0, mode, sub, inv, ldinv,
                       info);
    FR_HIP(ctx, hipGetLastError());
    return FR_OK;
}

}  // namespace fr
