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
// Column scaling (`col /= denom`) multiplies by 1 / sqrt(d) (v_rsq_f64 seed + two Newton steps, accurate to about an ulp);
// the stored diagonal sqrt(d) = d / sqrt(d) takes one Heron correction.
//
// The factorisation of a diagonal block is a chain of n dependent pivots: what matters is the latency of ONE step, and
// this kernel sits on the critical path of the look-ahead pipeline (N such steps per fit, next to a chip full of GEMM
// workgroups).  So the serial part runs inside ONE wavefront with no barrier:
//
//   * the 128 x 128 block is cut into 32 x 32 sub-blocks kept in LDS (lower triangle: 10 slots, 80 KiB, which still
//     fits beside one resident GEMM workgroup);
//   * F_b: wave 0 factors diagonal sub-block b wave-synchronously.  Row i of the sub-block is split over lanes i and
//     i + 32 (16 register slots each).  Step j writes the scaled column to LDS (one ds_write into the sub-block's own
//     slot, dead once it is in registers: the "image" of L_bb, reciprocal pivots on its diagonal), bumps a column counter
//     and reads L(c, j) back as broadcasts (operands of the rank-1 FMAs), all in flight at once.  The next pivot does
//     not wait for that round trip: lane j + 1 updates its own diagonal element from its own L(j + 1, j), one v_readlane
//     publishes it, and the stages of the reciprocal-pivot chain are issued between the other parts of the step;
//   * the triangular solves of the stage run CONCURRENTLY on the update waves, one column behind the pivot chain (they poll
//     the LDS counter): T: L_ib = A_ib L_bb^-T for the sub-blocks below, X_bb = L_bb^-1 (the same routine applied to e_c),
//     and the inverse's row solve W_bc = L_bb^-1 W_bc;
//   * the remaining updates are 32^3 products as MFMA 16 x 16 tiles over the update waves:  U: A_ik -= L_ib L_kb^T, and the
//     block elimination of the inverse (W_ic -= L_ib W_bc;  W_ib = -L_ib X_bb), which overwrites the L sub-blocks once
//     they are dead.  The inverse is kept transposed in LDS so that every product is an "N T" contraction.
//
// Factor columns are stored to global memory as they are produced (fire and forget: barriers order LDS only).
// Alone the kernel takes ~60 us for a full block (F: 4 x 7.8 us); next to a GEMM workgroup ~4x that: every dependent
// f64 operation then queues behind the neighbour's MFMAs and an LDS round trip costs ~900 cycles instead of ~150
// (scripts/contention_probe.hip, scripts/potf2_bench_phases.hip).
#include "fr_internal.hpp"
#include "handoff.hpp"

namespace fr {

constexpr int PB = 128;  // largest block
constexpr int SB = 32;   // sub-block
constexpr int SBE = SB * SB;
constexpr int PT = 512;  // 8 waves; <= 128 VGPRs so that they fit beside ONE resident GEMM workgroup
constexpr int NSLOT = 10;
#ifndef FR_K4_EXP
#define FR_K4_EXP 0
#endif
constexpr int K4X = FR_K4_EXP;  // developer experiments of scripts/mk_potf2_phases.py (0 in the product build)
constexpr size_t POTF2_LDS = (size_t)NSLOT * SBE * sizeof(double) + 16 + SB * sizeof(double);  // + the column counter of the panel wave + 32 dummy words (see f_step)

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

// Optimisation barrier: the value must be materialised in a VGPR at this point of the program.  Without it the
// instruction selector's scheduler sinks every rank-1 FMA down to the step that next reads the slot (a legal but
// pathological order: all multipliers and broadcast values of all steps stay live, hundreds of spills).
__device__ __forceinline__ void pin(double& x)
{
    asm volatile("" : "+v"(x));
}

// ---- split-row layout of the wave-synchronous routines -------------------------------------------------------------
// A vector of 32 (a row of a sub-block, or a column of the inverse) is held by TWO lanes: lane (r, h) = r + 32 h owns
// the 16 slots c = 16 h + k.  64 lanes work on one 32 x 32 sub-block, a lane needs 32 VGPRs for its slots and 32 for a
// whole column of broadcasts, so every LDS read of a step is in flight at once (a ~130-cycle latency paid once per
// step, behind the pivot chain, instead of once per FMA).
constexpr int HB = 16;  // slots per lane
typedef double d4_t __attribute__((ext_vector_type(4)));

// value of the same row in half H, delivered to both halves (one v_permlane32_swap per dword)
template <int H>
__device__ __forceinline__ double bcast_half(double x)
{
    const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)rh[H], (int)rl[H]);
}

// broadcasts of one column J of the factor image: L(16 h + k, J) for the lane's 16 slots, as 8 aligned pairs
struct ColBcast {
    double2 v[HB / 2];
};

constexpr int first_live_slot(int J)  // slots k >= this are still live in some half after column J
{
    return J < HB ? 0 : (J % HB) + 1;
}

template <int J, int P>
__device__ __forceinline__ void col_load(ColBcast& cb, const double* bufh)
{
    if constexpr (P < HB / 2) {
        if constexpr (2 * P + 1 >= first_live_slot(J)) cb.v[P] = *reinterpret_cast<const double2*>(bufh + 2 * P + SB * J);
        col_load<J, P + 1>(cb, bufh);
    }
}

// ---- F_b: wave-synchronous factorisation of diagonal sub-block b (executed by wave 0 only) --------------------------
struct FState {
    double* gptr;       // &A[row r, current column]
    double* cptr;       // &image[r]: where the owner half writes its element of the current column
    double* dummy;      // a word of this lane's own: where the OTHER half's lanes write instead (no branch in the step)
    const double* bufh; // &image[16 h]: base of this lane's broadcast reads
    int* flag;          // LDS counter: number of factor columns published so far (all stages)
    int flag_base;      // 32 b
    int64_t lda, colbase;  // colbase: global column index of the sub-block's first column
    double p_exc, ip_exc;  // replacement pivot of the exception rule (NaN = failure)
    int r, h, lane, mode, ncols_ok;  // ncols_ok: columns j < ncols_ok lie inside the matrix
    bool row_ok;
};

// The reciprocal pivot of step J + 1 is a chain of dependent f64 operations on the critical path of the whole
// factorisation (and every one of them queues behind the MFMAs of a co-resident GEMM wave), so it is kept minimal:
// v_rsq_f64 seed + two Newton steps give r = 1 / sqrt(d) to about an ulp, and that IS what the next column is
// multiplied by.  sqrt(d) itself (the stored diagonal) is derived off the chain.  The stages are interleaved with the
// rest of the step (pinned with sched_barrier).
struct PivotChain {
    double d, r, h, t;
};

template <bool M3, int K>
__device__ __forceinline__ void chain_stage(PivotChain& c)
{
    if constexpr (M3) {  // the block already holds the factor: 1 / d (reciprocal seed + two Newton steps)
        if constexpr (K == 0) c.r = __builtin_amdgcn_rcp(c.d);
        if constexpr (K == 1) c.t = __builtin_fma(-c.d, c.r, 1.0);
        if constexpr (K == 2) c.r = __builtin_fma(c.r, c.t, c.r);
        if constexpr (K == 3) c.t = __builtin_fma(-c.d, c.r, 1.0);
        if constexpr (K == 4) c.r = __builtin_fma(c.r, c.t, c.r);
        if constexpr (K == 0 || K == 2 || K == 4) pin(c.r);
        if constexpr (K == 1 || K == 3) pin(c.t);
    } else {
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
        if constexpr (K == 7) c.r = (c.d == 0.0) ? __builtin_inf() : c.r;  // plain-sqrt mode: the reference divides by 0
        if constexpr (K == 0 || K == 3 || K == 6 || K == 7) pin(c.r);
        if constexpr (K == 1 || K == 2 || K == 4 || K == 5) pin(c.t);
        if constexpr (K == 0) pin(c.h);
    }
}

template <bool M3>
struct ChainShape {
    static constexpr int NST = M3 ? 5 : 8;
};

// Pivot rule for a diagonal value that is 0, negative or NaN (any mode but plain-sqrt): the replacement pivot
// (sqrt(sub) and its reciprocal, or NaN = failure) is prepared once per launch and selected without a branch; the
// column is only noted in a bit mask.  The log in global memory is written after the unrolled steps: a memory access on
// a rare path inside them would make every step wait for the outstanding factor-column stores at the join.
__device__ __forceinline__ void pivot_select(const FState& st, double d, int j, double& ip, unsigned& excmask)
{
    const bool bad = !(st.mode == 2 || d > 0.0);
    ip = bad ? st.ip_exc : ip;
    excmask |= bad ? (1u << j) : 0u;
}

// pairs P .. P_END - 1 of step J: a(r, c) -= L(r, J) L(c, J) for the lane's slots.  Slots whose column is already final are dead
// registers in this routine: they are updated along with the others (no predicate).
template <int J, int P, int P_END>
__device__ __forceinline__ void f_pairs(double (&a)[HB], double l, const ColBcast& cb)
{
    if constexpr (P < P_END) {
        if constexpr (2 * P >= first_live_slot(J)) {
            a[2 * P] = __builtin_fma(-l, cb.v[P].x, a[2 * P]);
            pin(a[2 * P]);
        }
        if constexpr (2 * P + 1 >= first_live_slot(J)) {
            a[2 * P + 1] = __builtin_fma(-l, cb.v[P].y, a[2 * P + 1]);
            pin(a[2 * P + 1]);
        }
        f_pairs<J, P + 1, P_END>(a, l, cb);
    }
}

// one elimination step; J is a compile-time constant so that every register index and lane select is static
// (template recursion instead of `#pragma unroll`: the body is beyond clang's pragma-unroll budget).
// d = the diagonal value of column J, ip = its reciprocal pivot after the pivot rule.
//
// The step is bound by INSTRUCTION ISSUE (round 3: ~100 instructions per column at ~5.6 cycles each on a SIMD shared with an
// update wave -- not by the LDS round trip of the column broadcast, which was tested), so what does not have to happen per
// column does not: the factor's columns are no longer stored to global memory here (the stored diagonal sqrt(d) with its
// Heron correction, the pivot-exception selects and a predicated store were ~16 instructions of every step) -- the step only
// records d in lane J of a register pair and factor_subblock writes the whole 32 x 32 sub-block from its LDS image afterwards,
// every lane computing ITS row's diagonal entry once; the image write is branch-free (the half that does not own the column
// writes to a dummy word); FULL blocks (n = 128) drop the padding predicates.
template <bool M3, bool FULL, int J>
__device__ __forceinline__ void f_step(double (&a)[HB], FState& st, double d, double ip, unsigned& excmask, int& dlo, int& dhi)
{
    constexpr int hJ = J / HB, kJ = J % HB;
    constexpr int hN = (J + 1) / HB, kN = (J + 1) % HB;
    constexpr int NST = ChainShape<M3>::NST;
    constexpr bool next = J + 1 < SB;
    // ---- critical path first: column J scaled by the reciprocal pivot, the next diagonal value, the head of its chain.
    // The next diagonal value needs no broadcast: its lane multiplies by its own L(J + 1, J).
    const double v = a[kJ];
    const double q = M3 ? v : v * ip;
    PivotChain ch;
    if constexpr (next && hN == hJ) {
        double dn = a[kN];
        if constexpr (!M3) dn = __builtin_fma(-q, q, dn);
        ch.d = readlane_f64(dn, (J + 1) + SB * hN);  // next pivot candidate (uniform)
        chain_stage<M3, 0>(ch);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- L(r, J) in both halves.  Padding rows / columns (block smaller than 128) are forced to stay the identity:
    // 0 * inf = NaN would otherwise leak from an overflowing substituted factor into the log
    const bool keep = FULL ? (st.r > J) : (st.r > J && st.row_ok && J < st.ncols_ok);
    const double l_own = keep ? q : 0.0;
    const double l = bcast_half<hJ>(l_own);
    if constexpr (next && hN != hJ) {  // the pivot lane sits in the other half (J = 15): it needs the exchange first
        double dn = a[kN];
        if constexpr (!M3) dn = __builtin_fma(-l, l, dn);
        ch.d = readlane_f64(dn, (J + 1) + SB * hN);
        chain_stage<M3, 0>(ch);
    }
    if constexpr (next && 1 < NST) chain_stage<M3, 1>(ch);
    __builtin_amdgcn_sched_barrier(0);
    // ---- the LDS image of column J: 1 / pivot on the diagonal, L below, zeros above (the owner half into the image, the other
    // half into its dummy words: no branch); then the counter.  d_J goes into lane J of (dlo, dhi) for the store pass.
    {
        double* wp = (st.h == hJ) ? st.cptr + SB * J : st.dummy;
        *wp = (st.r == J) ? ip : l_own;
        *st.flag = st.flag_base + J + 1;  // after the column in this wave's LDS order: the solves of P1 may consume it
    }
    if constexpr (!M3) {
        // (v_writelane_b32: lane J of the pair takes the wave-uniform d; inline asm -- this hipcc has no builtin for it)
        asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(dlo) : "s"(__builtin_amdgcn_readfirstlane(__double2loint(d))), "n"(J));
        asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(dhi) : "s"(__builtin_amdgcn_readfirstlane(__double2hiint(d))), "n"(J));
    }
    if constexpr (next && 2 < NST) chain_stage<M3, 2>(ch);
    __builtin_amdgcn_sched_barrier(0);
    ColBcast cb;
    if constexpr (!M3) col_load<J, 0>(cb, st.bufh);
    if constexpr (next && 3 < NST) chain_stage<M3, 3>(ch);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!M3) {
        if constexpr (next && 4 < NST) chain_stage<M3, 4>(ch);
        if constexpr (next && 5 < NST) chain_stage<M3, 5>(ch);
        __builtin_amdgcn_sched_barrier(0);
        f_pairs<J, 0, HB / 4>(a, l, cb);
        if constexpr (next && 6 < NST) chain_stage<M3, 6>(ch);
        __builtin_amdgcn_sched_barrier(0);
        f_pairs<J, HB / 4, HB / 2>(a, l, cb);
        if constexpr (next && 7 < NST) chain_stage<M3, 7>(ch);
    } else {
        if constexpr (next && 4 < NST) chain_stage<M3, 4>(ch);
    }
    if constexpr (next) {
        double ipn = ch.r;
        if constexpr (!M3) pivot_select(st, ch.d, J + 1, ipn, excmask);
        f_step<M3, FULL, J + 1>(a, st, ch.d, ipn, excmask, dlo, dhi);
    }
}

// ---- the same elimination step, software-pipelined (round 5) ------------------------------------------------------
// Measured (scripts/mk_potf2_phases.py, FR_K4_EXP = 1): with every follower wave switched off F_b still takes ~480 cycles per
// column -- the step is bound by the pivot wave's OWN dependencies, in order: reciprocal-pivot chain (13 dependent f64
// operations), then the LDS round trip of the column (write -> eight broadcast reads -> s_waitcnt), then the sixteen rank-1
// FMAs; nothing overlaps because the wave issues in order and the FMAs wait for the round trip.  Here the rank-1 update of
// column J is applied one step LATE (step J + 1, from broadcasts requested in step J: they have landed, no wait), which leaves
// two things the next pivot needs in time:
//   * the next diagonal value: every lane keeps  dd = a(r, r) - sum_k L(r, k)^2  of ITS row, updated with its own L(r, J) --
//     no broadcast, no LDS;
//   * column J + 1 complete before step J + 1 scales it: its update from column J is one FMA with L(J + 1, J) read from the
//     pivot lane (v_readlane), issued in the shadow of the reciprocal-pivot chain.
// The dependent chain of a step is then  q = a ip -> dd -= q^2 -> readlane -> rsq + Newton -> select; everything else fills
// its issue gaps.
template <bool FULL, int J>
__device__ __forceinline__ void f_step_lazy(double (&a)[HB], FState& st, double& dd, double d, double ip, unsigned& excmask, int& dlo,
                                            int& dhi, double lprev, const ColBcast& cbp)
{
    constexpr int hJ = J / HB, kJ = J % HB;
    constexpr int hN = (J + 1) / HB, kN = (J + 1) % HB;
    constexpr int NST = ChainShape<false>::NST;
    constexpr bool next = J + 1 < SB;
    // ---- critical path: column J scaled by the reciprocal pivot, the row's diagonal accumulator, the next pivot candidate
    const double v = a[kJ];
    const double q = v * ip;
    const bool keep = FULL ? (st.r > J) : (st.r > J && st.row_ok && J < st.ncols_ok);
    const double l_own = keep ? q : 0.0;  // (padding rows / columns stay the identity: 0 * inf must not leak)
    PivotChain ch;
    if constexpr (next) {
        if constexpr (FULL) dd = __builtin_fma(-q, q, dd);
        else dd = __builtin_fma(-l_own, l_own, dd);
        pin(dd);
        ch.d = readlane_f64(dd, (J + 1) + SB * hJ);  // (the accumulators live in the half that owns column J)
        chain_stage<false, 0>(ch);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- L(r, J) in both halves; L(J + 1, J) for the early update of column J + 1
    const double l = bcast_half<hJ>(l_own);
    double sl = 0.0;
    if constexpr (next) sl = readlane_f64(q, (J + 1) + SB * hJ);
    if constexpr (next) chain_stage<false, 1>(ch);
    __builtin_amdgcn_sched_barrier(0);
    // ---- the LDS image of column J and the counter (as f_step); its broadcasts for the NEXT step's rank-1 update are requested
    // at once (a second set of 32 registers: nothing in this step waits for them)
    {
        double* wp = (st.h == hJ) ? st.cptr + SB * J : st.dummy;
        *wp = (st.r == J) ? ip : l_own;
        *st.flag = st.flag_base + J + 1;
    }
    ColBcast cb;
    if constexpr (J + 2 < SB) col_load<J + 1, 0>(cb, st.bufh + SB * J - SB * (J + 1));  // (column J, liveness of columns > J + 1)
    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(dlo) : "s"(__builtin_amdgcn_readfirstlane(__double2loint(d))), "n"(J));
    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(dhi) : "s"(__builtin_amdgcn_readfirstlane(__double2hiint(d))), "n"(J));
    if constexpr (next) chain_stage<false, 2>(ch);
    __builtin_amdgcn_sched_barrier(0);
    // ---- the rank-1 update of column J - 1, one step late (columns > J; its broadcasts were requested a step ago), in four
    // groups between the stages of the reciprocal-pivot chain
    constexpr bool lazy = (J >= 1 && J + 1 < SB);
    if constexpr (lazy) f_pairs<J, 0, HB / 8>(a, lprev, cbp);
    if constexpr (next) chain_stage<false, 3>(ch);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (lazy) f_pairs<J, HB / 8, HB / 4>(a, lprev, cbp);
    if constexpr (next) chain_stage<false, 4>(ch);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (lazy) f_pairs<J, HB / 4, 3 * HB / 8>(a, lprev, cbp);
    if constexpr (next) chain_stage<false, 5>(ch);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (lazy) f_pairs<J, 3 * HB / 8, HB / 2>(a, lprev, cbp);
    // ---- column J + 1 takes column J's update now (its slot belongs to half hN)
    if constexpr (next) {
        const double lf = (st.h == hN) ? l : 0.0;
        a[kN] = __builtin_fma(-lf, sl, a[kN]);
        pin(a[kN]);
        if constexpr (J == HB - 1) {  // the accumulators move to the half that owns the columns from here on
            dd = bcast_half<0>(dd);
            pin(dd);
        }
    }
    if constexpr (next) chain_stage<false, 6>(ch);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (next) chain_stage<false, 7>(ch);
    if constexpr (next) {
        double ipn = ch.r;
        pivot_select(st, ch.d, J + 1, ipn, excmask);
        f_step_lazy<FULL, J + 1>(a, st, dd, ch.d, ipn, excmask, dlo, dhi, l, cb);
    }
}

template <bool M3, bool FULL, bool LAZY = false>
__device__ __forceinline__ void factor_subblock(double* lds, int b, int lane, double* __restrict__ A, int64_t lda, int n,
                                                int64_t col0, int mode, double sub, int64_t* __restrict__ info)
{
    FState st;
    st.flag = reinterpret_cast<int*>(lds + NSLOT * SBE);
    st.flag_base = SB * b;
    st.lane = lane;
    st.r = lane & (SB - 1);
    st.h = lane >> 5;
    double* image = lds + slot_of(b, b);
    st.cptr = image + st.r;
    st.dummy = lds + NSLOT * SBE + 2 + st.r;  // behind the counter's 16 bytes
    st.bufh = image + HB * st.h;
    st.gptr = A + (SB * b + st.r) + (int64_t)(SB * b) * lda;
    st.row_ok = SB * b + st.r < n;
    st.ncols_ok = n - SB * b;
    st.lda = lda;
    st.colbase = col0 + SB * b;
    st.mode = mode;
    st.p_exc = __builtin_nan("");
    st.ip_exc = st.p_exc;
    const bool substitute = (mode == 1 && sub > 0.0);
    if (substitute) sqrt_rsqrt(sub, st.p_exc, st.ip_exc);
    double a[HB];
#pragma unroll
    for (int k = 0; k < HB; ++k) a[k] = image[st.r + SB * (HB * st.h + k)];
    // first pivot
    unsigned excmask = 0;
    const double d0 = readlane_f64(a[0], 0);
    double ip;
    if constexpr (M3) {
        ip = 1.0 / d0;
    } else {
        double p;
        sqrt_rsqrt(d0, p, ip);
        pivot_select(st, d0, 0, ip, excmask);
    }
    int dlo = 0, dhi = 0;  // lane J: the diagonal value d_J the pivot of column J was taken from
    if constexpr (LAZY && !M3) {
        double dd = image[st.r + SB * st.r];  // a(r, r): the row's diagonal accumulator (both halves)
        ColBcast none;
        f_step_lazy<FULL, 0>(a, st, dd, d0, ip, excmask, dlo, dhi, 0.0, none);
    } else {
        f_step<M3, FULL, 0>(a, st, d0, ip, excmask, dlo, dhi);
    }
    if constexpr (!M3) {
        // ---- the factored sub-block to global memory, from its image (L below the diagonal, 1 / pivot on it): every lane
        // forms the diagonal entry of ITS row, sqrt(d) = d ip with one Heron correction (the pivot rule's replacement where
        // it fired), and stores its 16 slots of the lower triangle.  (The image is overwritten in P2: this runs in front of
        // the stage's first barrier; the update waves are still a column behind.)
        const double dr = bcast_half<0>(__hiloint2double(dhi, dlo));  // d of row r (recorded in lane r of the lower half)
        const double ipr = image[st.r + SB * st.r];
        double pq = dr * ipr;
        pq = __builtin_fma(0.5 * ipr, __builtin_fma(-pq, pq, dr), pq);
        pq = (dr == 0.0) ? 0.0 : pq;                        // plain-sqrt mode, d == 0
        pq = ((excmask >> st.r) & 1u) ? st.p_exc : pq;      // substituted / failed pivot
        double vals[HB];
#pragma unroll
        for (int k = 0; k < HB; ++k) vals[k] = image[st.r + SB * (HB * st.h + k)];
#pragma unroll
        for (int k = 0; k < HB; ++k) {
            const int c = HB * st.h + k;
            if (st.r >= c && st.row_ok && c < st.ncols_ok) st.gptr[(int64_t)c * lda] = (st.r == c) ? pq : vals[k];
        }
    }
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
}

// ---- triangular solve against the LDS image of L_bb (one wave, 32 vectors of 32, split-row layout) -------------------
// x <- solution of  L y = x  (forward substitution, right-looking):  y[J] = x[J] / L(J, J);  x[c] -= y[J] L(c, J), c > J.
// Used for T (x = a row of A_ib: the row of L_ib is y) and for X_bb (x = e_c: y is column c of L_bb^-1).  Here the slots
// of finished columns hold results, so a slot is updated only where 16 h + k > J (two multipliers per step).  The
// broadcasts of column J + 1 are issued while column J is applied (the image is read-only in this phase).
// The solves run CONCURRENTLY with F_b on other waves: they consume column J of the image as soon as the panel wave has
// published it (LDS counter, polled), one step behind the pivot chain, and finish a step after it.
// `avail` caches the last value seen: a solve that has fallen behind the pivot chain does not touch the counter again
// until it has caught up (next to a GEMM workgroup an LDS round trip costs ~900 cycles).
__device__ __forceinline__ void wait_columns(const int* flag, int target, int& avail)
{
    while (avail < target) {
        avail = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const volatile int*>(flag));
        if (avail < target) __builtin_amdgcn_s_sleep(1);
    }
}

template <int J>
__device__ __forceinline__ void trsm_step(double (&x)[HB], const double* bufh, int h, const ColBcast& cb, double ipj,
                                          const int* flag, int flag_base, int& avail)
{
    constexpr int hJ = J / HB, kJ = J % HB;
    ColBcast nx;
    double ipn = 0.0;
    if constexpr (J + 1 < SB) {
        wait_columns(flag, flag_base + J + 2, avail);
        col_load<J + 1, 0>(nx, bufh);
        ipn = bufh[(J + 1) + SB * (J + 1) - HB * h];  // image[(J+1) + 32 (J+1)], uniform
    }
    const double y_own = x[kJ] * ipj;
    x[kJ] = (h == hJ) ? y_own : x[kJ];
    pin(x[kJ]);
    const double y = bcast_half<hJ>(y_own);
    const double m_ge = (hJ == 0) ? y : ((h == 1) ? y : 0.0);  // halves h >= hJ
    const double m_gt = (h == 1) ? y : 0.0;                    // halves h >  hJ (only when hJ == 0)
#pragma unroll
    for (int k = 0; k < HB; ++k) {
        const double lv = (k & 1) ? cb.v[k >> 1].y : cb.v[k >> 1].x;
        if (k > kJ) {
            x[k] = __builtin_fma(-m_ge, lv, x[k]);
            pin(x[k]);
        } else if (hJ == 0) {
            x[k] = __builtin_fma(-m_gt, lv, x[k]);
            pin(x[k]);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (J + 1 < SB) trsm_step<J + 1>(x, bufh, h, nx, ipn, flag, flag_base, avail);
}

__device__ __forceinline__ void trsm_fwd(double (&x)[HB], const double* image, int h, const int* flag, int flag_base)
{
    const double* bufh = image + HB * h;
    ColBcast cb;
    int avail = 0;
    wait_columns(flag, flag_base + 1, avail);
    col_load<0, 0>(cb, bufh);
    const double ip0 = image[0];
    trsm_step<0>(x, bufh, h, cb, ip0, flag, flag_base, avail);
}

template <bool LAZY>
__device__ __forceinline__ void potf2_block(double* lds, double* __restrict__ A, int64_t lda, int n, int64_t col0, int mode,
                                            double sub, double* __restrict__ inv, int64_t ldinv,
                                            int64_t* __restrict__ info, double* __restrict__ cest = nullptr)
{
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int w = t >> 6;  // wave, uniform
    const int r = lane & (SB - 1);
    const int h = lane >> 5;
    const int nblk = (n + SB - 1) / SB;
    const bool want_inv = inv != nullptr;
    const bool m3 = (mode == 3);

    // ---- load the lower triangle into the LDS slots; rows / columns >= n are padded with the identity.  All 20 loads
    //      of a thread are issued before the first LDS write (one memory round trip, not ten)
    {
        const int cg = t >> 5;  // 16 column groups, 2 columns each per slot
        double v[2 * NSLOT];
#pragma unroll
        for (int sl = 0; sl < NSLOT; ++sl) {
            const int i = (sl >= 6) ? 3 : ((sl >= 3) ? 2 : ((sl >= 1) ? 1 : 0));
            const int k = sl - i * (i + 1) / 2;
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                const int gr = SB * i + r, gc = SB * k + cg * 2 + cc;
                double x = (gr == gc) ? 1.0 : 0.0;
                if (gr < n && gc < n && gr >= gc) x = A[gr + (int64_t)gc * lda];
                if (gr < n && gc < n && gr < gc) x = 0.0;
                v[2 * sl + cc] = x;
            }
        }
#pragma unroll
        for (int sl = 0; sl < NSLOT; ++sl) {
            const int i = (sl >= 6) ? 3 : ((sl >= 3) ? 2 : ((sl >= 1) ? 1 : 0));
            if (i < nblk) {
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) lds[sl * SBE + r + SB * (cg * 2 + cc)] = v[2 * sl + cc];
            }
        }
    }
    int* colflag = reinterpret_cast<int*>(lds + NSLOT * SBE);
    if (t == 0) *colflag = 0;
    lds_barrier();

    // Per stage b (barriers between the phases):
    //   P1  wave 0: F_b, the serial pivot chain; concurrently, one step behind it (column counter in LDS):
    //       T: L_ib = A_ib L_bb^-T (one wave per sub-block below)  |  X_bb (one wave)  |  (1): W_bc = L_bb^-1 W_bc, c < b
    //       -- all three are the same triangular solve against the image of L_bb, on rows of A_ib, on e_c, on columns of W_bc
    //   P2  X_bb replaces the image  |  U: A_ik -= L_ib L_kb^T  |  (2): W_ic -= L_ib W_bc      (MFMA 16x16 tiles, all waves)
    //   P3  (3): W_ib = -L_ib X_bb, in place over the now dead L_ib (one wave per sub-block)
    // The inverse sub-blocks are kept TRANSPOSED in LDS (WT_ic[y + 32 x] = W_ic(x, y)): it makes every product an
    // "N T" contraction whose two operand fragments and whose result tile are contiguous along the 16 lanes of an MFMA
    // register, and it is the layout in which a lane pair of the solve naturally stores its column of X_bb.
    // Wave 0 runs only F_b and the barriers (its own branch keeps the unrolled pivot chain free of the update phases'
    // register pressure); waves 1..7 are the update waves.
    if (w == 0) {
        // the pivot chain is the critical path of the whole factorisation and shares its SIMD with a GEMM wave of the
        // co-resident trailing update: take the issue slot whenever both are ready
        __builtin_amdgcn_s_setprio(3);
        for (int b = 0; b < nblk; ++b) {
            if (m3)
                factor_subblock<true, false>(lds, b, lane, A, lda, n, col0, mode, sub, info);
            else if (n == PB)
                factor_subblock<false, true, LAZY>(lds, b, lane, A, lda, n, col0, mode, sub, info);
            else
                factor_subblock<false, false, LAZY>(lds, b, lane, A, lda, n, col0, mode, sub, info);
            lds_barrier();
            lds_barrier();
            lds_barrier();
        }
    } else {
    const int u = w - 1;
    for (int b = 0; b < nblk; ++b) {
        double* Lbb = lds + slot_of(b, b);  // image of L_bb (1 / pivot on the diagonal), later XT_bb

        // ---- P1
        double xs[HB];
#if FR_K4_EXP == 2 || FR_K4_EXP == 4
        // experiment: the four followers of a stage on waves 1, 2, 3, 5 -- wave 4 (the pivot wave's SIMD partner) idles in P1
        const int q4 = (u < 3) ? u : ((u == 4) ? 3 : -1);
        const int nt4 = m3 ? 0 : (nblk - 1 - b);
        const bool do_x = want_inv && q4 == nt4;
        {
            const int it = b + 1 + q4;
            const int c1 = q4 - nt4 - 1;
            const bool do_t = q4 >= 0 && q4 < nt4;
            const bool do_1 = want_inv && q4 > nt4 && c1 < b;
#else
        const bool do_x = (u == 3) && want_inv;
        {
            const int it = b + 1 + u;            // update waves 0..2: T on sub-block it
            const int c1 = u - 4;                // update waves 4..6: (1) on W_b,c1
            const bool do_t = (u < 3) && !m3 && (it < nblk);
            const bool do_1 = (u >= 4) && want_inv && (c1 < b);
#endif
            double* S = lds + (do_t ? slot_of(it, b) : (do_1 ? slot_of(b, c1) : 0));
            if (do_t || do_x || do_1) {
                double one = 1.0;
                pin(one);  // not loop-invariant for the compiler: it would hoist e_c out of the stage loop and spill it
#pragma unroll
                for (int k = 0; k < HB; ++k) {
                    // T: row r of A_ib (normal layout);  (1): column r of W_bc (transposed layout: same addresses);  X: e_r
                    const double v = S[r + SB * (HB * h + k)];
                    xs[k] = do_x ? ((HB * h + k == r) ? one : 0.0) : v;
                }
                if (K4X != 1 && !((K4X == 3 || K4X == 4) && !do_x)) trsm_fwd(xs, Lbb, h, colflag, SB * b);  // (experiments 1 / 3: no followers / only X)
            }
            if (do_t || do_1) {
                const int gr = SB * it + r;
#pragma unroll
                for (int k = 0; k < HB; ++k) {
                    S[r + SB * (HB * h + k)] = xs[k];
                    const int gc = SB * b + HB * h + k;
                    if (do_t && gr < n && gc < n) A[gr + (int64_t)gc * lda] = xs[k];
                }
            }
        }
        lds_barrier();

        // ---- P2
        if (do_x) {
            // lane pair (c, h) holds X(16 h + k, c): stored transposed, conflict-free
#pragma unroll
            for (int k = 0; k < HB; ++k) Lbb[r + SB * (HB * h + k)] = xs[k];
        }
        {
            const int rem = nblk - b - 1;
            const int nU = m3 ? 0 : rem * (rem + 1) / 2;
            const int n2 = want_inv ? rem * b : 0;
            const int l15 = lane & 15, lq = lane >> 4;
            // Update wave 3 joins after its store: the list is dealt to the other waves first.
            // One 32 x 32 product per wave and turn: its four 16 x 16 accumulators share the operand fragments (two reads per
            // k-step and operand instead of four -- the fragment reads are 4-way bank-conflicted at this sub-block stride and
            // seven waves share one LDS, so this phase is bound by LDS traffic, not by the matrix core).
            const int npair = nU + n2;
            for (int pidx = (u + 3) % 7; pidx < npair; pidx += 7) {
                const double* Pp;
                const double* Qp;
                double* Cp;
                bool tr;
                if (pidx < nU) {
                    // pairs in the order (1,1) (2,1) (2,2) (3,1) (3,2) (3,3), relative to b:  C[x + 32 y] (x along the lanes)
                    const int ri = (pidx >= 3) ? 3 : ((pidx >= 1) ? 2 : 1);
                    const int rk = pidx - ri * (ri - 1) / 2 + 1;
                    tr = false;
                    Pp = lds + slot_of(b + ri, b) + l15 + SB * lq;
                    Qp = lds + slot_of(b + rk, b) + l15 + SB * lq;
                    Cp = lds + slot_of(b + ri, b + rk);
                } else {
                    // (2): WT_i,cb[y + 32 x] (y along the lanes) -= L_ib WT_b,cb
                    const int q = pidx - nU;
                    const int bb = b > 0 ? b : 1;
                    const int i = b + 1 + q / bb, cb = q % bb;
                    tr = true;
                    Pp = lds + slot_of(i, b) + l15 + SB * lq;
                    Qp = lds + slot_of(b, cb) + l15 + SB * lq;
                    Cp = lds + slot_of(i, cb);
                }
                d4_t acc[2][2];  // [tx][ty]
#pragma unroll
                for (int tx = 0; tx < 2; ++tx)
#pragma unroll
                    for (int ty = 0; ty < 2; ++ty) acc[tx][ty] = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < SB / 4; ++kk) {
                    double pf[2], qf[2];
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        pf[tt] = Pp[HB * tt + SB * 4 * kk];
                        qf[tt] = Qp[HB * tt + SB * 4 * kk];
                    }
#pragma unroll
                    for (int tx = 0; tx < 2; ++tx)
#pragma unroll
                        for (int ty = 0; ty < 2; ++ty)
                            // result index of the FIRST operand runs over lq + 4 reg, of the SECOND over the lanes
                            acc[tx][ty] = tr ? __builtin_amdgcn_mfma_f64_16x16x4f64(pf[tx], qf[ty], acc[tx][ty], 0, 0, 0)
                                             : __builtin_amdgcn_mfma_f64_16x16x4f64(qf[ty], pf[tx], acc[tx][ty], 0, 0, 0);
                }
#pragma unroll
                for (int tx = 0; tx < 2; ++tx)
#pragma unroll
                    for (int ty = 0; ty < 2; ++ty) {
                        double* C = tr ? Cp + (HB * ty + l15) + SB * (HB * tx + lq) : Cp + (HB * tx + l15) + SB * (HB * ty + lq);
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) C[SB * 4 * reg] -= acc[tx][ty][reg];
                    }
            }
        }
        lds_barrier();

        // ---- P3
        if (want_inv && u < 3 && b + 1 + u < nblk) {
            const int l15 = lane & 15, lq = lane >> 4;
            double* Wib = lds + slot_of(b + 1 + u, b);  // L_ib (normal) in, WT_ib out
            d4_t acc[2][2];
#pragma unroll
            for (int tx = 0; tx < 2; ++tx)
#pragma unroll
                for (int ty = 0; ty < 2; ++ty) acc[tx][ty] = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < SB / 4; ++kk) {
                double pf[2], qf[2];
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    pf[tt] = Wib[(HB * tt + l15) + SB * (4 * kk + lq)];
                    qf[tt] = Lbb[(HB * tt + l15) + SB * (4 * kk + lq)];  // XT_bb
                }
#pragma unroll
                for (int tx = 0; tx < 2; ++tx)
#pragma unroll
                    for (int ty = 0; ty < 2; ++ty)
                        acc[tx][ty] = __builtin_amdgcn_mfma_f64_16x16x4f64(pf[tx], qf[ty], acc[tx][ty], 0, 0, 0);
            }
            // every read of L_ib is behind us (one wave, program order): overwrite it with -(...) transposed
#pragma unroll
            for (int tx = 0; tx < 2; ++tx)
#pragma unroll
                for (int ty = 0; ty < 2; ++ty)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg)
                        Wib[(HB * ty + l15) + SB * (HB * tx + lq + 4 * reg)] = -acc[tx][ty][reg];
        }
        lds_barrier();
    }
    }

    // ---- store the inverse: the LDS sub-blocks are transposed (zeros above the diagonal).  Along the way: a conditioning
    // estimate of the block, max |W_ij| * max L_jj (W = L^-1 has 1 / L_jj on its diagonal), for the host's decision whether
    // the products with W need a step of iterative refinement (chol.hip); and zeros above the diagonal of the factored
    // block in A, so that refinement can use L_bb as a plain 128 x 128 operand.
    if (want_inv) {
        // The read is a transposition (LDS holds W^T): with the 32 rows of a sub-block along the lanes every lane of a read hits
        // the same bank (32-way conflict, 12.7 k of the kernel's 131 k cycles).  8 rows x 8 columns per wave-instruction instead:
        // 8-way conflicts in LDS, 64-byte segments in memory.
        const int ra = t & 7, cl = (t >> 3) & 7;        // row / column inside the wave's 8 x 8 patch
        const int rb = (t >> 6) & 3, cq = t >> 8;        // wave: rows 8 rb .., columns 16 cq + 8 cc ..
        const int rr = 8 * rb + ra;
        double wmax = 0.0, dmin = __builtin_inf();
        for (int i = 0; i < nblk; ++i)
            for (int k = 0; k < nblk; ++k) {
                const double* s = lds + slot_of(i, k <= i ? k : 0);
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    const int c = 16 * cq + 8 * cc + cl;
                    const int gr = SB * i + rr, gc = SB * k + c;
                    if (gr < n && gc < n) {
                        const double v = (k <= i) ? s[c + SB * rr] : 0.0;
                        inv[gr + (int64_t)gc * ldinv] = v;
                        const double av = __builtin_fabs(v);
                        wmax = (av > wmax || av != av) ? av : wmax;  // NaN sticks
                        if (gr == gc) dmin = (av < dmin || av != av) ? av : dmin;
                        if (gr < gc) A[gr + (int64_t)gc * lda] = 0.0;
                    }
                }
            }
        if (cest) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double ow = __shfl_xor(wmax, off, 64), od = __shfl_xor(dmin, off, 64);
                wmax = (ow > wmax || ow != ow) ? ow : wmax;
                dmin = (od < dmin || od != od) ? od : dmin;
            }
            lds_barrier();  // every read of the inverse slots is done: their memory takes the per-wave partials
            if (lane == 0) {
                lds[2 * w] = wmax;
                lds[2 * w + 1] = dmin;
            }
            lds_barrier();
            if (t == 0) {
                for (int q = 1; q < PT / 64; ++q) {
                    const double ow = lds[2 * q], od = lds[2 * q + 1];
                    wmax = (ow > wmax || ow != ow) ? ow : wmax;
                    dmin = (od < dmin || od != od) ? od : dmin;
                }
                *cest = wmax / dmin;
            }
        }
    }
}

__global__ __launch_bounds__(PT, 4) void potf2_kernel(double* __restrict__ A, int64_t lda, int n, int64_t col0, int mode,
                                                   double sub, double* __restrict__ inv, int64_t ldinv,
                                                   int64_t* __restrict__ info, double* __restrict__ cest,
                                                   unsigned* __restrict__ xcc_word)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    if (xcc_word && threadIdx.x == 0) {
        // tell the trailing update which XCD to leave alone (gemm_f64.hip, option xcd_reserve)
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        __hip_atomic_store(xcc_word, (xcc & 7u) + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    potf2_block<false>(lds, A, lda, n, col0, mode, sub, inv, ldinv, info, cest);
}

// The same body without the register cap (157 VGPRs, nothing in scratch; capped at 128 it spills 76 B per lane, 41 scratch
// instructions spread over its phases).  The cap exists so that the kernel fits BESIDE a resident GEMM workgroup; while XCDs
// are set aside for the panel chain, and on the chain stream of a sharded factorisation, the diagonal-block kernel has its CU
// to itself and the cap only costs.  FRIEDRICH_AMD_K4_UNCAPPED = 0 / 1 forces a variant (A/B runs).
__global__ __launch_bounds__(PT, 2) void potf2_uncapped_kernel(double* __restrict__ A, int64_t lda, int n, int64_t col0, int mode,
                                                            double sub, double* __restrict__ inv, int64_t ldinv,
                                                            int64_t* __restrict__ info, double* __restrict__ cest,
                                                            unsigned* __restrict__ xcc_word)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    if (xcc_word && threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        __hip_atomic_store(xcc_word, (xcc & 7u) + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    potf2_block<K4X != 5>(lds, A, lda, n, col0, mode, sub, inv, ldinv, info, cest);
}

// =====================================================================================================================
// Flat variant of the diagonal-block kernel (round 5): full 128 x 128 blocks, factor + inverse, the CU to itself.
//
// What the phase stamps and scripts/lat_probe.hip said about the staged kernel above: a single wave issues one instruction
// per ~5.2 cycles, so a pivot step costs (instructions of the pivot wave) x 5.2 whatever the other seven waves do -- 75
// instructions, ~460 cycles per column, 4 x 14.6 k cycles -- and the 32^3 products between the stages (25 k cycles) wait
// behind barriers.  Here the pivot wave carries almost nothing but the chain, and there are no stages:
//
//   * one ROW per lane.  Wave PA holds rows 0..63 of the block, PB rows 64..127; XA / XB hold rows 0..63 / 64..127 of
//     X^T = I L^-T (the inverse as "one more block of rows below the factor": row c of X^T is column c of L^-1).
//   * columns in MICRO-PANELS of four.  A lane keeps only the four values of its row in the current micro-panel.  The pivot
//     wave (PA for columns 0..63, then PB) runs the chain  q = a ip -> dd -= q^2 -> v_readlane -> rsq + Newton  of the lazy
//     step above and the three in-panel updates (multipliers read from the pivot lanes with v_readlane); the other row
//     waves FOLLOW a whole micro-panel at a time from the 4 x 4 triangle the pivot wave published (ten f64 operations).
//   * the rank-4 update of everything to the right of a micro-panel is v_mfma_f64_16x16x4 -- K = 4 IS the instruction's
//     contraction -- on four UPDATE waves: the 36 lower tiles of the block and the 36 upper tiles of X^T stay in their
//     accumulator registers (18 tiles per wave, NEGATED so that the products add) from the first load to the end; finished
//     micro-panels travel through a ring in LDS as operands; after micro-panel p the update waves hand the columns of
//     micro-panel p + 2 (four accumulator registers' worth per tile) back through a staging buffer, and the row waves
//     apply micro-panel p + 1 to them themselves (sixteen FMAs with broadcast multipliers).
//   * no barrier inside: eight progress counters in LDS (one per wave), polled.
// The dependency chain is then 128 pivot steps of ~45 instructions; the matrix-core work (~960 MFMAs) and the followers
// run beside it.
namespace flat {

constexpr int R = 8;          // ring depth (micro-panels)
constexpr int LAG = 3;        // the update waves hand micro-panel q's columns back with every update of micro-panels <= q - LAG; the row
                              // waves apply the LAG - 1 micro-panels in between themselves.  (LAG = 2: the loop pivot wave -> followers ->
                              // update waves -> pivot wave has to close within ONE micro-panel of the pivot wave, and does not: measured
                              // 1.2 k of 3.0 k cycles per micro-panel blocked)
constexpr int NSTG = 4;       // staging slots
constexpr int PS = 144;       // column stride of a micro-panel image in LDS, in doubles: 128 rows + 16 (operand reads of the
                              // MFMA layout -- 16 rows x 4 columns per wave-instruction -- then hit 64 distinct banks)
constexpr int PSZ = 4 * PS;   // doubles per micro-panel image
constexpr int OFF_L = 0;                  // L ring: finished micro-panels of the factor (rows 0..127)
constexpr int OFF_Y = R * PSZ;            // Y ring: finished micro-panels of X^T
constexpr int OFF_FS = 2 * R * PSZ;          // staging, factor rows: columns of micro-panel q (slot q % NSTG) with every update of micro-panels <= q - LAG
constexpr int OFF_XS = OFF_FS + NSTG * PSZ;  // staging, X^T rows
constexpr int OFF_END = OFF_XS + NSTG * PSZ;
constexpr int OFF_IP = OFF_END + 16;      // reciprocal pivots of the micro-panels in the ring (R x 4 x 64 copies)
#ifdef FR_K4_TS
constexpr size_t LDS_BYTES = (size_t)(OFF_IP + 256 * R) * sizeof(double) + 512;
#else
constexpr size_t LDS_BYTES = (size_t)(OFF_IP + 256 * R) * sizeof(double);  // + progress counters, exception masks, reduction scratch, reciprocal pivots
#endif

struct Tile {
    int kind, rt, ct;  // kind 0: tile (rt, ct), rt >= ct, of the block being factored; 1: tile (rt, ct), rt <= ct, of X^T
};
// 18 tiles per update wave, dealt so that the number of live tiles per micro-panel is level across the waves (<= 11).  A tile
// of the block is live while its columns lie beyond the micro-panel (p <= 4 ct + 2); a tile of X^T also only once its rows have
// begun (p >= 4 rt).  The panel loop of an update wave is FULLY specialised (template parameters p4, pp): which tiles a
// micro-panel touches is known at compile time, the wave runs straight-line code -- reads, products, hand-back -- with no
// per-tile test (a run-time test per tile costs more than the product: the structurizer turns eighteen uniform branches, or a
// switch with fall-through, into chains of flag registers and serialises the LDS reads behind them; measured 2.1-2.7 k cycles per
// micro-panel against ~0.7 k of matrix-pipe time).
constexpr Tile TILES[4][18] = {
    {{0, 3, 3}, {0, 4, 1}, {0, 5, 2}, {0, 5, 5}, {0, 6, 0}, {0, 6, 2}, {0, 7, 3}, {0, 7, 4}, {0, 7, 7}, {1, 0, 1}, {1, 0, 5}, {1, 1, 1}, {1, 2, 4}, {1, 3, 6}, {1, 3, 7}, {1, 5, 5}, {1, 6, 6}, {1, 6, 7}},
    {{0, 1, 0}, {0, 2, 0}, {0, 2, 2}, {0, 3, 1}, {0, 5, 4}, {0, 6, 3}, {0, 6, 5}, {0, 7, 1}, {0, 7, 2}, {1, 0, 0}, {1, 0, 7}, {1, 1, 5}, {1, 1, 6}, {1, 2, 5}, {1, 3, 4}, {1, 4, 6}, {1, 5, 6}, {1, 5, 7}},
    {{0, 0, 0}, {0, 2, 1}, {0, 3, 2}, {0, 4, 4}, {0, 5, 3}, {0, 6, 1}, {0, 6, 6}, {0, 7, 0}, {1, 0, 2}, {1, 0, 4}, {1, 0, 6}, {1, 1, 4}, {1, 2, 3}, {1, 2, 7}, {1, 3, 3}, {1, 3, 5}, {1, 4, 5}, {1, 7, 7}},
    {{0, 1, 1}, {0, 3, 0}, {0, 4, 0}, {0, 4, 2}, {0, 4, 3}, {0, 5, 0}, {0, 5, 1}, {0, 6, 4}, {0, 7, 5}, {0, 7, 6}, {1, 0, 3}, {1, 1, 2}, {1, 1, 3}, {1, 1, 7}, {1, 2, 2}, {1, 2, 6}, {1, 4, 4}, {1, 4, 7}},
};
constexpr bool tile_live(Tile t, int p)
{
    return p <= 4 * t.ct + 2 && (t.kind == 0 || p >= 4 * t.rt);
}

#ifdef FR_K4_TS
__device__ long long k4ts[128];  // developer stamps (scripts/mk_potf2_phases.py): per wave [8 w + 0] role cycles, [1] cycles spent waiting, [2] waits that blocked
#define FR_K4_WAIT_BEGIN const long long tw0 = __builtin_amdgcn_s_memtime(); bool blocked = false;
#define FR_K4_WAIT_BLOCKED blocked = true;
// (accumulated in LDS with no-return atomics: a read-modify-write of global memory here would wait for every outstanding store)
extern __shared__ __attribute__((aligned(16))) double k4lds[];
#define FR_K4_TSLOT(i) (reinterpret_cast<unsigned long long*>(k4lds + flat::OFF_IP + 256 * flat::R) + 8 * (threadIdx.x >> 6) + (i))
#define FR_K4_TADD(i, v) if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(FR_K4_TSLOT(i), (unsigned long long)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#define FR_K4_WAIT_END if (blocked) { FR_K4_TADD(1, __builtin_amdgcn_s_memtime() - tw0) FR_K4_TADD(2, 1) }
#else
#define FR_K4_WAIT_BEGIN
#define FR_K4_WAIT_BLOCKED
#define FR_K4_WAIT_END
#endif
// The counters are read and written with explicit DS instructions: a `volatile` access through a generic pointer is NOT
// rewritten to the LDS address space by the compiler -- it becomes flat_load / flat_store with sc0 sc1 and an s_waitcnt vmcnt(0),
// i.e. every poll and every publish would also wait for the wave's outstanding global stores of factor columns (measured: the
// first version of this kernel spent ~25 k of a pivot wave's 60 k cycles there).
__device__ __forceinline__ unsigned lds_off(const void* p)
{
    return (unsigned)(uintptr_t)p;  // (low half of a generic pointer into LDS = the LDS byte offset)
}
// four counters p[0..3] all >= target (skip0: p[0] does not count): ONE 16-byte LDS read per poll
__device__ __forceinline__ void wait4(const int* p, int target, bool skip0 = false)
{
    FR_K4_WAIT_BEGIN
    typedef int i4_t __attribute__((ext_vector_type(4)));
    const unsigned addr = lds_off(p);
    for (;;) {
        i4_t a;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(a) : "v"(addr) : "memory");
        const int a0 = __builtin_amdgcn_readfirstlane(a[0]), a1 = __builtin_amdgcn_readfirstlane(a[1]);
        const int a2 = __builtin_amdgcn_readfirstlane(a[2]), a3 = __builtin_amdgcn_readfirstlane(a[3]);
        if (min(min(skip0 ? a1 : a0, a1), min(a2, a3)) >= target) break;
        FR_K4_WAIT_BLOCKED
        __builtin_amdgcn_s_sleep(1);
    }
    FR_K4_WAIT_END
}
__device__ __forceinline__ void wait1(const int* p, int target)
{
    FR_K4_WAIT_BEGIN
    const unsigned addr = lds_off(p);
    for (;;) {
        int a;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(a) : "v"(addr) : "memory");
        if (__builtin_amdgcn_readfirstlane(a) >= target) break;
        FR_K4_WAIT_BLOCKED
        __builtin_amdgcn_s_sleep(1);
    }
    FR_K4_WAIT_END
}
__device__ __forceinline__ void publish(int* p, int v)
{
    // (LDS operations of one wave execute in order: everything this wave wrote before is visible to a wave that sees v)
    asm volatile("ds_write_b32 %0, %1" : : "v"(lds_off(p)), "v"(v) : "memory");
}

// A row wave's registers: the row's four values in the current micro-panel, and its FINAL values in the two micro-panels before
// it (multipliers of the updates the wave applies itself).  Micro-panel q's final values land in lp[q & 1] -- the set of
// micro-panel q - 2, dead by then -- so the panel loops are unrolled by two and nothing is ever copied.
struct RowState {
    double n[4];
    double lp[2][4];
};
static_assert(LAG == 3, "the row waves keep exactly two finished micro-panels");

// columns of micro-panel q for this lane's row (every update of micro-panels <= q - 3 applied by the update waves), then the
// updates of micro-panels q - 2 and q - 1:  n[c] -= sum_k lp[k] L(4 q + c, 4 (q - g) + k), multipliers broadcast from the L ring
template <int PAR>
__device__ __forceinline__ void load_panel(RowState& s, const double* lds, int stage_off, int row, int q)
{
    const double* st = lds + stage_off + (q & (NSTG - 1)) * PSZ + row;
#pragma unroll
    for (int c = 0; c < 4; ++c) s.n[c] = st[PS * c];
#pragma unroll
    for (int g = 2; g >= 1; --g) {
        if (q >= g) {
            const double (&lp)[4] = s.lp[g == 2 ? PAR : 1 - PAR];
            const double* lr = lds + OFF_L + ((q - g) & (R - 1)) * PSZ + 4 * q;
            double2 m[4][2];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                m[k][0] = *reinterpret_cast<const double2*>(lr + PS * k);
                m[k][1] = *reinterpret_cast<const double2*>(lr + PS * k + 2);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s.n[0] = __builtin_fma(-lp[k], m[k][0].x, s.n[0]);
                s.n[1] = __builtin_fma(-lp[k], m[k][0].y, s.n[1]);
                s.n[2] = __builtin_fma(-lp[k], m[k][1].x, s.n[2]);
                s.n[3] = __builtin_fma(-lp[k], m[k][1].y, s.n[3]);
            }
        }
    }
}

// reciprocal pivots of the micro-panels in the ring: 64 copies of each (the pivot wave writes one per lane: a store of one word
// by all 64 lanes is a 64-way conflict), the readers broadcast copy 0
constexpr int IPS = 4 * 64;  // doubles per micro-panel

// a follower's micro-panel: the 4 x 4 triangle of micro-panel q from the L ring, its reciprocal pivots, then
// l_j = n_j ip_j;  n_c -= l_j L(4 q + c, 4 q + j);  the final values go to dst[]
__device__ __forceinline__ void follow_panel(RowState& s, double (&dst)[4], const double* lds, int q)
{
    const double* lr = lds + OFF_L + (q & (R - 1)) * PSZ + 4 * q;
    const double* ipr = lds + OFF_IP + IPS * (q & (R - 1));
    double2 t[3][2];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        t[k][0] = *reinterpret_cast<const double2*>(lr + PS * k);
        t[k][1] = *reinterpret_cast<const double2*>(lr + PS * k + 2);
    }
    const double ip0 = ipr[0], ip1 = ipr[64], ip2 = ipr[128], ip3 = ipr[192];
    dst[0] = s.n[0] * ip0;
    s.n[1] = __builtin_fma(-dst[0], t[0][0].y, s.n[1]);
    s.n[2] = __builtin_fma(-dst[0], t[0][1].x, s.n[2]);
    s.n[3] = __builtin_fma(-dst[0], t[0][1].y, s.n[3]);
    dst[1] = s.n[1] * ip1;
    s.n[2] = __builtin_fma(-dst[1], t[1][1].x, s.n[2]);
    s.n[3] = __builtin_fma(-dst[1], t[1][1].y, s.n[3]);
    dst[2] = s.n[2] * ip2;
    s.n[3] = __builtin_fma(-dst[2], t[2][1].y, s.n[3]);
    dst[3] = s.n[3] * ip3;
}

// The pivot wave does not store its columns: the OTHER block-row wave does (it follows / idles meanwhile) -- rows rbase ..
// rbase + 63 of micro-panel q from the L ring to A, the diagonal entries as 1 / ip (the reciprocal pivot is 1 / sqrt(d) to an ulp;
// the substitute, the failure NaN and plain-sqrt mode's sqrt(0) = 1 / inf all come out by themselves).
__device__ __forceinline__ void store_panel(const double* lds, int lane, int rbase, int q, double* __restrict__ gcol /* &A[rbase + lane] */, int64_t lda)
{
    const double* rr = lds + OFF_L + (q & (R - 1)) * PSZ + rbase + lane;
    const double* ipr = lds + OFF_IP + IPS * (q & (R - 1));
    double v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = rr[PS * c];
    const double ipv[4] = {ipr[0], ipr[64], ipr[128], ipr[192]};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        double p = __builtin_amdgcn_rcp(ipv[c]);
        p = __builtin_fma(__builtin_fma(-ipv[c], p, 1.0), p, p);
        p = __builtin_fma(__builtin_fma(-ipv[c], p, 1.0), p, p);
        p = (ipv[c] == __builtin_inf()) ? 0.0 : p;  // (rcp(inf) = 0, but the Newton steps make 0 * inf of it)
        gcol[(int64_t)(4 * q + c) * lda] = (rbase + lane == 4 * q + c) ? p : v[c];
    }
}

// one elimination step of the pivot wave: column J = 4 q + j; jl = lane of the micro-panel's first diagonal row; the column's final
// values go to dst[j]
template <bool MODE2, int j>
__device__ __forceinline__ void pivot_step(RowState& s, double (&dst)[4], double alt, double& dd, double& ip, int lane, int jl, double* ring_row,
                                           double* ipr_lane)
{
    const double qv = s.n[j] * ip;
    const bool keep = lane > jl + j;
    const double l = keep ? qv : 0.0;
    PivotChain ch;
    dd = __builtin_fma(-l, l, dd);
    pin(dd);
    ch.d = readlane_f64(dd, (jl + j + 1) & 63);  // next pivot candidate (unused after the wave's last column)
    chain_stage<false, 0>(ch);
    __builtin_amdgcn_sched_barrier(0);
    // column J for the followers and the update waves (zeros on and above the diagonal), its reciprocal pivot beside it
    ring_row[PS * j] = l;
    ipr_lane[64 * j] = ip;
    chain_stage<false, 1>(ch);
    __builtin_amdgcn_sched_barrier(0);
    chain_stage<false, 2>(ch);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (j + 1 < 4) {
        const double m1 = readlane_f64(l, (jl + j + 1) & 63);
        s.n[j + 1] = __builtin_fma(-l, m1, s.n[j + 1]);
        pin(s.n[j + 1]);
    }
    chain_stage<false, 3>(ch);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (j + 2 < 4) {
        const double m2 = readlane_f64(l, (jl + j + 2) & 63);
        s.n[j + 2] = __builtin_fma(-l, m2, s.n[j + 2]);
        pin(s.n[j + 2]);
    }
    chain_stage<false, 4>(ch);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (j + 3 < 4) {
        const double m3 = readlane_f64(l, (jl + j + 3) & 63);
        s.n[j + 3] = __builtin_fma(-l, m3, s.n[j + 3]);
        pin(s.n[j + 3]);
    }
    chain_stage<false, 5>(ch);
    chain_stage<false, 6>(ch);
    dst[j] = l;
    // pivot rule for the next column (which columns it fired on is worked out at the end, from the lanes' d): plain-sqrt mode
    // tests d != 0 (negative and NaN propagate through rsq), the others d > 0
    const bool ok = MODE2 ? (ch.d != 0.0) : (ch.d > 0.0);
    ip = ok ? ch.r : alt;
}

template <bool MODE2, int H, int PAR>
__device__ __forceinline__ void pivot_panel(RowState& s, double* lds, int* prog, int lane, int row, double alt, double& dd, double& ip, int q)
{
    const int jl = 4 * q - 64 * H;
#ifdef FR_K4_TS
    const long long tpw = __builtin_amdgcn_s_memtime();
#endif
    // micro-panel q's columns have come back from the update waves (they are past micro-panel q - LAG) ...
    wait4(prog + 4, q - LAG + 1);
    // ... and the ring slot of micro-panel q - R may be overwritten when every follower is past micro-panel q - R + LAG - 1, the
    // last to read its multipliers; looked at every fourth micro-panel, for the four to come
    if ((q & 3) == 0) wait4(prog, q + 3 - R + LAG);
#ifdef FR_K4_TS
    const long long tp0 = __builtin_amdgcn_s_memtime();
    FR_K4_TADD(6, tp0 - tpw)
#endif
    load_panel<PAR>(s, lds, OFF_FS, row, q);
#ifdef FR_K4_TS
    pin(s.n[0]); pin(s.n[1]); pin(s.n[2]); pin(s.n[3]);
    const long long tp1 = __builtin_amdgcn_s_memtime();
#endif
    double* rr = lds + OFF_L + (q & (R - 1)) * PSZ + row;
    double* ipr = lds + OFF_IP + IPS * (q & (R - 1)) + lane;
    pivot_step<MODE2, 0>(s, s.lp[PAR], alt, dd, ip, lane, jl, rr, ipr);
    pivot_step<MODE2, 1>(s, s.lp[PAR], alt, dd, ip, lane, jl, rr, ipr);
    pivot_step<MODE2, 2>(s, s.lp[PAR], alt, dd, ip, lane, jl, rr, ipr);
    pivot_step<MODE2, 3>(s, s.lp[PAR], alt, dd, ip, lane, jl, rr, ipr);
#ifdef FR_K4_TS
    pin(ip);
    const long long tp2 = __builtin_amdgcn_s_memtime();
    FR_K4_TADD(4, tp1 - tp0) FR_K4_TADD(5, tp2 - tp1)
#endif
    publish(prog + H, q + 1);
}

template <bool MODE2, int H>
__device__ __forceinline__ void pivot_phase(RowState& s, double* lds, int* prog, int lane, int row, double alt, double& dd)
{
    double d = readlane_f64(dd, 0), ip, p0;
    sqrt_rsqrt(d, p0, ip);
    ip = (MODE2 ? (d != 0.0) : (d > 0.0)) ? ip : alt;
    // the chain is the critical path of the launch: its wave wins the issue slot over whichever wave shares its SIMD (an update
    // wave's products, a follower's polls)
    __builtin_amdgcn_s_setprio(3);
    for (int q = 16 * H; q < 16 * H + 16; q += 2) {
        pivot_panel<MODE2, H, 0>(s, lds, prog, lane, row, alt, dd, ip, q);
        pivot_panel<MODE2, H, 1>(s, lds, prog, lane, row, alt, dd, ip, q + 1);
    }
    __builtin_amdgcn_s_setprio(0);
}

// a follower's micro-panel of the block rows 64 .. 127 (wave 1 while wave 0 pivots)
template <int PAR>
__device__ __forceinline__ void d_follow(RowState& s, double* lds, int* prog, int lane, int row, double& dd, int q, double* __restrict__ A,
                                         double* __restrict__ gcol, int64_t lda)
{
    // (the columns and the earlier micro-panels' multipliers are there BEFORE the pivot wave is through micro-panel q: only the
    // 4 x 4 triangle waits for it -- this wave's latency sits between the pivot wave and the update waves)
    wait1(prog + 0, q);
    wait4(prog + 4, q - LAG + 1);
    load_panel<PAR>(s, lds, OFF_FS, row, q);
    wait1(prog + 0, q + 1);
    follow_panel(s, s.lp[PAR], lds, q);
    double* rr = lds + OFF_L + (q & (R - 1)) * PSZ + row;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        dd = __builtin_fma(-s.lp[PAR][c], s.lp[PAR][c], dd);
        rr[PS * c] = s.lp[PAR][c];
    }
    publish(prog + 1, q + 1);  // (the update waves go on; the stores are nobody's dependency)
#pragma unroll
    for (int c = 0; c < 4; ++c) gcol[(int64_t)(4 * q + c) * lda] = s.lp[PAR][c];
    store_panel(lds, lane, 0, q, A + lane, lda);
}

// Rows 64 H .. 64 H + 63 of the block: pivot wave for its own 64 columns; before that (H = 1) follower of the other wave's columns,
// after it (H = 0) nothing of its own is left -- in both cases it also stores the pivoting wave's columns
template <int H>
__device__ __forceinline__ void d_wave(double* lds, int* prog, int lane, double* __restrict__ A, int64_t lda, bool mode2, double alt, double& dd)
{
    const int row = 64 * H + lane;
    RowState s;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int c = 0; c < 4; ++c) s.lp[g][c] = 0.0;
    dd = A[row + (int64_t)row * lda];
    double* gcol = A + row;
    if constexpr (H == 1) {
        for (int q = 0; q < 16; q += 2) {
            d_follow<0>(s, lds, prog, lane, row, dd, q, A, gcol, lda);
            d_follow<1>(s, lds, prog, lane, row, dd, q + 1, A, gcol, lda);
        }
    }
#ifdef FR_K4_TS
    const long long tpl = __builtin_amdgcn_s_memtime();
#endif
    if (mode2) pivot_phase<true, H>(s, lds, prog, lane, row, alt, dd);
    else pivot_phase<false, H>(s, lds, prog, lane, row, alt, dd);
#ifdef FR_K4_TS
    FR_K4_TADD(7, __builtin_amdgcn_s_memtime() - tpl)
#endif
    if constexpr (H == 0) {
        // the other wave pivots now: its columns 64 .. 127 go to memory from here, with zeros in rows 0 .. 63 (above the diagonal:
        // refinement uses L_bb as a plain 128 x 128 operand)
        for (int q = 16; q < 32; ++q) {
            wait1(prog + 1, q + 1);
            store_panel(lds, lane, 64, q, A + 64 + lane, lda);
            publish(prog + 0, q + 1);
#pragma unroll
            for (int c = 0; c < 4; ++c) gcol[(int64_t)(4 * q + c) * lda] = 0.0;
        }
    }
}

// a micro-panel of rows 64 G .. 64 G + 63 of X^T = I L^-T
template <int G, int PAR>
__device__ __forceinline__ void x_follow(RowState& s, double* lds, int* prog, int row, int q, double* __restrict__ gi, double& wmax, double& nanacc)
{
    // (rows 4 q .. 4 q + 3 of the earlier micro-panels -- the multipliers -- belong to the wave that pivots micro-panel q)
    const int* pv = prog + (q < 16 ? 0 : 1);
    wait1(pv, q);
    wait4(prog + 4, q - LAG + 1);
    load_panel<PAR>(s, lds, OFF_XS, row, q);
    wait1(pv, q + 1);
    follow_panel(s, s.lp[PAR], lds, q);
    double* rr = lds + OFF_Y + (q & (R - 1)) * PSZ + row;
#pragma unroll
    for (int c = 0; c < 4; ++c) rr[PS * c] = s.lp[PAR][c];
    publish(prog + 2 + G, q + 1);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        wmax = __builtin_fmax(wmax, __builtin_fabs(s.lp[PAR][c]));
        nanacc = __builtin_fma(s.lp[PAR][c], 0.0, nanacc);
    }
    *reinterpret_cast<double2*>(gi + 4 * q) = double2{s.lp[PAR][0], s.lp[PAR][1]};
    *reinterpret_cast<double2*>(gi + 4 * q + 2) = double2{s.lp[PAR][2], s.lp[PAR][3]};
}

// Rows 64 G .. 64 G + 63 of X^T = I L^-T
template <int G>
__device__ __forceinline__ void x_wave(double* lds, int* prog, int lane, double* __restrict__ inv, int64_t ldinv, double& wmax, double& nanacc)
{
    const int row = 64 * G + lane;
    RowState s;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int c = 0; c < 4; ++c) s.lp[g][c] = 0.0;
    double* gi = inv + (int64_t)row * ldinv;  // column `row` of the inverse: W(j, row) = X^T(row, j)
    if constexpr (G == 1) {
        // columns 0..63 of these rows are zero (above the diagonal of X^T's transpose): stored while the wave has nothing to do
        for (int j = 0; j < 64; j += 2) *reinterpret_cast<double2*>(gi + j) = double2{0.0, 0.0};
    }
    for (int q = 16 * G; q < 32; q += 2) {
        x_follow<G, 0>(s, lds, prog, row, q, gi, wmax, nanacc);
        x_follow<G, 1>(s, lds, prog, row, q + 1, gi, wmax, nanacc);
    }
}

// ---- update waves ----------------------------------------------------------------------------------------------------
template <int U, int I>
__device__ __forceinline__ void u_init(d4_t (&acc)[18], const double* __restrict__ A, int64_t lda, int l15, int lq)
{
    if constexpr (I < 18) {
        constexpr Tile T = TILES[U][I];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int r = 16 * T.rt + l15, c = 16 * T.ct + lq + 4 * g;
            if constexpr (T.kind == 0) acc[I][g] = -A[r + (int64_t)c * lda];
            else acc[I][g] = (r == c) ? -1.0 : 0.0;
        }
        u_init<U, I + 1>(acc, A, lda, l15, lq);
    }
}

// micro-panel P applied to the wave's live tiles (first operand <-> column of the tile, second <-> row)
// (FIRST = true: the tiles whose columns are handed back after this micro-panel; false: the others -- their products may still be
// in the matrix pipe while the hand-back is written)
template <int U, int P, bool FIRST, int I>
__device__ __forceinline__ void u_apply(d4_t (&acc)[18], const double* lring, const double* yring)
{
    if constexpr (I < 18) {
        constexpr Tile T = TILES[U][I];
        if constexpr (tile_live(T, P) && ((T.ct == (P + LAG) / 4) == FIRST)) {
            const double a = lring[16 * T.ct];
            const double b = (T.kind == 0) ? lring[16 * T.rt] : yring[16 * T.rt];
            if (K4X != 6) acc[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[I], 0, 0, 0);
            else acc[I][0] += a + b;  // (experiment 6: no matrix-core work: wrong results, chain pace only)
        }
        u_apply<U, P, FIRST, I + 1>(acc, lring, yring);
    }
}

// columns of micro-panel Q = P + LAG (tile column Q / 4, accumulator register Q % 4) handed back through the staging buffers
template <int U, int Q, int I>
__device__ __forceinline__ void u_extract(const d4_t (&acc)[18], double* fs, double* xs)
{
    if constexpr (I < 18) {
        constexpr Tile T = TILES[U][I];
        if constexpr (T.ct == Q / 4) {
            double* dst = (T.kind == 0) ? fs : xs;
            dst[16 * T.rt] = -acc[I][Q % 4];
        }
        u_extract<U, Q, I + 1>(acc, fs, xs);
    }
}

template <int U, int P>
__device__ __forceinline__ void u_panels(d4_t (&acc)[18], double* lds, int* prog, int lane_off)
{
    if constexpr (P <= 30) {
#ifdef FR_K4_TS
        const long long tu0 = __builtin_amdgcn_s_memtime();
#endif
        // (from micro-panel 16 on, wave 0 only stores the other wave's columns: its counter guards the ring, nobody's operands)
        wait4(prog, P + 1, P >= 16);
#ifdef FR_K4_TS
        const long long tu1 = __builtin_amdgcn_s_memtime();
#endif
        const double* lring = lds + OFF_L + (P & (R - 1)) * PSZ + lane_off;
        const double* yring = lds + OFF_Y + (P & (R - 1)) * PSZ + lane_off;
        u_apply<U, P, true, 0>(acc, lring, yring);
        u_apply<U, P, false, 0>(acc, lring, yring);
#ifdef FR_K4_TS
        const long long tu2 = __builtin_amdgcn_s_memtime();
#endif
        constexpr int Q = P + LAG;
        if constexpr (Q < 32) {
            double* fs = lds + OFF_FS + (Q & (NSTG - 1)) * PSZ + lane_off;
            double* xs = lds + OFF_XS + (Q & (NSTG - 1)) * PSZ + lane_off;
            u_extract<U, Q, 0>(acc, fs, xs);
        }
        publish(prog + 4 + U, P + 1);
#ifdef FR_K4_TS
        const long long tu3 = __builtin_amdgcn_s_memtime();
        FR_K4_TADD(6, tu1 - tu0) FR_K4_TADD(4, tu2 - tu1) FR_K4_TADD(5, tu3 - tu2)
#endif
        u_panels<U, P + 1>(acc, lds, prog, lane_off);
    }
}

template <int U>
__device__ __forceinline__ void u_wave(double* lds, int* prog, int lane, const double* __restrict__ A, int64_t lda)
{
    const int l15 = lane & 15, lq = lane >> 4;
    d4_t acc[18];
    u_init<U, 0>(acc, A, lda, l15, lq);
    u_panels<U, 0>(acc, lds, prog, l15 + PS * lq);
    publish(prog + 4 + U, 64);
}

}  // namespace flat

// The flat kernel's body: one full 128 x 128 block at A, its inverse to inv; all PT threads of the workgroup, `lds` = the
// workgroup's flat::LDS_BYTES.  Called once by potf2_flat_kernel and once per diagonal block by the resident panel-chain kernel
// (panel_chain_kernel below), which is why it is a function: the LDS image is rebuilt from scratch on every call.
__device__ __forceinline__ void potf2_flat_body(double* __restrict__ lds, double* __restrict__ A, int64_t lda, int64_t col0, int mode,
                                                double sub, double* __restrict__ inv, int64_t ldinv, int64_t* __restrict__ info,
                                                double* __restrict__ cest, unsigned* __restrict__ xcc_word, const int t)
{
    using namespace flat;
    const int lane = t & 63;
    const int hw = __builtin_amdgcn_readfirstlane(t >> 6);
    // role of hardware wave hw: 0 / 1 the block's row waves, 2 / 3 the rows of X^T, 4 .. 7 the update waves.  Waves hw and hw + 4
    // of a workgroup share a SIMD (observed on every run, not promised): the pivot waves are paired with the X^T followers and the
    // update waves with each other -- an update wave's FP64 products on the pivot wave's SIMD hold its dependent f64 operations up
    // (38.2 -> 34.2 us; with no products at all 33.6; experiment 8 = roles in wave order)
    const int w = (K4X != 8) ? ((hw & 1) | ((hw & 2) << 1) | ((hw & 4) >> 1)) : hw;
    if (xcc_word && t == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        __hip_atomic_store(xcc_word, (xcc & 7u) + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    int* prog = reinterpret_cast<int*>(lds + OFF_END);                                  // 8 progress counters
    unsigned long long* excs = reinterpret_cast<unsigned long long*>(lds + OFF_END + 4);  // 2 exception masks
    double* red = lds + OFF_END + 6;                                                    // 8 reduction slots
    // ---- prologue: micro-panels 0 .. LAG - 1 straight into the staging slots, zeros where a wave reads before anyone wrote
    if (w < 2) {
        const int r = 64 * w + lane;
        double v[4 * LAG];
#pragma unroll
        for (int c = 0; c < 4 * LAG; ++c) v[c] = A[r + (int64_t)c * lda];
#pragma unroll
        for (int c = 0; c < 4 * LAG; ++c) lds[OFF_FS + (c >> 2) * PSZ + r + PS * (c & 3)] = v[c];
    } else if (w < 4) {
        const int r = 64 * (w - 2) + lane;
#pragma unroll
        for (int c = 0; c < 4 * NSTG; ++c) lds[OFF_XS + (c >> 2) * PSZ + r + PS * (c & 3)] = (r == c && c < 4 * LAG) ? 1.0 : 0.0;
#pragma unroll
        for (int sl = 0; sl < R; ++sl)
#pragma unroll
            for (int c = 0; c < 4; ++c) lds[OFF_Y + sl * PSZ + r + PS * c] = 0.0;
    }
    if (t < 8) prog[t] = (t == 3) ? 16 : 0;
#ifdef FR_K4_TS
    if (t < 64) reinterpret_cast<unsigned long long*>(lds + OFF_IP + 256 * R)[t] = 0ull;
#endif
    lds_barrier();

    const bool mode2 = (mode == 2);
    double p_exc = __builtin_nan(""), ip_exc = p_exc;
    const bool substitute = (mode == 1 && sub > 0.0);
    if (substitute) sqrt_rsqrt(sub, p_exc, ip_exc);
    // what the reciprocal pivot becomes when the rule's test fails: 1 / sqrt(substitute) or NaN (failure); plain-sqrt mode
    // (add_rows): +inf, the reference's division by sqrt(0)
    const double alt = mode2 ? __builtin_inf() : ip_exc;
    double wmax = 0.0, nanacc = 0.0, dmin = __builtin_inf();
#ifdef FR_K4_TS
    const long long trole0 = __builtin_amdgcn_s_memtime();
    (void)0;
#endif
    double dd = 0.0;
    if (w == 0) d_wave<0>(lds, prog, lane, A, lda, mode2, alt, dd);
    else if (w == 1) d_wave<1>(lds, prog, lane, A, lda, mode2, alt, dd);
    else if (w == 2) x_wave<0>(lds, prog, lane, inv, ldinv, wmax, nanacc);
    else if (w == 3) x_wave<1>(lds, prog, lane, inv, ldinv, wmax, nanacc);
    else if (w == 4) u_wave<0>(lds, prog, lane, A, lda);
    else if (w == 5) u_wave<1>(lds, prog, lane, A, lda);
    else if (w == 6) u_wave<2>(lds, prog, lane, A, lda);
    else u_wave<3>(lds, prog, lane, A, lda);

#ifdef FR_K4_TS
    if (lane == 0) { unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid)); const long long tr = __builtin_amdgcn_s_memtime() - trole0;
        for (int i = 0; i < 8; ++i) k4ts[8 * w + i] = (i == 0) ? tr : ((i == 3) ? (long long)((hwid >> 4) & 3) : (long long)*FR_K4_TSLOT(i)); }
#endif
    // ---- epilogue: the diagonal of the factor from the recorded d, the log, the conditioning estimate
    if (w < 2) {
        const int r = 64 * w + lane;
        // a lane's accumulator dd stayed d_r, the value its pivot was taken from, once its column had passed
        const double d = dd;
        double p, ipd;
        sqrt_rsqrt(d, p, ipd);
        const bool bad = !(mode == 2 || d > 0.0);
        ipd = bad ? ip_exc : ipd;
        dmin = __builtin_fabs(ipd);
        dmin = (dmin != dmin) ? -1.0 : dmin;  // (NaN marker: resolved below)
        const unsigned long long exc = __ballot(bad);
        if (lane == 0) excs[w] = exc;
    }
    // wave reductions: max |W| (NaN sticks through nanacc), min |W_ii|
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        wmax = __builtin_fmax(wmax, __shfl_xor(wmax, off, 64));
        nanacc += __shfl_xor(nanacc, off, 64);
        dmin = __builtin_fmin(dmin, __shfl_xor(dmin, off, 64));
    }
    if (lane == 0) red[w] = (w < 2) ? dmin : ((w < 4) ? ((nanacc != nanacc) ? nanacc : wmax) : 0.0);
    lds_barrier();
    if (t == 0) {
        const unsigned long long e0 = excs[0], e1 = excs[1];
        if (e0 | e1) {
            if (substitute) {
                int64_t qn = info[1];
                for (int j = 0; j < 64; ++j)
                    if ((e0 >> j) & 1ull) info[3 + qn++] = col0 + j;
                for (int j = 0; j < 64; ++j)
                    if ((e1 >> j) & 1ull) info[3 + qn++] = col0 + 64 + j;
                info[1] = qn;
            } else if (info[0] == 0) {
                const int first = e0 ? (__builtin_ffsll((long long)e0) - 1) : (64 + __builtin_ffsll((long long)e1) - 1);
                info[0] = 1 + col0 + first;
            }
        }
        if (cest) {
            const double dm = __builtin_fmin(red[0], red[1]);
            const double wa = red[2], wb = red[3];
            double wm = (wa != wa) ? wa : ((wb != wb) ? wb : __builtin_fmax(wa, wb));
            *cest = (dm < 0.0) ? __builtin_nan("") : wm / dm;
        }
    }
}

__global__ __launch_bounds__(PT, 2) void potf2_flat_kernel(double* __restrict__ A, int64_t lda, int n, int64_t col0, int mode,
                                                        double sub, double* __restrict__ inv, int64_t ldinv,
                                                        int64_t* __restrict__ info, double* __restrict__ cest,
                                                        unsigned* __restrict__ xcc_word)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    (void)n;
    potf2_flat_body(lds, A, lda, col0, mode, sub, inv, ldinv, info, cest, xcc_word, (int)threadIdx.x);
}

// =====================================================================================================================
// Resident panel chain (round 6): the kb x kb diagonal block of a panel (kb = 128 J, J <= 4) -- and, optionally, rows below it --
// factored by ONE launch instead of J diagonal-block kernels + 2 J - 2 launch-bound products (+ the one-launch solve of the rows
// below).  Between two diagonal-block kernels of a stream sat two chip-wide launches whose matrix-core time is 3.4 us and whose
// launch-bound time is 12.6 - 13.6 us each ("what the chain costs now"): 43 % of the chain at N = 4096, 40 %
// of the sharded schedule's D_p.  Here the workgroups are resident and hand over through memory:
//
//   workgroup 0        the flat diagonal-block kernel's body (potf2_flat_body), once per 128-block j: waits until the eight slabs
//                      of tile row j have published their rows of D_j, factors, writes L_jj and W_j = L_jj^-1, publishes "W_j".
//   workgroup 1 + i    SLAB i: rows 128 + 16 i .. + 15 of the panel (tile row t = 1 + i / 8), resident in LDS as a strip of
//                      16 x 128 (min(t, J) + 1) doubles for the whole launch (the layout of rows_solve16_kernel, gemm_f64.hip:
//                      right operands as matrix-core fragments straight from memory).  LEFT-looking, everything that does not need
//                      the newest inverse done before it arrives.  Per sub-panel q < min(t, J):
//                          wait W_q;  S_q <- S_q W_q^T  (S_q already carries the updates of the sub-panels before);  store, publish;
//                          t < J: wait for the eight slabs of MY tile row (their S_q): D_t rows -= S_q S_{t,q}^T; after q = t - 1 the
//                                 rows of D_t are stored and published -- workgroup 0 takes over, the slab retires;
//                          for every later sub-panel s: wait for tile row s's S_q:  S_s -= S_q L[s, q]^T.
//                      Slabs of rows below the diagonal block (t >= J) only solve; nobody waits for them.
//
// The chain between W_q and the start of diagonal block q + 1 is then: flag -> one 16 x 128 x 128 product -> store + flag -> one
// product -> store + flag, on the eight slabs of tile row q + 1 side by side.  Hand-offs are the agent-scope release / acquire of
// handoff.hpp (placement-independent; every spin bounded, a timed-out launch raises the context's status word and the factorisation
// is repeated on the launch chain).  Flags carry epoch * 16 + progress, so nothing is zeroed between launches.
// Deterministic: every element is computed by one workgroup in a fixed order.  A slab only ever waits for workgroup 0 and for slabs
// of tile rows <= its own, and the grid is small (1 + 8 (J - 1) + extra rows / 16 workgroups of one CU each): the launcher
// keeps it below the CU count, the dispatcher places workgroups in order.
static inline int64_t imin64(int64_t a, int64_t b) { return a < b ? a : b; }
static inline int64_t imax64(int64_t a, int64_t b) { return a > b ? a : b; }

namespace chain {

constexpr int SSTR = 516;  // doubles per strip row (516 mod 32 = 4: rows 0..7 x four k's hit 32 distinct 8-byte banks)
constexpr int NFLAGS = 8 + 8 * 3;  // [0] = W progress; [8 + 8 (t - 1) + i] = slab i of tile row t (t = 1 .. 3)
constexpr size_t SLAB_LDS = sizeof(double) * 16 * SSTR;

struct Args {
    double* A;       // element (0, 0) of the diagonal block
    int64_t lda;
    int64_t rows;    // rows of the panel handled by this launch, counted from the block's first row (>= 128 J)
    int J;           // 128-blocks on the diagonal
    int64_t col0;    // global column of the block (log entries, failure column)
    int mode;
    double sub;
    double* dinv;    // J inverse blocks (ld 128)
    int64_t* info;
    double* cest;    // J estimates (may be NULL)
    int* flags;
    int base;        // epoch * 16
    unsigned* status;
    unsigned* xcc_word;
    int place_r;     // XCD-level reservation in force: only the workgroups with blockIdx % 8 < place_r work (they run on the panel stream's
                     // XCDs: workgroup b of a launch runs on XCD (X + b) % 8), the others exit at once; 0: every workgroup works
    int nwork;       // working workgroups
    int bulk_groups, bulk_workers;  // (bulk_mt != 0) row groups of 16 bulk_mt rows below the diagonal block, workgroups that share them
    int bulk_mt;     // rows below the diagonal block: 0 = resident 16-row slabs like the block's own; 2 / 4 = workgroups of 32 / 64 rows (bulk_role)
    unsigned long long* ts;  // developer stamps (FRIEDRICH_AMD_CHAIN_TS=1, read through the counters "chain_ts:<i>"): 100 MHz wall clock; NULL in normal operation
};

// stamp slot layout: workgroup 0: [8 j + 0] D_j seen, [8 j + 1] block j factored, [8 j + 2] W_j published;
// first slab of tile row t: [32 + 16 t + 4 q + 0] W_q seen, [+ 1] S_q published, [+ 2] tile row's S_q seen, [+ 3] product done / D published
__device__ __forceinline__ void stamp(const Args& a, int slot)
{
    if (a.ts && threadIdx.x == 0) a.ts[slot] = wall_clock64();
}

// strip[:, oc .. oc + 128) (op)= strip[:, lc .. lc + 128) X^T, X element (n, k) at X[n + k ldx]; a wave owns 16 result columns
template <bool SUB>
__device__ __forceinline__ void slab_product(double* __restrict__ strip, int lc, const double* __restrict__ X, int64_t ldx, int oc, int wave,
                                             int l15, int lq)
{
    const double* p = X + (16 * wave + l15) + (int64_t)lq * ldx;
    double rb[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) rb[u] = p[(int64_t)(4 * u) * ldx];
    const double* left = strip + l15 * SSTR + lc + lq;
    // (even and odd k-groups on accumulators of their own: dependent matrix-core instructions do not issue back to back)
    d4_t a0 = {0.0, 0.0, 0.0, 0.0}, a1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int u = 0; u < 32; ++u) {
        const double av = left[4 * u];
        if (u & 1) a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(rb[u], av, a1, 0, 0, 0);
        else a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(rb[u], av, a0, 0, 0, 0);
    }
    __syncthreads();  // every wave has read the left operand before anybody overwrites it (oc == lc: the solve)
    a0 += a1;
    double* out = strip + l15 * SSTR + oc + 16 * wave + lq;
#pragma unroll
    for (int r = 0; r < 4; ++r) out[4 * r] = SUB ? out[4 * r] - a0[r] : a0[r];
    __syncthreads();
}

// columns [c0, c0 + 128) of the strip -> rows r0 .. r0 + 15 of A (a column is 128 contiguous bytes)
__device__ __forceinline__ void slab_store(const double* __restrict__ strip, double* __restrict__ A, int64_t lda, int64_t r0, int64_t rows, int c0)
{
    const int t = threadIdx.x, r = t & 15;
    if (r0 + r < rows) {
        double* dst = A + (r0 + r) + (int64_t)(c0 + (t >> 4)) * lda;
        const double* src = strip + r * SSTR + c0 + (t >> 4);
#pragma unroll
        for (int c = 0; c < 128; c += 32) dst[(int64_t)c * lda] = src[c];
    }
}

__device__ __forceinline__ void slab_role(double* __restrict__ strip, const Args& a, int slab)
{
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int64_t r0 = 128 + 16 * (int64_t)slab;
    const int trow = (int)(r0 >> 7);
    const int nsolve = trow < a.J ? trow : a.J;
    const bool diag = trow < a.J;
    const int ncols = 128 * (nsolve + (diag ? 1 : 0));
    {
        // (several loads in flight per thread)
        const int r = t & 15;
        const bool rok = r0 + r < a.rows;
        const double* src = a.A + (r0 + r) + (int64_t)(t >> 4) * a.lda;
        double* dst = strip + r * SSTR + (t >> 4);
        for (int cb = 0; cb < ncols; cb += 128) {
            double v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = rok ? src[(int64_t)(cb + 32 * i) * a.lda] : 0.0;
#pragma unroll
            for (int i = 0; i < 4; ++i) dst[cb + 32 * i] = v[i];
        }
    }
    __syncthreads();
    int* myflag = a.flags + 8 + 8 * (trow - 1) + (slab & 7);
    const bool st = (slab & 7) == 0 && trow < 4;
    for (int q = 0; q < nsolve; ++q) {
        if (!handoff_wait_ge(a.flags, a.base + q + 1, a.status)) return;
        if (st) stamp(a, 32 + 16 * trow + 4 * q + 0);
        slab_product<false>(strip, 128 * q, a.dinv + (int64_t)q * (128 * 128), 128, 128 * q, wave, l15, lq);
        slab_store(strip, a.A, a.lda, r0, a.rows, 128 * q);
        if (diag) {
            handoff_publish(myflag, a.base + q + 1);
            if (st) stamp(a, 32 + 16 * trow + 4 * q + 1);
            // the rows of my tile row in sub-panel q, from all eight slabs: my share of D_t
            if (!handoff_wait_all_ge(a.flags + 8 + 8 * (trow - 1), 8, a.base + q + 1, a.status)) return;
            if (st) stamp(a, 32 + 16 * trow + 4 * q + 2);
            slab_product<true>(strip, 128 * q, a.A + 128 * trow + (int64_t)(128 * q) * a.lda, a.lda, 128 * trow, wave, l15, lq);
            if (q == trow - 1) {
                slab_store(strip, a.A, a.lda, r0, a.rows, 128 * trow);
                handoff_publish(myflag, a.base + trow + 1);
                if (st) stamp(a, 32 + 16 * trow + 4 * q + 3);
                return;
            }
            if (st) stamp(a, 32 + 16 * trow + 4 * q + 3);
        }
        for (int s = q + 1; s < nsolve; ++s) {
            if (!handoff_wait_all_ge(a.flags + 8 + 8 * (s - 1), 8, a.base + q + 1, a.status)) return;
            slab_product<true>(strip, 128 * q, a.A + 128 * s + (int64_t)(128 * q) * a.lda, a.lda, 128 * s, wave, l15, lq);
        }
    }
}

// Rows BELOW the diagonal block, 16 MT of them per workgroup (MT = 2, 4): nobody inside the launch waits for them, so they need not
// keep up with the chain row for row -- what they must not do is hold a CU each for the length of the panel at a quarter of its
// matrix-core time (the resident 16-row slabs: N = 4096, 224 of them next to the trailing update of the panel before).  RIGHT-looking
// over the sub-panels: step q loads the rows' sub-panel q (it carries every earlier update) into LDS, waits for W_q, solves in
// place, stores, and applies it to the later sub-panels of its rows as read-modify-writes of memory -- a wave owns 16 result columns
// and all MT row tiles, so one right-operand fragment feeds MT matrix-core instructions.  The workgroups wait for workgroup 0 and
// for the diagonal block's slabs only (lower block indices), so any number of them is safe whatever the residency.
constexpr int BSTR = 132;  // doubles per row of the LDS tile (132 mod 32 = 4)

// `worker` of `a.bulk_workers` takes the row groups worker, worker + bulk_workers, ... through every step: with fewer workers than
// groups a workgroup is busy for most of the panel instead of waiting for the next inverse three quarters of the time -- the
// launcher sizes the workers to the CUs the trailing update leaves to the panel stream.
template <int MT>
__device__ __forceinline__ void bulk_role(double* __restrict__ S, const Args& a, int worker)
{
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    constexpr int ROWS = 16 * MT, CPP = PT / ROWS;  // columns per pass of the loaders
    const int lr = t % ROWS, lc = t / ROWS;
    for (int q = 0; q < a.J; ++q)
    for (int grp = worker; grp < a.bulk_groups; grp += a.bulk_workers) {
        const int64_t r0 = 128 * (int64_t)a.J + (int64_t)ROWS * grp;
        const bool rok = r0 + lr < a.rows;
        {
            const double* src = a.A + (r0 + lr) + (int64_t)(128 * q + lc) * a.lda;
            double* dst = S + lr * BSTR + lc;
#pragma unroll 4
            for (int c = 0; c < 128; c += CPP) dst[c] = rok ? src[(int64_t)c * a.lda] : 0.0;
        }
        if (!handoff_wait_ge(a.flags, a.base + q + 1, a.status)) return;  // (its barrier also closes the tile's image)
        for (int s = q; s < a.J; ++s) {
            const bool solve = s == q;
            if (!solve && !handoff_wait_all_ge(a.flags + 8 + 8 * (s - 1), 8, a.base + q + 1, a.status)) return;
            const double* X = solve ? a.dinv + (int64_t)q * (128 * 128) : a.A + 128 * s + (int64_t)(128 * q) * a.lda;
            const int64_t ldx = solve ? 128 : a.lda;
            const double* p = X + (16 * wave + l15) + (int64_t)lq * ldx;
            double rb[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) rb[u] = p[(int64_t)(4 * u) * ldx];
            d4_t acc[MT][2];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt][0] = acc[mt][1] = d4_t{0.0, 0.0, 0.0, 0.0};
            const double* left = S + l15 * BSTR + lq;
#pragma unroll
            for (int u = 0; u < 32; ++u)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt][u & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(rb[u], left[16 * mt * BSTR + 4 * u], acc[mt][u & 1], 0, 0, 0);
            if (solve) {
                __syncthreads();
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const d4_t v = acc[mt][0] + acc[mt][1];
                    double* out = S + (16 * mt + l15) * BSTR + 16 * wave + lq;
#pragma unroll
                    for (int r = 0; r < 4; ++r) out[4 * r] = v[r];
                }
                __syncthreads();
                if (rok) {
                    double* dst = a.A + (r0 + lr) + (int64_t)(128 * q + lc) * a.lda;
                    const double* src = S + lr * BSTR + lc;
#pragma unroll 4
                    for (int c = 0; c < 128; c += CPP) dst[(int64_t)c * a.lda] = src[c];
                }
            } else {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const d4_t v = acc[mt][0] + acc[mt][1];
                    const int64_t row = r0 + 16 * mt + l15;
                    if (row < a.rows) {
                        double* g = a.A + row + (int64_t)(128 * s + 16 * wave + lq) * a.lda;
#pragma unroll
                        for (int r = 0; r < 4; ++r) g[(int64_t)(4 * r) * a.lda] -= v[r];
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // this workgroup's stores, drained, before its own loads of the same rows in the next step
    }
}

}  // namespace chain

__global__ __launch_bounds__(PT, 2) void panel_chain_kernel(const chain::Args a)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    int wg = (int)blockIdx.x;
    if (a.place_r) {
        if ((wg & 7) >= a.place_r) return;
        wg = (wg >> 3) * a.place_r + (wg & 7);
    }
    if (wg >= a.nwork) return;
    if (wg != 0) {
        const int slab = wg - 1, ndiag = 8 * (a.J - 1);
        if (a.bulk_mt == 0 || slab < ndiag) chain::slab_role(lds, a, slab);
        else if (a.bulk_mt == 2) chain::bulk_role<2>(lds, a, slab - ndiag);
        else chain::bulk_role<4>(lds, a, slab - ndiag);
        return;
    }
#pragma nounroll
    for (int j = 0; j < a.J; ++j) {
        if (j > 0 && !handoff_wait_all_ge(a.flags + 8 + 8 * (j - 1), 8, a.base + j + 1, a.status)) return;
        chain::stamp(a, 8 * j + 0);
        // (the body's per-lane addresses are formed from an opaque copy of the thread index and of the LDS base: hoisted out of this
        // loop they would stay live across the whole body -- 32 registers spilled, measured with -Rpass-analysis)
        int tid = (int)threadIdx.x;
        asm volatile("" : "+v"(tid));
        potf2_flat_body(lds, a.A + (int64_t)(128 * j) * (a.lda + 1), a.lda, a.col0 + 128 * j, a.mode, a.sub, a.dinv + (int64_t)j * (128 * 128), 128,
                        a.info, a.cest ? a.cest + j : nullptr, j == 0 ? a.xcc_word : nullptr, tid);
        chain::stamp(a, 8 * j + 1);
        handoff_publish(a.flags, a.base + j + 1);  // (its barrier also closes the block's LDS image before the next one is built)
        chain::stamp(a, 8 * j + 2);
    }
}

// The kb x kb diagonal block at A (kb = 128 J, 2 <= J <= 4) factored with its J inverse blocks, and rows - kb further rows below it
// solved against it, in ONE launch.  FR_UNSUPPORTED_KERNEL-like contract: returns 1 when the shape is not taken (the caller keeps
// the launch chain), FR_OK (0) when launched, an error code otherwise.
int launch_panel_chain(fr_ctx* ctx, double* A, int64_t lda, int64_t kb, int64_t rows, int64_t col0, int mode, double sub, double* dinv,
                       int64_t* info, double* cest)
{
    if (kb % PB != 0 || kb < 2 * PB || kb > 4 * PB || rows < kb || mode == 3 || !dinv || !info) return 1;
    // rows below the diagonal block: resident 16-row slabs while they are few (they finish 4 us behind the last diagonal block), 32- or
    // 64-row workgroups beyond (6 - 10 us behind it, a half / a quarter of the CUs)
    const int64_t below = rows - kb;
    static const int force_mt = getenv("FRIEDRICH_AMD_CHAIN_BULK") ? atoi(getenv("FRIEDRICH_AMD_CHAIN_BULK")) : -1;
    const int bulk_mt = force_mt >= 0 ? force_mt : (below <= 512 ? 0 : (below <= 2048 ? 2 : 4));
    const int64_t ndiag = (kb - PB) / 16;
    const int64_t groups = bulk_mt == 0 ? 0 : (below + 16 * bulk_mt - 1) / (16 * bulk_mt);
    // workers for the rows below: one per group.  (Measured, N = 8192: 8 workers for ~90 groups -- what a one-unit reservation leaves
    // beside the diagonal block's 25 workgroups -- make a panel 1.7 ms instead of 0.18: a worker's products are latency-bound, 16 us
    // each where the matrix core needs 6.8, so the rows below want many workgroups in flight, not few busy ones.)
    int64_t workers = groups;
    static const int force_workers = getenv("FRIEDRICH_AMD_CHAIN_WORKERS") ? atoi(getenv("FRIEDRICH_AMD_CHAIN_WORKERS")) : 0;
    if (force_workers > 0) workers = imin64(groups, force_workers);
    const int64_t nslab = ndiag + (bulk_mt == 0 ? (below + 15) / 16 : workers);
    // (the diagonal block's own workgroups take a CU each and wait for one another: they must be resident together; the others only
    // wait for lower block indices)
    if (1 + (bulk_mt == 0 ? nslab : ndiag) > ctx->num_cus - 8) return 1;
    FR_TRY(ensure_status_word(ctx));
    if (!ctx->chain_flags) {
        FR_HIP(ctx, dev_malloc(ctx, (void**)&ctx->chain_flags, sizeof(int) * chain::NFLAGS * kChainRing));
        FR_HIP(ctx, hipMemsetAsync(ctx->chain_flags, 0, sizeof(int) * chain::NFLAGS * kChainRing, ctx->ls));
        FR_HIP(ctx, hipStreamSynchronize(ctx->ls));
        ctx->chain_epoch = 0;
    }
    if (!ctx->chain_lds_set) {
        FR_TRY(set_dyn_lds(ctx, reinterpret_cast<const void*>(panel_chain_kernel), (int)flat::LDS_BYTES));
        ctx->chain_lds_set = true;
    }
    static_assert(flat::LDS_BYTES >= chain::SLAB_LDS, "the slabs' strip fits in the diagonal-block kernel's LDS");
    if (ctx->chain_epoch >= (1 << 26)) {  // (flags carry epoch * 16 + progress in an int)
        FR_HIP(ctx, hipDeviceSynchronize());
        FR_HIP(ctx, hipMemset(ctx->chain_flags, 0, sizeof(int) * chain::NFLAGS * kChainRing));
        ctx->chain_epoch = 0;
    }
    const int64_t e = ++ctx->chain_epoch;
    chain::Args a;
    a.A = A; a.lda = lda; a.rows = rows; a.J = (int)(kb / PB); a.col0 = col0; a.mode = mode; a.sub = sub; a.dinv = dinv; a.info = info;
    a.cest = cest;
    a.flags = ctx->chain_flags + chain::NFLAGS * (e % kChainRing);
    a.base = (int)(e * 16);
    a.status = ctx->dev_status;
    a.xcc_word = ctx->xcd_reserve != 0 ? ctx->xcc_word : nullptr;
    a.bulk_mt = bulk_mt;
    a.bulk_groups = (int)groups;
    a.bulk_workers = (int)(workers > 0 ? workers : 1);
    a.nwork = (int)(1 + nslab);
    // by XCDs (at most 4096 trailing rows, gemm_f64.hip): the launch carries 8 / R times the workgroups, those dealt to the panel
    // stream's XCDs work.  By CUs the trailing update keeps CUs free everywhere: the workgroups go where there is room.
    a.place_r = (ctx->reserve_now > 0 && !ctx->reserve_by_cu_now && ctx->ls == ctx->stream2 && ctx->world == 1) ? ctx->reserve_now : 0;
    const unsigned grid = a.place_r ? (unsigned)((a.nwork + a.place_r - 1) / a.place_r * 8) : (unsigned)a.nwork;
    static const bool want_ts = getenv("FRIEDRICH_AMD_CHAIN_TS") != nullptr;
    if (want_ts && !ctx->chain_ts) {
        void* h = nullptr;
        if (hipHostMalloc(&h, sizeof(unsigned long long) * 128, hipHostMallocMapped) == hipSuccess) {
            memset(h, 0, sizeof(unsigned long long) * 128);
            ctx->chain_ts = (unsigned long long*)h;
        } else {
            (void)hipGetLastError();
        }
    }
    a.ts = nullptr;
    if (ctx->chain_ts) {
        void* d = nullptr;
        if (hipHostGetDevicePointer(&d, ctx->chain_ts, 0) == hipSuccess) a.ts = (unsigned long long*)d;
    }
    const double J = (double)a.J;
    ProfScope ps(ctx, FR_PROF_POTF2, (double)kb * kb * kb / 3.0 + (double)(rows - kb) * kb * kb, 8.0 * (double)rows * kb * 2.0 + J * 128.0 * 128.0 * 8.0);
    hipLaunchKernelGGL(panel_chain_kernel, dim3(grid), dim3(PT), flat::LDS_BYTES, ctx->ls, a);
    FR_HIP(ctx, hipGetLastError());
    ctx->persistent_pending = true;
    if (ctx->test_force_timeout) {  // test hook: behave as if a hand-off of this launch had timed out
        FR_HIP(ctx, hipStreamSynchronize(ctx->ls));
        ((volatile unsigned*)ctx->host_status)[0] = 1u;
    }
    return FR_OK;
}

int launch_potf2(fr_ctx* ctx, double* A, int64_t lda, int64_t nbk, int64_t col0, int mode, double sub, double* inv,
                 int64_t ldinv, int64_t* info, double* cest)
{
    if (nbk <= 0) return FR_OK;
    if (nbk > PB) return set_err(ctx, FR_INVALID_ARGUMENT, "potf2 block too large");
    if (!ctx->potf2_lds_set) {  // per context (= per device): the attribute belongs to the device's copy of the kernel
        FR_TRY(set_dyn_lds(ctx, reinterpret_cast<const void*>(potf2_kernel), (int)POTF2_LDS));
        FR_TRY(set_dyn_lds(ctx, reinterpret_cast<const void*>(potf2_uncapped_kernel), (int)POTF2_LDS));
        FR_TRY(set_dyn_lds(ctx, reinterpret_cast<const void*>(potf2_flat_kernel), (int)flat::LDS_BYTES));
        ctx->potf2_lds_set = true;
    }
    static const int force = getenv("FRIEDRICH_AMD_K4_UNCAPPED") ? atoi(getenv("FRIEDRICH_AMD_K4_UNCAPPED")) : -1;
    static const int force_flat = getenv("FRIEDRICH_AMD_K4_FLAT") ? atoi(getenv("FRIEDRICH_AMD_K4_FLAT")) : -1;
    const bool uncapped = force >= 0 ? force == 1 : (ctx->reserve_now > 0 || ctx->world > 1 || ctx->k4_alone);
    // the flat kernel: full blocks with their inverse, wherever the diagonal-block kernel has its CU to itself
    const bool flat_ok = nbk == PB && mode != 3 && inv != nullptr;
    const bool use_flat = flat_ok && (force_flat >= 0 ? force_flat == 1 : (ctx->k4_flat >= 0 ? ctx->k4_flat == 1 : uncapped));
    ProfScope ps(ctx, FR_PROF_POTF2, (double)nbk * nbk * nbk * (2.0 / 3.0), (double)nbk * nbk * 8.0 * 3.0);
    if (use_flat)
        hipLaunchKernelGGL(potf2_flat_kernel, dim3(1), dim3(PT), flat::LDS_BYTES, ctx->ls, A, lda, (int)nbk, col0, mode, sub, inv, ldinv, info,
                           cest, ctx->xcd_reserve != 0 ? ctx->xcc_word : nullptr);
    else
        hipLaunchKernelGGL(uncapped ? potf2_uncapped_kernel : potf2_kernel, dim3(1), dim3(PT), POTF2_LDS, ctx->ls, A, lda, (int)nbk, col0, mode,
                           sub, inv, ldinv, info, cest, ctx->xcd_reserve != 0 ? ctx->xcc_word : nullptr);
    FR_HIP(ctx, hipGetLastError());
    return FR_OK;
}

}  // namespace fr
