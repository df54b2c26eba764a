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
// One workgroup of 512 threads; the block lives in REGISTERS: thread (i = t & 127, cg = t >> 7) owns the 32
// elements (i, c = cg + 4k), statically indexed.  Right-looking, one column per step, two barriers per step; the only
// LDS traffic is one 128-entry vector V per step:
//   Lc[x] = L[x, j] (scaled column j),   Vc[c] = L[c, j] for c > j,   Vc[128 + c] = X[j, c] (row j of the inverse, c < j)
// so that BOTH rank-1 updates of the step are the same predicate-free expression  a(i, c) -= Lc[i] * Vc[..c..]:
//   c > j : trailing factor update   A[i, c] -= L[i, j] L[c, j]                (c <= i)
//   c < j : forward substitution     X[i, c] -= L[i, j] X[j, c]               (the inverse rides in the dead columns)
//   c = j : new inverse column       X[i, j]  = -L[i, j] / p
// Column j is stored to global memory by its owners as soon as it is scaled (the barriers order LDS only).  Bound by
// (~1k cycles per column: sqrt -> divide -> barrier -> 17 LDS reads + 16 FMAs -> barrier), one launch per block.
#include "fr_internal.hpp"

namespace fr {

constexpr int PB = 128;
constexpr int PT = 512;  // threads: 8 waves x <= 128 VGPRs fit beside ONE resident GEMM workgroup (look-ahead overlap)
constexpr int PE = 32;   // elements per thread
constexpr int PG = 4;    // column groups
constexpr int PBATCH = 8;  // LDS reads in flight per thread in the update phase

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it would wait for the
// column store of every step to be acknowledged by memory (~2 us per step, 6x the rest of the step).
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// sqrt(d) and 1/sqrt(d) from ONE v_rsq_f64 seed and Newton steps (~15 dependent instructions).  The step's critical
// path is sqrt -> reciprocal -> column scaling; libm's sqrt + two IEEE divisions are ~100 dependent f64 instructions,
// most of a step.  Results agree with sqrt()/division to the last bit or 1 ulp (the oracle comparison bounds it).
__device__ __forceinline__ void sqrt_rsqrt(double d, double& p, double& ip)
{
    if (d == 0.0) {  // plain-sqrt mode: sqrt(0) = 0, then the reference divides by zero
        p = 0.0;
        ip = __builtin_inf();
        return;
    }
    double r = __builtin_amdgcn_rsq(d);
    const double h = 0.5 * d;
    r = r * __builtin_fma(-h * r, r, 1.5);
    r = r * __builtin_fma(-h * r, r, 1.5);
    double q = d * r;
    q = __builtin_fma(0.5 * r, __builtin_fma(-q, q, d), q);  // sqrt(d), corrected
    r = r * __builtin_fma(-q, r, 2.0);                        // 1 / q
    p = q;
    ip = r;
}

// pivot rule for one diagonal value (executed by ONE thread per step): pivot and its reciprocal; logs substitutions
// and failures
__device__ __forceinline__ void pivot_of(double d, int mode, double sub, int64_t col, int64_t* __restrict__ info, double& p,
                                         double& ip)
{
    if (mode == 3) {  // already a factor
        p = d;
        ip = 1.0 / d;
        return;
    }
    if (mode == 2 || d > 0.0) {  // insert_column: plain sqrt (NaN for d < 0)
        sqrt_rsqrt(d, p, ip);
        return;
    }
    if (mode == 1 && sub > 0.0) {
        const int64_t q = info[1];
        info[3 + q] = col;
        info[1] = q + 1;
        sqrt_rsqrt(sub, p, ip);
        return;
    }
    if (info[0] == 0) info[0] = 1 + col;
    p = __builtin_nan("");
    ip = p;
}

__global__ __launch_bounds__(PT, 4) void potf2_kernel(double* __restrict__ A, int64_t lda, int n, int64_t col0, int mode,
                                                   double sub, double* __restrict__ inv, int64_t ldinv,
                                                   int64_t* __restrict__ info)
{
    // Lc[x]      : L(x, j), the scaled column j (x > j)                          -- the row multiplier of the step
    // Vc[c]      : L(c, j) for c > j (0 in mode 3: no factor update), Vc[PB + c] : X(j, c) for c < j (0 for c >= j)
    // so that the update of element (i, c) is  a -= Lc[i] * Vc[c > j ? c : PB + c]  with no per-element predicate.
    // The panel is the critical path of the look-ahead pipeline and this workgroup shares its CU with a trailing-update
    // GEMM workgroup: take the issue slots first (raised wave priority), the GEMM waves fill what is left.
    __builtin_amdgcn_s_setprio(3);
    __shared__ double Lc[PB];
    __shared__ double Vc[2 * PB];
    __shared__ __attribute__((aligned(16))) double piv[2];  // {pivot, 1/pivot} of the current step
    const int t = threadIdx.x;
    const int i = t & (PB - 1);
    const int cg = t >> 7;  // 4 column groups of 128 threads (two waves): cg is wave-uniform
    const bool row_ok = i < n;
    const int wave_row0 = i & 64;  // first row held by this wave

    double a[PE];  // working element (i, cg + 4k): A, then (once column c is done) the inverse X
#pragma unroll
    for (int k = 0; k < PE; ++k) {
        const int c = cg + PG * k;
        a[k] = (row_ok && c < n && i >= c) ? A[i + (int64_t)c * lda] : 0.0;
    }
    if (t == 0) {
        double p0, ip0;
        pivot_of(a[0], mode, sub, col0, info, p0, ip0);
        piv[0] = p0;
        piv[1] = ip0;
    }
    lds_barrier();

    for (int j = 0; j < n; ++j) {
        const int jcg = j & (PG - 1), jk = j / PG;
        // ---- phase 1: owners of column j / row j scale and publish (sqrt and reciprocal were computed by ONE thread
        //      at the end of the previous step)
        const double p = piv[0], ip = piv[1];
        if (cg == jcg && i >= j && row_ok) {  // the threads holding column j
#pragma unroll
            for (int k = 0; k < PE; ++k) {
                if (k == jk) {  // uniform: exactly one of the 16 statically indexed bodies runs
                    const double v = a[k];
                    // col /= denom: quotient by reciprocal + one residual correction (== IEEE division except for rare
                    // last-bit ties); the reciprocal is already on hand, a full division is ~35 dependent instructions
                    double q = v * ip;
                    q = __builtin_fma(__builtin_fma(-q, p, v), ip, q);
                    q = (mode == 3) ? v : q;
                    const bool diag = (i == j);
                    const double lv = diag ? p : q;  // L(i, j)
                    Lc[i] = lv;
                    Vc[i] = (mode == 3) ? 0.0 : lv;
                    if (mode != 3) A[i + (int64_t)j * lda] = lv;  // fire and forget: the barriers wait on LDS only
                    a[k] = diag ? ip : -q * ip;  // X(i, j): 1/p on the diagonal, else 0 - L(i,j) X(j,j)
                }
            }
        }
        if (i == j) {  // the 4 threads holding row j: scale and publish X(j, c), c < j
#pragma unroll
            for (int k = 0; k < PE; ++k) {
                const int c = cg + PG * k;
                const double sc = a[k] * ip;
                const bool lt = c < j;
                a[k] = lt ? sc : a[k];
                Vc[PB + c] = lt ? sc : 0.0;  // zero for c >= j: the update of column j itself must be a no-op
            }
        }
        lds_barrier();
        // ---- phase 2: a(i, c) -= L(i, j) * (c > j ? L(c, j) : X(j, c)).  One LDS read + one FMA per element; waves whose
        //      rows are all finished skip it (the block is VALU-throughput bound: 1024 threads x 16 elements per step)
        if (wave_row0 + 63 > j) {
            const bool act = (i > j) && row_ok;
            const double lraw = Lc[i];
            const double lij = act ? lraw : 0.0;
            if (act && i == j + 1 && cg == ((j + 1) & (PG - 1))) {
                // owner of the next diagonal element: take the next pivot now; the other waves overlap it with their
                // 16 updates
                const int nk = (j + 1) / PG;
                double nd = 0.0;
#pragma unroll
                for (int k = 0; k < PE; ++k)
                    if (k == nk) nd = a[k];
                if (mode != 3) nd = nd - lij * lij;
                double pn, ipn;
                pivot_of(nd, mode, sub, col0 + j + 1, info, pn, ipn);
                piv[0] = pn;
                piv[1] = ipn;
            }
            // two batches of 16: all LDS reads of a batch first, then its FMAs (keeps the kernel under 128 VGPRs so that
            // 8 waves fit next to one resident GEMM workgroup: 512 - 240 = 272 registers per SIMD lane)
#pragma unroll
            for (int k0 = 0; k0 < PE; k0 += PBATCH) {
                double vc[PBATCH];
#pragma unroll
                for (int k = 0; k < PBATCH; ++k) {
                    const int c = cg + PG * (k0 + k);
                    vc[k] = Vc[c + ((c > j) ? 0 : PB)];
                }
#pragma unroll
                for (int k = 0; k < PBATCH; ++k) a[k0 + k] = __builtin_fma(-lij, vc[k], a[k0 + k]);
            }
        }
        lds_barrier();
    }

#pragma unroll
    for (int k = 0; k < PE; ++k) {
        const int c = cg + PG * k;
        if (row_ok && c < n && i >= c) {
            if (inv) inv[i + (int64_t)c * ldinv] = a[k];
        } else if (row_ok && c < n && inv) {
            inv[i + (int64_t)c * ldinv] = 0.0;
        }
    }
}

int launch_potf2(fr_ctx* ctx, double* A, int64_t lda, int64_t nbk, int64_t col0, int mode, double sub, double* inv,
                 int64_t ldinv, int64_t* info)
{
    if (nbk <= 0) return FR_OK;
    if (nbk > PB) return set_err(ctx, FR_INVALID_ARGUMENT, "potf2 block too large");
    ProfScope ps(ctx, FR_PROF_POTF2, (double)nbk * nbk * nbk * (2.0 / 3.0), (double)nbk * nbk * 8.0 * 3.0);
    hipLaunchKernelGGL(potf2_kernel, dim3(1), dim3(PT), 0, ctx->ls, A, lda, (int)nbk, col0, mode, sub, inv, ldinv, info);
    FR_HIP(ctx, hipGetLastError());
    return FR_OK;
}

}  // namespace fr
