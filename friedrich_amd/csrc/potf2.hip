// potf2.hip -- K4: Cholesky of one diagonal block (<= 64 x 64) inside LDS, with friedrich's pivot rules,
// plus the explicit inverse of the factored block (consumed by the GEMM-recast triangular solves).
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
// Column scaling is a true division (`col /= denom`).
//
// Latency-bound by construction (64 dependent column steps); one workgroup, two barriers per column.
#include "fr_internal.hpp"

namespace fr {

constexpr int PB = 64;
constexpr int PS = 65;  // LDS column stride (conflict-free for both column and row walks)

__global__ __launch_bounds__(256) void potf2_kernel(double* __restrict__ A, int64_t lda, int n, int64_t col0, int mode,
                                                    double sub, double* __restrict__ inv, int64_t ldinv,
                                                    int64_t* __restrict__ info)
{
    __shared__ double L[PB * PS];  // element (r, c) at L[c * PS + r]
    __shared__ double X[PB * PS];  // inverse, same layout
    const int t = threadIdx.x;
    const int r = t & 63;
    const int cg = t >> 6;

    for (int c = cg; c < PB; c += 4) {
        double v = 0.0;
        if (r < n && c < n && r >= c) v = A[r + (int64_t)c * lda];
        L[c * PS + r] = v;
        X[c * PS + r] = 0.0;
    }
    __syncthreads();

    for (int j = 0; j < n && mode != 3; ++j) {
        const double d = L[j * PS + j];
        double p;
        bool ok = d > 0.0;
        if (mode == 2) {
            p = sqrt(d);
        } else if (ok) {
            p = sqrt(d);
        } else if (mode == 1 && sub > 0.0) {
            p = sqrt(sub);
            if (t == 0) {
                const int64_t k = info[1];
                info[3 + k] = col0 + j;
                info[1] = k + 1;
            }
        } else {
            p = __builtin_nan("");
            if (t == 0 && info[0] == 0) info[0] = 1 + col0 + j;
        }
        if (cg == 0 && r > j && r < n) L[j * PS + r] = L[j * PS + r] / p;
        __syncthreads();
        const double ljr = (r > j) ? L[j * PS + r] : 0.0;
        for (int c = j + 1 + cg; c < n; c += 4)
            if (r >= c && r < n) L[c * PS + r] -= ljr * L[j * PS + c];
        if (t == j) L[j * PS + j] = p;  // nobody reads the pivot slot during the update
        __syncthreads();
    }

    // write the factor back (lower triangle only; the strict upper triangle of A is never read)
    if (mode != 3)
        for (int c = cg; c < n; c += 4)
            if (r < n && r >= c) A[r + (int64_t)c * lda] = L[c * PS + r];

    // inverse by forward substitution, 4 lanes per column of X
    const int c = t >> 2, part = t & 3;
    for (int i = 0; i < n; ++i) {
        double acc = 0.0;
        if (c < n && i > c)
            for (int k = c + part; k < i; k += 4) acc += L[k * PS + i] * X[c * PS + k];
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        if (c < n && i >= c && part == 0) X[c * PS + i] = ((i == c ? 1.0 : 0.0) - acc) / L[i * PS + i];
        __syncthreads();
    }
    if (inv)
        for (int cc = cg; cc < n; cc += 4)
            if (r < n) inv[r + (int64_t)cc * ldinv] = (r >= cc) ? X[cc * PS + r] : 0.0;
}

int launch_potf2(fr_ctx* ctx, double* A, int64_t lda, int64_t nbk, int64_t col0, int mode, double sub, double* inv,
                 int64_t ldinv, int64_t* info)
{
    if (nbk <= 0) return FR_OK;
    if (nbk > PB) return set_err(ctx, FR_INVALID_ARGUMENT, "potf2 block too large");
    ProfScope ps(ctx, FR_PROF_POTF2, (double)nbk * nbk * nbk * (2.0 / 3.0), (double)nbk * nbk * 8.0 * 3.0);
    hipLaunchKernelGGL(potf2_kernel, dim3(1), dim3(256), 0, ctx->ls, A, lda, (int)nbk, col0, mode, sub, inv, ldinv,
                       info);
    FR_HIP(ctx, hipGetLastError());
    return FR_OK;
}

}  // namespace fr
