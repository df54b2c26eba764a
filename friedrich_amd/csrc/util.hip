// util.hip -- K7: small HBM-bound helpers (fills, copies, per-column reductions) used by the epilogues of
// predict / predict_variance / predict_mean_variance / likelihood (src/gaussian_process/mod.rs:203-219,
// 241, 266-270, 313-319) and by the factor download paths.
#include "fr_internal.hpp"
#include <mutex>
#include <set>
#include <utility>

namespace fr {

__global__ void fill_kernel(double* p, int64_t rows, int64_t cols, int64_t ld, double v)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t c = blockIdx.y;
    if (r < rows && c < cols) p[r + c * ld] = v;
}

__global__ void copy_kernel(const double* src, int64_t lds, double* dst, int64_t ldd, int64_t rows, int64_t cols)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t c = blockIdx.y; c < cols; c += gridDim.y)
        if (r < rows) dst[r + c * ldd] = src[r + c * lds];
}

// 512-block inverses, step 0: block q (512 x 512, contiguous) gets the four 128 x 128 inverses 4q .. 4q+3 on its
// diagonal and zeros everywhere else.  grid = (2, 512, blocks): thread -> row, blockIdx.y -> column
__global__ void blockdiag512_kernel(const double* __restrict__ dinv128, double* __restrict__ w, int64_t nblocks)
{
    const int r = (int)(blockIdx.x * blockDim.x + threadIdx.x);  // 0..511
    const int c = (int)blockIdx.y;
    const int64_t q = blockIdx.z;
    if (q >= nblocks) return;
    double v = 0.0;
    if ((r >> 7) == (c >> 7)) v = dinv128[(4 * q + (r >> 7)) * (128 * 128) + (r & 127) + (int64_t)(c & 127) * 128];
    w[q * (512 * 512) + r + (int64_t)c * 512] = v;
}

__global__ void identity_kernel(double* p, int64_t n, int64_t ld)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t c = blockIdx.y; c < n; c += gridDim.y)
        if (r < n) p[r + c * ld] = (r == c) ? 1.0 : 0.0;
}

__global__ void tri_fill_kernel(double* p, int64_t n, int64_t ld, double v)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t c = blockIdx.y; c < n; c += gridDim.y)
        if (r < n && r < c) p[r + c * ld] = v;
}

// out (cols x rows) = in (rows x cols)^T through a 64x64 LDS tile (coalesced on both sides)
__global__ __launch_bounds__(256) void transpose_kernel(const double* __restrict__ in, int64_t rows, int64_t cols, int64_t ldi,
                                                        double* __restrict__ out, int64_t ldo)
{
    __shared__ double tile[64][65];
    const int64_t r0 = (int64_t)blockIdx.x * 64, c0 = (int64_t)blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int k = ty; k < 64; k += 4) {
        const int64_t r = r0 + tx, c = c0 + k;
        tile[k][tx] = (r < rows && c < cols) ? in[r + c * ldi] : 0.0;
    }
    __syncthreads();
    for (int k = ty; k < 64; k += 4) {
        const int64_t c = c0 + tx, r = r0 + k;  // out(c, r) = in(r, c)
        if (r < rows && c < cols) out[c + r * ldo] = tile[tx][k];
    }
}

// upper := lower^T through a 64x64 LDS tile (coalesced on both sides)
__global__ __launch_bounds__(256) void symmetrize_kernel(double* p, int64_t n, int64_t ld)
{
    __shared__ double tile[64][65];
    const int64_t bi = blockIdx.x, bj = blockIdx.y;
    if (bj > bi) return;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int c = ty; c < 64; c += 4) {
        const int64_t r = bi * 64 + tx, cc = bj * 64 + c;
        tile[c][tx] = (r < n && cc < n) ? p[r + cc * ld] : 0.0;
    }
    __syncthreads();
    for (int c = ty; c < 64; c += 4) {
        // write element (row = bj*64 + tx, col = bi*64 + c) = lower(bi*64 + c, bj*64 + tx)
        const int64_t r = bj * 64 + tx, cc = bi * 64 + c;
        if (r < n && cc < n && r < cc) p[r + cc * ld] = tile[tx][c];
    }
}

__device__ __forceinline__ double block_sum(double v, double* red)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double tot = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) tot += red[i];
    __syncthreads();
    return tot;
}

// out[j] = sum_i U[i,j] * V[i,j]   (U == V: column norm^2, mod.rs:268)
__global__ __launch_bounds__(256) void col_dot_kernel(const double* U, int64_t ldu, const double* V, int64_t ldv,
                                                      int64_t n, double* out)
{
    __shared__ double red[4];
    const int64_t j = blockIdx.x;
    const double* u = U + j * ldu;
    const double* v = V + j * ldv;
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) acc += u[i] * v[i];
    const double tot = block_sum(acc, red);
    if (threadIdx.x == 0) out[j] = tot;
}

// out[j] = alpha * V[:,j] . y + beta * out[j]   (gemm_tr on a vector, mod.rs:241, 306, 388)
__global__ __launch_bounds__(256) void gemv_t_kernel(const double* V, int64_t n, int64_t ldv, const double* y,
                                                     double alpha, double beta, double* out)
{
    __shared__ double red[4];
    const int64_t j = blockIdx.x;
    const double* v = V + j * ldv;
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) acc += v[i] * y[i];
    const double tot = block_sum(acc, red);
    if (threadIdx.x == 0) out[j] = alpha * tot + (beta != 0.0 ? beta * out[j] : 0.0);
}

// y[i] = alpha * sum_j A[i + j lda] x[j] + beta * y[i]   (single right-hand side of the triangular solves: HBM-bound,
// A is read once).  64 rows per workgroup, four column groups per row, partials combined through LDS in a fixed order.
__global__ __launch_bounds__(256) void gemv_n_kernel(const double* __restrict__ A, int64_t rows, int64_t cols, int64_t lda,
                                                     const double* __restrict__ x, double alpha, double beta,
                                                     double* __restrict__ y)
{
    __shared__ double part[4][64];
    const int r = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + r;
    double acc0 = 0.0, acc1 = 0.0;
    if (i < rows) {
        const double* a = A + i;
        int64_t j = g;
        for (; j + 4 < cols; j += 8) {
            acc0 += a[j * lda] * x[j];
            acc1 += a[(j + 4) * lda] * x[j + 4];
        }
        if (j < cols) acc0 += a[j * lda] * x[j];
    }
    part[g][r] = acc0 + acc1;
    __syncthreads();
    if (g == 0 && i < rows) {
        const double tot = (part[0][r] + part[1][r]) + (part[2][r] + part[3][r]);
        y[i] = alpha * tot + (beta != 0.0 ? beta * y[i] : 0.0);
    }
}

// ---- "skinny" products: a handful of right-hand sides (2 .. 32 query rows) --------------------------------------------
// The FP64 GEMM works in 128-wide tiles: with 16 right-hand sides 7/8 of every tile is padding and the solve costs as much
// as with 128.  These two kernels stream A once from HBM against up to SKC columns of the small operand staged in LDS.
constexpr int SKC = 16;  // right-hand-side columns per pass

// Y[i, c] = alpha * sum_j A[i + j lda] X[j + c ldx] + beta * Y[i, c],  c < mc <= SKC.  64 rows per workgroup; A is consumed
// in chunks of 64 columns whose X rows are staged in LDS; the columns of a chunk are dealt to the four waves (16 loads of A
// in flight per thread), partial sums combined through LDS in a fixed order.
__global__ __launch_bounds__(256) void skinny_n_kernel(const double* __restrict__ A, int64_t rows, int64_t cols, int64_t lda,
                                                       const double* __restrict__ X, int64_t ldx, int mc, double alpha,
                                                       double beta, double* __restrict__ Y, int64_t ldy)
{
    __shared__ __attribute__((aligned(16))) double xs[64][SKC];  // [j in chunk][c]
    __shared__ double part[3][SKC][64];
    const int t = threadIdx.x;
    const int r = t & 63;
    const int g = t >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + r;
    const bool row_ok = i < rows;
    const double* a = A + (row_ok ? i : 0);
    double acc[SKC];
#pragma unroll
    for (int c = 0; c < SKC; ++c) acc[c] = 0.0;
    for (int64_t j0 = 0; j0 < cols; j0 += 64) {
        // stage X[j0 .. j0+63, 0 .. mc): thread (jj = t & 63, cq = t >> 6) loads columns cq, cq + 4, ...
#pragma unroll
        for (int k = 0; k < SKC / 4; ++k) {
            const int c = (t >> 6) + 4 * k;
            const int64_t j = j0 + (t & 63);
            xs[t & 63][c] = (c < mc && j < cols) ? X[j + (int64_t)c * ldx] : 0.0;
        }
        __syncthreads();
        double av[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int64_t j = j0 + g + 4 * k;
            av[k] = (row_ok && j < cols) ? a[j * lda] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const double2* xr = reinterpret_cast<const double2*>(&xs[g + 4 * k][0]);
#pragma unroll
            for (int c = 0; c < SKC; c += 2) {
                const double2 xv = xr[c >> 1];
                acc[c] = __builtin_fma(av[k], xv.x, acc[c]);
                acc[c + 1] = __builtin_fma(av[k], xv.y, acc[c + 1]);
            }
        }
        __syncthreads();
    }
    if (g > 0) {
#pragma unroll
        for (int c = 0; c < SKC; ++c) part[g - 1][c][r] = acc[c];
    }
    __syncthreads();
    if (g == 0 && row_ok) {
#pragma unroll
        for (int c = 0; c < SKC; ++c) {
            if (c < mc) {
                const double tot = (acc[c] + part[0][c][r]) + (part[1][c][r] + part[2][c][r]);
                double* y = Y + i + (int64_t)c * ldy;
                *y = alpha * tot + (beta != 0.0 ? beta * *y : 0.0);
            }
        }
    }
}

// OUT[j, c] = alpha * sum_i A[i + j lda] Y[i + c ldy] + beta * OUT[j, c]   (A^T Y), c < mc <= SKC.  A 64 x 64 tile of A goes
// through LDS (read along i, consumed along j) together with the 64 rows of Y it meets; 64 output rows per workgroup, the
// rows i of a tile dealt to the four waves.
__global__ __launch_bounds__(256) void skinny_t_kernel(const double* __restrict__ A, int64_t rows, int64_t cols, int64_t lda,
                                                       const double* __restrict__ Y, int64_t ldy, int mc, double alpha,
                                                       double beta, double* __restrict__ OUT, int64_t ldo)
{
    __shared__ double tile[64][65];                                  // [i][j]
    __shared__ __attribute__((aligned(16))) double ys[64][SKC];      // [i][c]
    __shared__ double part[3][SKC][64];
    const int t = threadIdx.x;
    const int jl = t & 63;
    const int g = t >> 6;
    const int64_t j0 = (int64_t)blockIdx.x * 64;
    double acc[SKC];
#pragma unroll
    for (int c = 0; c < SKC; ++c) acc[c] = 0.0;
    // software pipeline: the global loads of tile k + 1 are in flight while tile k is consumed from LDS
    double ra[16], ry[SKC / 4];
    auto fetch = [&](int64_t i0) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int jj = (t >> 6) + 4 * k;
            const int64_t ii = i0 + (t & 63), jc = j0 + jj;
            ra[k] = (ii < rows && jc < cols) ? A[ii + jc * lda] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < SKC / 4; ++k) {
            const int c = (t >> 6) + 4 * k;
            const int64_t ii = i0 + (t & 63);
            ry[k] = (c < mc && ii < rows) ? Y[ii + (int64_t)c * ldy] : 0.0;
        }
    };
    fetch(0);
    for (int64_t i0 = 0; i0 < rows; i0 += 64) {
        // 64 x 64 tile of A: thread (il = t & 63, jg = t >> 6) holds row il of columns jg, jg + 4, ...;  64 rows of Y alongside
#pragma unroll
        for (int k = 0; k < 16; ++k) tile[t & 63][(t >> 6) + 4 * k] = ra[k];
#pragma unroll
        for (int k = 0; k < SKC / 4; ++k) ys[t & 63][(t >> 6) + 4 * k] = ry[k];
        __syncthreads();
        if (i0 + 64 < rows) fetch(i0 + 64);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int ii = g + 4 * k;
            const double av = tile[ii][jl];
            const double2* yr = reinterpret_cast<const double2*>(&ys[ii][0]);
#pragma unroll
            for (int c = 0; c < SKC; c += 2) {
                const double2 yv = yr[c >> 1];
                acc[c] = __builtin_fma(av, yv.x, acc[c]);
                acc[c + 1] = __builtin_fma(av, yv.y, acc[c + 1]);
            }
        }
        __syncthreads();
    }
    if (g > 0) {
#pragma unroll
        for (int c = 0; c < SKC; ++c) part[g - 1][c][jl] = acc[c];
    }
    __syncthreads();
    const int64_t j = j0 + jl;
    if (g == 0 && j < cols) {
#pragma unroll
        for (int c = 0; c < SKC; ++c) {
            if (c < mc) {
                const double tot = (acc[c] + part[0][c][jl]) + (part[1][c][jl] + part[2][c][jl]);
                double* o = OUT + j + (int64_t)c * ldo;
                *o = alpha * tot + (beta != 0.0 ? beta * *o : 0.0);
            }
        }
    }
}

__global__ void axpby_kernel(int64_t n, double a, const double* x, double b, double* y)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = a * x[i] + (b != 0.0 ? b * y[i] : 0.0);
}

__global__ void diag_zero_kernel(const double* A, int64_t n, int64_t lda, int64_t* flag)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && A[i + i * lda] == 0.0) flag[0] = 1;
}

// What fr_chol_add_rows reads back after an append, in ONE launch that writes pinned host memory directly: out[0] (as int64) = is
// any diagonal entry of the n new rows zero, out[1 ..] = the conditioning estimates of the nb diagonal blocks that changed (round
// 4: a memset, this check, two device-to-host copies -- 25 us of launches on the tail of every append)
__global__ __launch_bounds__(256) void append_status_kernel(const double* A, int64_t n, int64_t lda, const double* cest, int64_t nb, double* out)
{
    __shared__ int any;
    if (threadIdx.x == 0) any = 0;
    __syncthreads();
    int mine = 0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x)
        if (A[i + i * lda] == 0.0) mine = 1;
    if (mine) any = 1;  // (benign race: every writer writes 1)
    __syncthreads();
    if (threadIdx.x == 0) reinterpret_cast<int64_t*>(out)[0] = any;
    for (int64_t b = threadIdx.x; b < nb; b += blockDim.x) out[1 + b] = cest[b];
}

__global__ __launch_bounds__(256) void sum_log_abs_kernel(const double* v, int64_t n, double* out)
{
    __shared__ double red[4];
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) acc += log(fabs(v[i]));
    const double tot = block_sum(acc, red);
    if (threadIdx.x == 0) out[0] = tot;
}

static inline unsigned ydim(int64_t cols) { return (unsigned)(cols < 65535 ? (cols > 0 ? cols : 1) : 65535); }

int launch_fill(fr_ctx* ctx, double* p, int64_t rows, int64_t cols, int64_t ld, double v)
{
    if (rows <= 0 || cols <= 0) return FR_OK;
    for (int64_t c0 = 0; c0 < cols; c0 += 65535) {
        const int64_t cc = (cols - c0) < 65535 ? (cols - c0) : 65535;
        hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((rows + 255) / 256), (unsigned)cc), dim3(256), 0, ctx->ls,
                           p + c0 * ld, rows, cc, ld, v);
    }
    FR_HIP(ctx, hipGetLastError());
    return FR_OK;
}

int launch_copy(fr_ctx* ctx, const double* src, int64_t lds, double* dst, int64_t ldd, int64_t rows, int64_t cols)
{
    if (rows <= 0 || cols <= 0) return FR_OK;
    hipLaunchKernelGGL(copy_kernel, dim3((unsigned)((rows + 255) / 256), ydim(cols)), dim3(256), 0, ctx->ls, src,
                       lds, dst, ldd, rows, cols);
    FR_HIP(ctx, hipGetLastError());
    return FR_OK;
}

int launch_blockdiag512(fr_ctx* ctx, const double* dinv128, double* w, int64_t nblocks)
{
    if (nblocks <= 0) return FR_OK;
    hipLaunchKernelGGL(blockdiag512_kernel, dim3(2, 512, (unsigned)nblocks), dim3(256), 0, ctx->ls, dinv128, w, nblocks);
    FR_HIP(ctx, hipGetLastError());
    return FR_OK;
}

int launch_set_identity(fr_ctx* ctx, double* p, int64_t n, int64_t ld)
{
    if (n <= 0) return FR_OK;
    hipLaunchKernelGGL(identity_kernel, dim3((unsigned)((n + 255) / 256), ydim(n)), dim3(256), 0, ctx->ls, p, n, ld);
    FR_HIP(ctx, hipGetLastError());
    return FR_OK;
}

int launch_tri_fill(fr_ctx* ctx, double* p, int64_t n, int64_t ld, double v)
{
    if (n <= 0) return FR_OK;
    hipLaunchKernelGGL(tri_fill_kernel, dim3((unsigned)((n + 255) / 256), ydim(n)), dim3(256), 0, ctx->ls, p, n, ld,
                       v);
    FR_HIP(ctx, hipGetLastError());
    return FR_OK;
}

int launch_symmetrize(fr_ctx* ctx, double* p, int64_t n, int64_t ld)
{
    if (n <= 0) return FR_OK;
    const int64_t nbk = (n + 63) / 64;
    if (nbk > 65535) return set_err(ctx, FR_INVALID_ARGUMENT, "matrix too large to symmetrize");
    hipLaunchKernelGGL(symmetrize_kernel, dim3((unsigned)nbk, (unsigned)nbk), dim3(256), 0, ctx->ls, p, n, ld);
    FR_HIP(ctx, hipGetLastError());
    return FR_OK;
}

int launch_col_dot(fr_ctx* ctx, const double* U, int64_t ldu, const double* V, int64_t ldv, int64_t n, int64_t m,
                   double* out)
{
    if (m <= 0) return FR_OK;
    ProfScope ps(ctx, FR_PROF_REDUCE, 2.0 * (double)n * m, 8.0 * (double)n * m * (U == V ? 1.0 : 2.0));
    hipLaunchKernelGGL(col_dot_kernel, dim3((unsigned)m), dim3(256), 0, ctx->ls, U, ldu, V, ldv, n, out);
    FR_HIP(ctx, hipGetLastError());
    return FR_OK;
}

int launch_col_norm2(fr_ctx* ctx, const double* V, int64_t n, int64_t m, int64_t ldv, double* out)
{
    return launch_col_dot(ctx, V, ldv, V, ldv, n, m, out);
}

int launch_gemv_t(fr_ctx* ctx, const double* V, int64_t n, int64_t m, int64_t ldv, const double* y, double alpha,
                  double beta, double* out)
{
    if (m <= 0) return FR_OK;
    ProfScope ps(ctx, FR_PROF_REDUCE, 2.0 * (double)n * m, 8.0 * (double)n * m);
    hipLaunchKernelGGL(gemv_t_kernel, dim3((unsigned)m), dim3(256), 0, ctx->ls, V, n, ldv, y, alpha, beta, out);
    FR_HIP(ctx, hipGetLastError());
    return FR_OK;
}

int launch_transpose(fr_ctx* ctx, const double* in, int64_t rows, int64_t cols, int64_t ldi, double* out, int64_t ldo)
{
    if (rows <= 0 || cols <= 0) return FR_OK;
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)((rows + 63) / 64), (unsigned)((cols + 63) / 64)), dim3(256), 0,
                       ctx->ls, in, rows, cols, ldi, out, ldo);
    FR_HIP(ctx, hipGetLastError());
    return FR_OK;
}

// Y (rows x m) = alpha A (rows x cols) X (cols x m) + beta Y, m small: passes of SKC columns
int launch_skinny_n(fr_ctx* ctx, const double* A, int64_t rows, int64_t cols, int64_t lda, const double* X, int64_t ldx,
                    int64_t m, double alpha, double beta, double* Y, int64_t ldy)
{
    if (rows <= 0 || m <= 0) return FR_OK;
    ProfScope ps(ctx, FR_PROF_REDUCE, 2.0 * (double)rows * cols * m, 8.0 * (double)rows * cols * ((m + SKC - 1) / SKC));
    for (int64_t c0 = 0; c0 < m; c0 += SKC) {
        const int mc = (int)((m - c0) < SKC ? (m - c0) : SKC);
        hipLaunchKernelGGL(skinny_n_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(256), 0, ctx->ls, A, rows, cols, lda,
                           X + c0 * ldx, ldx, mc, alpha, beta, Y + c0 * ldy, ldy);
    }
    FR_HIP(ctx, hipGetLastError());
    return FR_OK;
}

// OUT (cols x m) = alpha A^T (cols x rows) Y (rows x m) + beta OUT
int launch_skinny_t(fr_ctx* ctx, const double* A, int64_t rows, int64_t cols, int64_t lda, const double* Y, int64_t ldy,
                    int64_t m, double alpha, double beta, double* OUT, int64_t ldo)
{
    if (cols <= 0 || m <= 0) return FR_OK;
    ProfScope ps(ctx, FR_PROF_REDUCE, 2.0 * (double)rows * cols * m, 8.0 * (double)rows * cols * ((m + SKC - 1) / SKC));
    for (int64_t c0 = 0; c0 < m; c0 += SKC) {
        const int mc = (int)((m - c0) < SKC ? (m - c0) : SKC);
        hipLaunchKernelGGL(skinny_t_kernel, dim3((unsigned)((cols + 63) / 64)), dim3(256), 0, ctx->ls, A, rows, cols, lda,
                           Y + c0 * ldy, ldy, mc, alpha, beta, OUT + c0 * ldo, ldo);
    }
    FR_HIP(ctx, hipGetLastError());
    return FR_OK;
}

int launch_gemv_n(fr_ctx* ctx, const double* A, int64_t rows, int64_t cols, int64_t lda, const double* x, double alpha,
                  double beta, double* y)
{
    if (rows <= 0) return FR_OK;
    ProfScope ps(ctx, FR_PROF_REDUCE, 2.0 * (double)rows * cols, 8.0 * (double)rows * cols);
    hipLaunchKernelGGL(gemv_n_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(256), 0, ctx->ls, A, rows, cols, lda, x,
                       alpha, beta, y);
    FR_HIP(ctx, hipGetLastError());
    return FR_OK;
}

int launch_axpby_vec(fr_ctx* ctx, int64_t n, double a, const double* x, double b, double* y)
{
    if (n <= 0) return FR_OK;
    hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->ls, n, a, x, b, y);
    FR_HIP(ctx, hipGetLastError());
    return FR_OK;
}

int launch_diag_check_zero(fr_ctx* ctx, const double* A, int64_t n, int64_t lda, int64_t* flag)
{
    if (n <= 0) return FR_OK;
    hipLaunchKernelGGL(diag_zero_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->ls, A, n, lda, flag);
    FR_HIP(ctx, hipGetLastError());
    return FR_OK;
}

int launch_append_status(fr_ctx* ctx, const double* A, int64_t n, int64_t lda, const double* cest, int64_t nb, double* host_out)
{
    hipLaunchKernelGGL(append_status_kernel, dim3(1), dim3(256), 0, ctx->ls, A, n, lda, cest, nb, host_out);
    FR_HIP(ctx, hipGetLastError());
    return FR_OK;
}

int launch_sum_log_abs(fr_ctx* ctx, const double* v, int64_t n, double* out)
{
    hipLaunchKernelGGL(sum_log_abs_kernel, dim3(1), dim3(256), 0, ctx->ls, v, n, out);
    FR_HIP(ctx, hipGetLastError());
    return FR_OK;
}

int set_dyn_lds(fr_ctx* ctx, const void* fn, int bytes)
{
    static std::mutex m;
    static std::set<std::pair<int, const void*>> done;
    std::lock_guard<std::mutex> lk(m);
    if (done.count({ctx->device, fn})) return FR_OK;
    FR_HIP(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.insert({ctx->device, fn});
    return FR_OK;
}

}  // namespace fr
