// prior.hip -- LinearPrior::fit (src/parameters/prior.rs:139-159) on the device: SURVEY.md section 8, row f5.
//
// The reference solves  min | [1 | X] w - y |  with nalgebra's SVD (`svd(true, true).solve(y, 0.)`) on the host: an
// n x (d + 1) tall-skinny problem that would otherwise force the training inputs back to the host at large n.  Here the
// inputs stay in HBM: a tall-skinny Householder QR (TSQR) reduces [1 | X | y] to its (d + 1) x (d + 2) triangular factor
// [R | z] -- 256-row blocks, one workgroup each, the triangles stacked and reduced again until one is left -- and the host
// finishes with the SVD of the tiny R (one-sided Jacobi):  w = V diag(1 / sigma_i, sigma_i > 0) U^T z.  R has exactly the
// singular values and right singular vectors of [1 | X], and U^T z = (Q U)^T y, so this IS the reference's SVD solve up to
// rounding (both are backward stable; the normal equations are never formed).
#include "fr_internal.hpp"

namespace fr {

constexpr int QRB = 256;        // rows per block
constexpr int QLD = QRB + 1;    // LDS leading dimension (odd: the column-owning threads hit distinct banks)
constexpr int QP_MAX = 64;      // columns of a block, right-hand side included (d <= 62)

// One block: rows [blockIdx.x * QRB, ...) of the `rows` x P input (column-major, ld lda; level 1: column 0 = ones,
// columns 1 .. d = X, column P - 1 = y, none of it materialised) -> the leading p x P triangle, written as rows
// [blockIdx.x * p, ...) of `out` (ld ldo).
__global__ __launch_bounds__(QRB) void tsqr_kernel(const double* __restrict__ A, int64_t rows, int64_t lda, int p, int P,
                                                   const double* __restrict__ X, int64_t ldx, const double* __restrict__ y,
                                                   double* __restrict__ out, int64_t ldo)
{
    extern __shared__ __attribute__((aligned(16))) double M[];  // QLD x P
    __shared__ double partial[4][QP_MAX];
    __shared__ double coef[QP_MAX];
    __shared__ double head[2];  // alpha, 2 / v^T v
    const int t = threadIdx.x;
    const int64_t r = (int64_t)blockIdx.x * QRB + t;
    for (int c = 0; c < P; ++c) {
        double v = 0.0;
        if (r < rows) {
            if (X)
                v = (c == 0) ? 1.0 : ((c == P - 1) ? y[r] : X[r + (int64_t)(c - 1) * ldx]);
            else
                v = A[r + (int64_t)c * lda];
        }
        M[t + c * QLD] = v;
    }
    __syncthreads();
    const int cc = t & 63, chunk = t >> 6;
    for (int j = 0; j < p; ++j) {
        // dots[c] = sum_{i >= j} M[i][j] M[i][c], c = j .. P - 1: thread (c, chunk) sums 64 rows
        if (cc >= j && cc < P) {
            double s = 0.0;
            const int i0 = chunk * 64 > j ? chunk * 64 : j;
            for (int i = i0; i < (chunk + 1) * 64; ++i) s = __builtin_fma(M[i + j * QLD], M[i + cc * QLD], s);
            partial[chunk][cc] = s;
        }
        __syncthreads();
        if (t >= j && t < P) {
            const double dot = (partial[0][t] + partial[1][t]) + (partial[2][t] + partial[3][t]);
            const double nrm2 = (partial[0][j] + partial[1][j]) + (partial[2][j] + partial[3][j]);
            const double ajj = M[j + j * QLD];
            const double alpha = (nrm2 > 0.0) ? ((ajj >= 0.0) ? -sqrt(nrm2) : sqrt(nrm2)) : 0.0;
            const double vtv = 2.0 * (nrm2 - alpha * ajj);  // |a_j - alpha e_j|^2
            const double scale = vtv > 0.0 ? 2.0 / vtv : 0.0;
            // v^T M[:, c] = dots[c] - alpha M[j][c]
            coef[t] = (dot - alpha * M[j + t * QLD]) * scale;
            if (t == j) {
                head[0] = alpha;
                head[1] = scale;
            }
        }
        __syncthreads();
        const double alpha = head[0];
        if (head[1] != 0.0 && t >= j) {
            const double vt = (t == j) ? M[j + j * QLD] - alpha : M[t + j * QLD];
            for (int c = j + 1; c < P; ++c) M[t + c * QLD] -= coef[c] * vt;
            M[t + j * QLD] = (t == j) ? alpha : 0.0;
        }
        __syncthreads();
    }
    if (t < p)
        for (int c = 0; c < P; ++c) out[(int64_t)blockIdx.x * p + t + (int64_t)c * ldo] = (c >= t) ? M[t + c * QLD] : 0.0;
}

// w = V diag(1 / sigma) U^T z for the p x p upper-triangular R: one-sided Jacobi on the columns of R (R V = U Sigma)
static void svd_solve_small(int p, std::vector<double>& R /* p x p column-major */, const std::vector<double>& z, std::vector<double>& w)
{
    std::vector<double> V((size_t)p * p, 0.0);
    for (int i = 0; i < p; ++i) V[(size_t)i + (size_t)i * p] = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int a = 0; a < p; ++a)
            for (int b = a + 1; b < p; ++b) {
                double aa = 0.0, bb = 0.0, ab = 0.0;
                for (int i = 0; i < p; ++i) {
                    const double x = R[(size_t)i + (size_t)a * p], y = R[(size_t)i + (size_t)b * p];
                    aa += x * x;
                    bb += y * y;
                    ab += x * y;
                }
                if (ab == 0.0) continue;
                off = std::fmax(off, std::fabs(ab) / std::sqrt(aa * bb + 1e-300));
                const double zeta = (bb - aa) / (2.0 * ab);
                const double tt = (zeta >= 0.0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / std::sqrt(1.0 + tt * tt), sn = cs * tt;
                for (int i = 0; i < p; ++i) {
                    double x = R[(size_t)i + (size_t)a * p], y = R[(size_t)i + (size_t)b * p];
                    R[(size_t)i + (size_t)a * p] = cs * x - sn * y;
                    R[(size_t)i + (size_t)b * p] = sn * x + cs * y;
                    x = V[(size_t)i + (size_t)a * p];
                    y = V[(size_t)i + (size_t)b * p];
                    V[(size_t)i + (size_t)a * p] = cs * x - sn * y;
                    V[(size_t)i + (size_t)b * p] = sn * x + cs * y;
                }
            }
        if (off < 1e-15) break;
    }
    // column a of R is now sigma_a u_a:  (U^T z)_a / sigma_a = (col_a . z) / sigma_a^2
    w.assign((size_t)p, 0.0);
    for (int a = 0; a < p; ++a) {
        double s2 = 0.0, uz = 0.0;
        for (int i = 0; i < p; ++i) {
            const double x = R[(size_t)i + (size_t)a * p];
            s2 += x * x;
            uz += x * z[(size_t)i];
        }
        if (!(s2 > 0.0)) continue;  // sigma == 0 (eps = 0 in the reference's solve): the direction is dropped
        const double g = uz / s2;
        for (int i = 0; i < p; ++i) w[(size_t)i] += V[(size_t)i + (size_t)a * p] * g;
    }
}

}  // namespace fr

using namespace fr;

extern "C" int fr_linear_prior_fit(fr_ctx* ctx, const double* X, int64_t n, int64_t ldx, int64_t d, const double* y,
                                   double* out_weights, double* out_intercept)
{
    if (!ctx || !out_intercept || (d > 0 && !out_weights)) return FR_INVALID_ARGUMENT;
    FR_LOCK(ctx);
    FR_HIP(ctx, hipSetDevice(ctx->device));
    if (n < 0 || d < 0 || ldx < (n > 0 ? n : 1)) return set_err(ctx, FR_SHAPE, "bad training input shape");
    const int p = (int)d + 1, P = p + 1;
    if (P > QP_MAX) return set_err(ctx, FR_UNSUPPORTED_KERNEL, "LinearPrior::fit on the device handles up to %d features", QP_MAX - 2);
    if (n > 0 && ((d > 0 && !X) || !y)) return set_err(ctx, FR_INVALID_ARGUMENT, "null training data");
    Staged xs(ctx), ys(ctx);
    FR_TRY(xs.in(X, n, d, ldx));
    FR_TRY(ys.in(y, n, 1, n > 0 ? n : 1));
    std::vector<double> Rz((size_t)p * P, 0.0);
    if (n > 0) {
        const size_t lds = sizeof(double) * (size_t)QLD * (size_t)P;
        if (!ctx->prior_lds_set) {  // per context (= per device): the attribute belongs to the device's copy of the kernel
            FR_TRY(set_dyn_lds(ctx, reinterpret_cast<const void*>(tsqr_kernel), (int)(sizeof(double) * QLD * QP_MAX)));
            ctx->prior_lds_set = true;
        }
        WsGuard g0(ctx), g1(ctx);
        int64_t rows = n;
        const double* cur = nullptr;
        int64_t cur_ld = 0;
        bool first = true;
        int flip = 0;
        while (true) {
            const int64_t nblk = (rows + QRB - 1) / QRB;
            const int64_t orow = nblk * p;
            double* out = (flip ? g1 : g0).get(sizeof(double) * (size_t)orow * (size_t)P);
            if (!out) return FR_OUT_OF_MEMORY;
            hipLaunchKernelGGL(tsqr_kernel, dim3((unsigned)nblk), dim3(QRB), lds, ctx->ls, cur, rows, cur_ld, p, P,
                               first ? (d > 0 ? xs.dev : (const double*)ys.dev) : nullptr, xs.ld, first ? ys.dev : nullptr, out, orow);
            FR_HIP(ctx, hipGetLastError());
            cur = out;
            cur_ld = orow;
            rows = orow;
            first = false;
            flip ^= 1;
            if (nblk == 1) break;
        }
        FR_HIP(ctx, hipMemcpyAsync(Rz.data(), cur, sizeof(double) * (size_t)p * (size_t)P, hipMemcpyDeviceToHost, ctx->stream));
        FR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        FR_TRY(check_status_word(ctx));
    }
    std::vector<double> R((size_t)p * p), z((size_t)p), w;
    for (int c = 0; c < p; ++c)
        for (int r = 0; r < p; ++r) R[(size_t)r + (size_t)c * p] = Rz[(size_t)r + (size_t)c * p];
    for (int r = 0; r < p; ++r) z[(size_t)r] = Rz[(size_t)r + (size_t)p * p];
    svd_solve_small(p, R, z, w);
    *out_intercept = w[0];                                       // prior.rs:157
    for (int64_t c = 0; c < d; ++c) out_weights[c] = w[(size_t)c + 1];  // :158
    return FR_OK;
}
