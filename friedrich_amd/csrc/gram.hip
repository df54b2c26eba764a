// gram.hip -- K1 Gram (covariance) assembly and K3 mean-pairwise-distance reduction.
//
// Replaces the per-pair loops of src/algebra/mod.rs:41-54 (make_covariance_matrix), :70-79 (lower triangle
// + noise^2 inside make_cholesky_cov_matrix), :115-121 (the new columns of add_rows_cholesky_cov_matrix) and
// src/parameters/kernel.rs:94-113 (fit_bandwidth_mean).
//
// Roofline: HBM-bound (8 B written per pair; the n x d inputs are read once per tile row/column through
// LDS).  One workgroup (4 wave64) produces a 128 x 64 tile: training rows are the fast (contiguous) axis of
// both the column-major inputs and the column-major output, so every global access is a 512 B wave
// transaction.  Feature columns are staged through LDS in chunks of 16; a lane keeps 2 rows x 16 columns of
// running ||x-y||^2 / x.y in registers, the 16 column values are LDS broadcasts.
#include "fr_internal.hpp"
#include "kprog_device.hpp"

namespace fr {

constexpr int GT_M = 128;  // tile rows   (rows of A / output rows)
constexpr int GT_N = 64;   // tile cols   (rows of B / output cols)
constexpr int GT_DC = 16;  // feature chunk staged in LDS

struct GramArgs {
    fr_kprog prog;
    const double* A;
    int64_t n1, lda;
    const double* B;
    int64_t n2, ldb;
    int64_t d;
    double* out;
    int64_t ldo;
    int sym;  // 1: A == B, only tiles touching the lower triangle (by 128-blocks), + noise2 on the diagonal
    double noise2;
    int64_t tiles_n;  // cross mode: tiles along n2
    int own_world, own_rank;  // sym mode, multi-GPU: only block columns (width own_nb) of own_rank are assembled
    int64_t own_nb;
    // cross mode, fused with a product: instead of storing the tile, part[ti * n2 + j] = sum over the tile's rows i of
    // k(a_i, b_j) dot_vec[i]  (predict with the cached alpha: the n x m cross-covariance never exists)
    const double* dot_vec;
    double* dot_part;
};

__device__ __forceinline__ void sym_tile(int64_t t, int64_t& bi, int64_t& tj)
{
    // row block bi owns column tiles [0, 2(bi+1)); prefix count = bi(bi+1)
    int64_t b = (int64_t)((sqrt(4.0 * (double)t + 1.0) - 1.0) * 0.5);
    while (b * (b + 1) > t) --b;
    while ((b + 1) * (b + 2) <= t) ++b;
    bi = b;
    tj = t - b * (b + 1);
}

// (three waves per SIMD for the one-accumulator modes: measured 2.93 -> 2.66 ms at N = 32768 -- the kernel waits on LDS
// broadcasts and dependent f64 chains, more waves hide more of it; four waves spill: 4.2 ms)
// LEAF >= 0: the program is that single built-in kernel -- the epilogue is then straight-line code (no program dispatch, no
// rotation of the accumulators, nothing in scratch); LEAF = -1: any program, evaluated by the stack machine.
template <int MODE, int LEAF>
__global__ __launch_bounds__(256, (MODE == (NEED_S | NEED_U)) ? 2 : 3) void gram_kernel(const GramArgs a)
{
#pragma clang fp contract(off)
    __shared__ double XA[GT_DC][GT_M];
    __shared__ double XB[GT_DC][GT_N];

    int64_t ti, tj;
    if (a.sym) {
        sym_tile((int64_t)blockIdx.x, ti, tj);
    } else {
        ti = (int64_t)blockIdx.x / a.tiles_n;
        tj = (int64_t)blockIdx.x % a.tiles_n;
    }
    const int64_t i0 = ti * GT_M, j0 = tj * GT_N;
    if (a.own_world > 1 && (int)((j0 / a.own_nb) % a.own_world) != a.own_rank) return;
    const int t = threadIdx.x;
    const int r = t & 63;
    const int g = t >> 6;

    double s[2][16], u[2][16];
#pragma unroll
    for (int b = 0; b < 16; ++b) {
        s[0][b] = s[1][b] = 0.0;
        u[0][b] = u[1][b] = 0.0;
    }

    for (int64_t c0 = 0; c0 < a.d; c0 += GT_DC) {
        const int dc = (int)((a.d - c0) < GT_DC ? (a.d - c0) : GT_DC);
        // stage the feature chunk: consecutive lanes read consecutive rows of one feature column
#pragma unroll
        for (int p = 0; p < GT_DC / 2; ++p) {
            const int row = t & 127, c = (t >> 7) + 2 * p;
            const int64_t gi = i0 + row;
            XA[c][row] = (c < dc && gi < a.n1) ? a.A[gi + (c0 + c) * a.lda] : 0.0;
        }
#pragma unroll
        for (int p = 0; p < GT_DC / 4; ++p) {
            const int row = t & 63, c = (t >> 6) + 4 * p;
            const int64_t gj = j0 + row;
            XB[c][row] = (c < dc && gj < a.n2) ? a.B[gj + (c0 + c) * a.ldb] : 0.0;
        }
        __syncthreads();
        for (int c = 0; c < dc; ++c) {
            const double xa0 = XA[c][r], xa1 = XA[c][r + 64];
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                const double xb = XB[c][g * 16 + b];
                if (MODE & NEED_S) {
                    // (x1 - x2).norm_squared(): difference, square, sequential sum (kernel.rs:558)
                    const double d0 = xa0 - xb, d1 = xa1 - xb;
                    s[0][b] = s[0][b] + d0 * d0;
                    s[1][b] = s[1][b] + d1 * d1;
                }
                if (MODE & NEED_U) {
                    // x1.dot(x2): sequential over the feature columns (kernel.rs:381)
                    u[0][b] = u[0][b] + xa0 * xb;
                    u[1][b] = u[1][b] + xa1 * xb;
                }
            }
        }
        __syncthreads();
    }

    double av[2] = {0.0, 0.0};
    if (a.dot_vec) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t gi = i0 + r + 64 * h;
            av[h] = gi < a.n1 ? a.dot_vec[gi] : 0.0;
        }
    }
    // one output column (two pairs) of the tile: the values are computed unconditionally and only STORED under the bounds
    // test (padding rows / columns hold zeros: finite inputs, results discarded)
    auto emit = [&](int b, double k0, double k1) {
        const int64_t gj = j0 + g * 16 + b;
        double dotv = 0.0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t gi = i0 + r + 64 * h;
            if (gi < a.n1 && gj < a.n2) {
                double kv = h ? k1 : k0;
                if (a.sym && gi == gj) kv = kv + a.noise2;  // algebra/mod.rs:78
                if (a.dot_vec)
                    dotv = dotv + kv * av[h];
                else
                    a.out[gi + gj * a.ldo] = kv;
            }
        }
        if (a.dot_vec) {  // (uniform) the wave's 128 rows of column gj: fixed-order tree over the 64 lanes
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) dotv = dotv + __shfl_xor(dotv, off, 64);
            if (r == 0 && gj < a.n2) a.dot_part[ti * a.n2 + gj] = dotv;
        }
    };
    if constexpr (LEAF >= 0) {
        // single built-in kernel: 32 independent straight-line evaluations, static register indices
        const bool interior = !a.dot_vec && i0 + GT_M <= a.n1 && j0 + GT_N <= a.n2 && !(a.sym && j0 + GT_N > i0 && j0 < i0 + GT_M);
        if (interior) {
            // whole tile inside the matrix and off the diagonal (all but O(n) of the n^2 / 8192 tiles): no bounds tests, no
            // diagonal test; a column's address is a wave-uniform base (scalar registers) + the lane's row -- measured with
            // SQ_INSTS_VALU: the tests and 64-bit address arithmetic were a third of the epilogue's instructions
            const int gw = __builtin_amdgcn_readfirstlane(g);
            double* col = a.out + i0 + (j0 + gw * 16) * a.ldo;
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                col[(unsigned)r] = kprog_eval_k<LEAF>(a.prog, s[0][b], u[0][b]);
                col[(unsigned)r + 64u] = kprog_eval_k<LEAF>(a.prog, s[1][b], u[1][b]);
                col += a.ldo;
            }
            return;
        }
#pragma unroll
        for (int b = 0; b < 16; ++b)
            emit(b, kprog_eval_k<LEAF>(a.prog, s[0][b], u[0][b]), kprog_eval_k<LEAF>(a.prog, s[1][b], u[1][b]));
    } else {
        // Any program: its body (the stack machine over nine leaf kinds) is too large to unroll 32 times, and a rolled loop that
        // indexes s[h][b] dynamically sends the accumulators through scratch memory (measured: 4x the algorithmic bytes written).
        // So the loop stays rolled, always consumes elements 0 and 1, and then rotates the register arrays (static indices only);
        // two columns (four pairs) per turn so that the chains of dependent f64 operations overlap four ways.
#pragma unroll 1
        for (int b = 0; b < 16; b += 2) {
            double k[2][2];
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int h = 0; h < 2; ++h) k[e][h] = kprog_eval(a.prog, s[h][e], u[h][e]);
            emit(b, k[0][0], k[0][1]);
            emit(b + 1, k[1][0], k[1][1]);
#pragma unroll
            for (int i = 0; i < 14; ++i) {
                s[0][i] = s[0][i + 2];
                s[1][i] = s[1][i + 2];
                u[0][i] = u[0][i + 2];
                u[1][i] = u[1][i + 2];
            }
        }
    }
}

__global__ __launch_bounds__(256) void gram_diag_kernel(const fr_kprog prog, const double* X, int64_t n, int64_t ldx,
                                                        int64_t d, double add, double* out)
{
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0.0, u = 0.0;
    for (int64_t c = 0; c < d; ++c) {
        const double x = X[i + c * ldx];
        const double diff = x - x;
        s = s + diff * diff;
        u = u + x * x;
    }
    out[i] = kprog_eval(prog, s, u) + add;
}

// The epilogue of predict_variance / predict_mean_variance in ONE launch (mod.rs:266-270, 313-319):
//   out[j] = k(x*_j, x*_j) - sum_i U[i, j] V[i, j]        (U == V = L^-1 K*: the column's squared norm)
// one workgroup per query point: the column dot product in a fixed order, the prior variance by the kernel program.
// (Round 2 ran this as three launches -- column norms, gram_diag, axpby -- and a temporary.)
__global__ __launch_bounds__(256) void variance_epilogue_kernel(const fr_kprog prog, const double* __restrict__ Xq, int64_t ldq, int64_t d,
                                                                const double* __restrict__ U, int64_t ldu, const double* __restrict__ V,
                                                                int64_t ldv, int64_t n, double* __restrict__ out)
{
#pragma clang fp contract(off)
    __shared__ double red[4];
    const int64_t j = blockIdx.x;
    const double* u = U + j * ldu;
    const double* v = V + j * ldv;
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) acc += u[i] * v[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int w = 0; w < 4; ++w) tot += red[w];
        double s2 = 0.0, uu = 0.0;
        for (int64_t c = 0; c < d; ++c) {
            const double x = Xq[j + c * ldq];
            const double diff = x - x;
            s2 = s2 + diff * diff;
            uu = uu + x * x;
        }
        out[j] = kprog_eval(prog, s2, uu) - tot;
    }
}

int launch_variance_epilogue(fr_ctx* ctx, const fr_kprog& prog, const double* Xq, int64_t m, int64_t ldq, int64_t d, const double* U,
                             int64_t ldu, const double* V, int64_t ldv, int64_t n, double* out)
{
    if (m <= 0) return FR_OK;
    ProfScope ps(ctx, FR_PROF_REDUCE, 2.0 * (double)n * m, 8.0 * (double)n * m * (U == V ? 1.0 : 2.0));
    hipLaunchKernelGGL(variance_epilogue_kernel, dim3((unsigned)m), dim3(256), 0, ctx->ls, prog, Xq, ldq, d, U, ldu, V, ldv, n, out);
    FR_HIP(ctx, hipGetLastError());
    return FR_OK;
}

// K3: sum over strictly-lower pairs of ||x_i - x_j||; per-block partials, then one reducing block
__device__ __forceinline__ double block_reduce_sum(double v, double* red /* >= 4 doubles of LDS */)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    double tot = 0.0;
    const int nw = (blockDim.x + 63) >> 6;
    for (int i = 0; i < nw; ++i) tot += red[i];
    __syncthreads();
    return tot;
}

__global__ __launch_bounds__(256) void pairdist_kernel(const double* X, int64_t n, int64_t ldx, int64_t d,
                                                       double* partials)
{
#pragma clang fp contract(off)
    __shared__ double XA[GT_DC][GT_M];
    __shared__ double XB[GT_DC][GT_N];
    __shared__ double red[4];
    int64_t ti, tj;
    sym_tile((int64_t)blockIdx.x, ti, tj);
    const int64_t i0 = ti * GT_M, j0 = tj * GT_N;
    const int t = threadIdx.x, r = t & 63, g = t >> 6;
    double s[2][16];
#pragma unroll
    for (int b = 0; b < 16; ++b) s[0][b] = s[1][b] = 0.0;
    for (int64_t c0 = 0; c0 < d; c0 += GT_DC) {
        const int dc = (int)((d - c0) < GT_DC ? (d - c0) : GT_DC);
#pragma unroll
        for (int p = 0; p < GT_DC / 2; ++p) {
            const int row = t & 127, c = (t >> 7) + 2 * p;
            const int64_t gi = i0 + row;
            XA[c][row] = (c < dc && gi < n) ? X[gi + (c0 + c) * ldx] : 0.0;
        }
#pragma unroll
        for (int p = 0; p < GT_DC / 4; ++p) {
            const int row = t & 63, c = (t >> 6) + 4 * p;
            const int64_t gj = j0 + row;
            XB[c][row] = (c < dc && gj < n) ? X[gj + (c0 + c) * ldx] : 0.0;
        }
        __syncthreads();
        for (int c = 0; c < dc; ++c) {
            const double xa0 = XA[c][r], xa1 = XA[c][r + 64];
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                const double xb = XB[c][g * 16 + b];
                const double d0 = xa0 - xb, d1 = xa1 - xb;
                s[0][b] = s[0][b] + d0 * d0;
                s[1][b] = s[1][b] + d1 * d1;
            }
        }
        __syncthreads();
    }
    double acc = 0.0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int64_t gi = i0 + r + 64 * h;
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            const int64_t gj = j0 + g * 16 + b;
            if (gi < n && gj < gi) acc += sqrt(s[h][b]);
        }
    }
    const double tot = block_reduce_sum(acc, red);
    if (t == 0) partials[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void reduce_partials_kernel(const double* partials, int64_t n, double scale,
                                                              double* out)
{
    __shared__ double red[4];
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) acc += partials[i];
    const double tot = block_reduce_sum(acc, red);
    if (threadIdx.x == 0) out[0] = tot * scale;
}

int kprog_check(fr_ctx* ctx, const fr_kprog* p)
{
    if (!p) return set_err(ctx, FR_INVALID_ARGUMENT, "null kernel program");
    if (p->nops < 1 || p->nops > FR_KPROG_MAX_OPS) return set_err(ctx, FR_UNSUPPORTED_KERNEL, "kernel program length %d", p->nops);
    int depth = 0;
    for (int i = 0; i < p->nops; ++i) {
        const int k = p->ops[i].kind;
        if (kind_is_leaf(k)) {
            if (p->ops[i].nparams != leaf_nvalues(k))
                return set_err(ctx, FR_UNSUPPORTED_KERNEL, "kernel op %d: kind %d expects %d parameters, got %d", i, k,
                               leaf_nvalues(k), p->ops[i].nparams);
            if (++depth > 8) return set_err(ctx, FR_UNSUPPORTED_KERNEL, "kernel program too deep");
        } else if (k == FR_K_SUM || k == FR_K_PROD) {
            if (depth < 2) return set_err(ctx, FR_UNSUPPORTED_KERNEL, "malformed kernel program (op %d)", i);
            --depth;
        } else {
            return set_err(ctx, FR_UNSUPPORTED_KERNEL, "unknown kernel kind %d", k);
        }
    }
    if (depth != 1) return set_err(ctx, FR_UNSUPPORTED_KERNEL, "malformed kernel program (final depth %d)", depth);
    return FR_OK;
}

static int launch_gram(fr_ctx* ctx, const GramArgs& a, int64_t nblocks, int needs, double pairs)
{
    if (nblocks <= 0) return FR_OK;
    if (nblocks > 0x7fffffffLL) return set_err(ctx, FR_INVALID_ARGUMENT, "Gram grid too large");
    ProfScope ps(ctx, FR_PROF_GRAM, pairs * (3.0 * (double)a.d + 20.0), pairs * 8.0);
    dim3 grid((unsigned)nblocks), block(256);
    const int leaf = a.prog.nops == 1 ? a.prog.ops[0].kind : -1;
#define FR_GRAM_LEAF(KIND, MODE_)                                                            \
    case KIND: hipLaunchKernelGGL((gram_kernel<MODE_, KIND>), grid, block, 0, ctx->ls, a); break;
    switch (leaf) {
        FR_GRAM_LEAF(FR_K_LINEAR, NEED_U)
        FR_GRAM_LEAF(FR_K_POLYNOMIAL, NEED_U)
        FR_GRAM_LEAF(FR_K_SQUAREDEXP, NEED_S)
        FR_GRAM_LEAF(FR_K_EXPONENTIAL, NEED_S)
        FR_GRAM_LEAF(FR_K_MATERN1, NEED_S)
        FR_GRAM_LEAF(FR_K_MATERN2, NEED_S)
        FR_GRAM_LEAF(FR_K_HYPERTAN, NEED_U)
        FR_GRAM_LEAF(FR_K_MULTIQUADRIC, NEED_S)
        FR_GRAM_LEAF(FR_K_RATIONALQUADRATIC, NEED_S)
    default:
        if (needs == NEED_S)
            hipLaunchKernelGGL((gram_kernel<NEED_S, -1>), grid, block, 0, ctx->ls, a);
        else if (needs == NEED_U)
            hipLaunchKernelGGL((gram_kernel<NEED_U, -1>), grid, block, 0, ctx->ls, a);
        else
            hipLaunchKernelGGL((gram_kernel<NEED_S | NEED_U, -1>), grid, block, 0, ctx->ls, a);
    }
#undef FR_GRAM_LEAF
    FR_HIP(ctx, hipGetLastError());
    return FR_OK;
}

int launch_gram_cross(fr_ctx* ctx, const fr_kprog& prog, const double* A, int64_t n1, int64_t lda, const double* B,
                      int64_t n2, int64_t ldb, int64_t d, double* out, int64_t ldo)
{
    if (n1 == 0 || n2 == 0) return FR_OK;
    GramArgs a;
    a.prog = prog;
    a.A = A;
    a.n1 = n1;
    a.lda = lda;
    a.B = B;
    a.n2 = n2;
    a.ldb = ldb;
    a.d = d;
    a.out = out;
    a.ldo = ldo;
    a.sym = 0;
    a.dot_vec = nullptr;
    a.dot_part = nullptr;
    a.noise2 = 0.0;
    a.own_world = 1;
    a.own_rank = 0;
    a.own_nb = 1;
    a.tiles_n = (n2 + GT_N - 1) / GT_N;
    const int64_t tiles_m = (n1 + GT_M - 1) / GT_M;
    return launch_gram(ctx, a, tiles_m * a.tiles_n, kprog_needs(prog), (double)n1 * (double)n2);
}

// out[j] += sum over the row tiles of part[ti * m + j], in tile order (deterministic)
__global__ __launch_bounds__(256) void gram_dot_reduce_kernel(const double* __restrict__ part, int64_t tiles, int64_t m,
                                                             double* __restrict__ out)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    double acc = out[j];
    for (int64_t ti = 0; ti < tiles; ++ti) acc = acc + part[ti * m + j];
    out[j] = acc;
}

// out (n2, holding the prior) += K(A, B)^T v without materialising the n1 x n2 covariance: K1's tiles are multiplied by v
// as they are produced (per-tile partial sums, then one deterministic reduction).
int launch_gram_cross_dot(fr_ctx* ctx, const fr_kprog& prog, const double* A, int64_t n1, int64_t lda, const double* B,
                          int64_t n2, int64_t ldb, int64_t d, const double* v, double* out)
{
    if (n1 == 0 || n2 == 0) return FR_OK;
    const int64_t tiles_m = (n1 + GT_M - 1) / GT_M;
    WsGuard w(ctx);
    double* part = w.get(sizeof(double) * (size_t)tiles_m * (size_t)n2);
    if (!part) return FR_OUT_OF_MEMORY;
    GramArgs a;
    a.prog = prog;
    a.A = A;
    a.n1 = n1;
    a.lda = lda;
    a.B = B;
    a.n2 = n2;
    a.ldb = ldb;
    a.d = d;
    a.out = nullptr;
    a.ldo = 0;
    a.sym = 0;
    a.dot_vec = v;
    a.dot_part = part;
    a.noise2 = 0.0;
    a.own_world = 1;
    a.own_rank = 0;
    a.own_nb = 1;
    a.tiles_n = (n2 + GT_N - 1) / GT_N;
    FR_TRY(launch_gram(ctx, a, tiles_m * a.tiles_n, kprog_needs(prog), (double)n1 * (double)n2));
    hipLaunchKernelGGL(gram_dot_reduce_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, ctx->ls, part, tiles_m, n2, out);
    FR_HIP(ctx, hipGetLastError());
    return FR_OK;
}

int launch_gram_sym(fr_ctx* ctx, const fr_kprog& prog, const double* X, int64_t n, int64_t ldx, int64_t d,
                    double noise2, double* out, int64_t ldo, int own_world, int own_rank, int64_t own_nb)
{
    if (n == 0) return FR_OK;
    GramArgs a;
    a.prog = prog;
    a.A = X;
    a.n1 = n;
    a.lda = ldx;
    a.B = X;
    a.n2 = n;
    a.ldb = ldx;
    a.d = d;
    a.out = out;
    a.ldo = ldo;
    a.sym = 1;
    a.dot_vec = nullptr;
    a.dot_part = nullptr;
    a.noise2 = noise2;
    a.tiles_n = 0;
    a.own_world = own_world;
    a.own_rank = own_rank;
    a.own_nb = own_nb > 0 ? own_nb : 1;
    const int64_t nbk = (n + GT_M - 1) / GT_M;
    return launch_gram(ctx, a, nbk * (nbk + 1), kprog_needs(prog), 0.5 * (double)n * (double)(n + 1));
}

int launch_gram_diag(fr_ctx* ctx, const fr_kprog& prog, const double* X, int64_t n, int64_t ldx, int64_t d, double add,
                     double* out)
{
    if (n == 0) return FR_OK;
    ProfScope ps(ctx, FR_PROF_GRAM, 0.0, (double)n * 8.0);
    hipLaunchKernelGGL(gram_diag_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->ls, prog, X, n, ldx,
                       d, add, out);
    FR_HIP(ctx, hipGetLastError());
    return FR_OK;
}

int launch_pairwise_distance_sum(fr_ctx* ctx, const double* X, int64_t n, int64_t ldx, int64_t d, double* out_dev)
{
    const int64_t nbk = (n + GT_M - 1) / GT_M;
    const int64_t nblocks = nbk * (nbk + 1);
    WsGuard part(ctx);
    double* partials = part.get(sizeof(double) * (size_t)nblocks);
    if (!partials) return FR_OUT_OF_MEMORY;
    const double npairs = (double)((n * n - n) / 2);  // kernel.rs:108-109
    {
        ProfScope ps(ctx, FR_PROF_GRAM, npairs * (3.0 * (double)d + 10.0), (double)n * (double)d * 8.0);
        hipLaunchKernelGGL(pairdist_kernel, dim3((unsigned)nblocks), dim3(256), 0, ctx->ls, X, n, ldx, d, partials);
        FR_HIP(ctx, hipGetLastError());
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, ctx->ls, (const double*)partials, nblocks,
                           1.0 / npairs, out_dev);
        FR_HIP(ctx, hipGetLastError());
    }
    return FR_OK;
}

}  // namespace fr

using namespace fr;

extern "C" {

int fr_gram(fr_ctx* ctx, const fr_kprog* kernel, const double* A, int64_t n1, int64_t lda, const double* B, int64_t n2,
            int64_t ldb, int64_t d, double* out, int64_t ldo)
{
    if (!ctx) return FR_INVALID_ARGUMENT;
    FR_LOCK(ctx);
    FR_HIP(ctx, hipSetDevice(ctx->device));
    FR_TRY(kprog_check(ctx, kernel));
    if (d < 0) return set_err(ctx, FR_SHAPE, "negative feature count");
    Staged a(ctx), b(ctx), o(ctx);
    FR_TRY(a.in(A, n1, d, lda));
    FR_TRY(b.in(B, n2, d, ldb));
    FR_TRY(o.out(out, n1, n2, ldo));
    FR_TRY(launch_gram_cross(ctx, *kernel, a.dev, n1, a.ld, b.dev, n2, b.ld, d, o.dev, o.ld));
    return o.commit();
}

int fr_mean_pairwise_distance(fr_ctx* ctx, const double* X, int64_t n, int64_t ldx, int64_t d, double* out)
{
    if (!ctx || !out) return FR_INVALID_ARGUMENT;
    FR_LOCK(ctx);
    FR_HIP(ctx, hipSetDevice(ctx->device));
    if (n < 2) {  // 0/0 in the reference (kernel.rs:112)
        *out = std::nan("");
        return FR_OK;
    }
    Staged x(ctx);
    FR_TRY(x.in(X, n, d, ldx));
    WsGuard res(ctx);
    double* r = res.get(sizeof(double));
    if (!r) return FR_OUT_OF_MEMORY;
    FR_TRY(launch_pairwise_distance_sum(ctx, x.dev, n, x.ld, d, r));
    FR_HIP(ctx, hipMemcpyAsync(out, r, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    FR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FR_OK;
}

}  // extern "C"
