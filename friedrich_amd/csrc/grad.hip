// grad.hip -- K2 + K8: the per-iteration reductions of the marginal-likelihood gradient
// (src/gaussian_process/optimizer.rs:24-60 `gradient_marginal_likelihood`, :159-203
// `scaled_gradient_marginal_likelihood`) without materialising the p gradient Gram matrices of
// src/algebra/mod.rs:129-155 (`make_gradient_covariance_matrices`, p x n x n f64 in the reference):
//
//   g_q = 1/2 ( alpha^T G_q alpha [/ scale]  -  tr(K^-1 G_q) ),   G_q[i,j] = d k(x_i, x_j) / d theta_q
//       = 1/2 sum_{i >= j} w_ij g_q(i,j) ( alpha_i alpha_j [/ scale] - Kinv_ij ),   w = 1 on the diagonal, 2 below it
//
// K8: K^-1 = W^T W with W = L^-1 (one GEMM-recast TRSM on the identity + one lower-triangular FP64-MFMA GEMM);
//     the reference's Cholesky::inverse() (optimizer.rs:32,169) is an identity solve_mut.
// K2: tile reduction over the lower triangle, same 128 x 64 tiling / LDS staging as the Gram kernel; each pair's
//     gradient vector is evaluated on the fly from (||x-y||^2, x.y) with the formulas of src/parameters/kernel.rs
//     `gradient` bodies (file:line at each case), including the reference's quirks (Matern2 grad_ls as written,
//     Multiquadric declaring 2 parameters but yielding 1 gradient).
#include "fr_internal.hpp"
#include <vector>
#include "kprog_device.hpp"

namespace fr {

constexpr int GR_M = 128, GR_N = 64, GR_DC = 16;
constexpr int MAXG = 24;  // 8 leaves x 3 values

__device__ __forceinline__ double signum_d(double x) { return (x != x) ? x : copysign(1.0, x); }
__device__ __forceinline__ double pow3(double x) { return x * (x * x); }  // powi(3): x * x^2 (compiler-rt __powidf2)

// gradient of one leaf; returns the number of values written (kernel.rs `gradient` bodies)
template <int KIND>
__device__ __forceinline__ int leaf_grad_k(const fr_kernel_op& op, double s, double u, double* g)
{
#pragma clang fp contract(off)
    const double p0 = op.params[0], p1 = op.params[1], p2 = op.params[2];
    const int kind = KIND >= 0 ? KIND : op.kind;
    switch (kind) {
    case FR_K_LINEAR:  // :384-391
        g[0] = 1.0;
        return 1;
    case FR_K_POLYNOMIAL: {  // :459-472
        const double inner = p0 * u + p1;
        const double grad_c = p2 * pow(inner, p2 - 1.0);
        g[0] = u * grad_c;
        g[1] = grad_c;
        g[2] = log(inner) * pow(inner, p2);
        return 3;
    }
    case FR_K_SQUAREDEXP: {  // :563-576
        const double e = exp_fast(-div_uniform(s, 2.0 * p0 * p0));
        g[0] = div_uniform(s * fabs(p1) * e, pow3(p0));
        g[1] = signum_d(p1) * e;
        return 2;
    }
    case FR_K_EXPONENTIAL: {  // :668-681
        const double r = sqrt(s);
        const double e = exp_fast(-div_uniform(r, 2.0 * p0 * p0));
        g[0] = div_uniform(r * fabs(p1) * e, pow3(p0));
        g[1] = signum_d(p1) * e;
        return 2;
    }
    case FR_K_MATERN1: {  // :774-788
        const double l = fabs(p0), r = sqrt(s);
        const double x = div_uniform(sqrt(3.0) * r, l);
        const double e = exp_fast(-x);
        g[0] = div_uniform(3.0 * fabs(p1) * (r * r) * e, pow3(p0));
        g[1] = signum_d(p1) * (1.0 + x) * e;
        return 2;
    }
    case FR_K_MATERN2: {  // :881-900 (x uses the signed ls; grad_ls as written in the reference)
        const double l = fabs(p0), r = sqrt(s);
        const double x = div_uniform(sqrt(5.0) * r, p0);
        const double e = exp_fast(-x);
        g[0] = signum_d(p0) * fabs(p1) * ((2.0 * l / 3.0 + 1.0) + r * sqrt(5.0) * (((l * l) / 3.0 + l + 1.0) / (l * l))) * e;
        g[1] = signum_d(p1) * (1.0 + x + div_uniform(5.0 * r * r, 3.0 * l * l)) * e;
        return 2;
    }
    case FR_K_HYPERTAN: {  // :979-989
        const double ch = cosh(p0 * u + p1);
        const double grad_c = 1.0 / (ch * ch);
        g[0] = u * grad_c;
        g[1] = grad_c;
        return 2;
    }
    case FR_K_MULTIQUADRIC:  // :1052-1059
        g[0] = p0 / hypot(sqrt(s), p0);
        return 1;
    case FR_K_RATIONALQUADRATIC: {  // :1125-1145
        const double alpha = p0, l = fabs(p1), l2 = l * l;
        g[0] = pow((s + 2.0 * l2 * alpha) / (l2 * alpha), -alpha) *
               (pow(2.0, alpha) * (1.0 - log((s + 2.0 * l2 * alpha) / (2.0 * l2 * alpha))) -
                (l2 * pow(2.0, alpha + 1.0) * alpha) / (s + 2.0 * l2 * alpha));
        g[1] = s * pow(s / (2.0 * alpha * l * l) + 1.0, -alpha - 1.0) / pow3(p1);
        return 2;
    }
    default: return 0;
    }
}

__device__ inline int leaf_grad(const fr_kernel_op& op, double s, double u, double* g) { return leaf_grad_k<-1>(op, s, u, g); }

// number of gradient values of a leaf kind (compile-time twin of leaf_nvalues)
constexpr int leaf_ng(int kind) { return kind == FR_K_POLYNOMIAL ? 3 : ((kind == FR_K_LINEAR || kind == FR_K_MULTIQUADRIC) ? 1 : 2); }

// gradient of a whole program: leaves write in program order (= k1-then-k2 concatenation, kernel.rs:168-171);
// Prod applies the product rule g1*k2, g2*k1 (:252-262)
__device__ inline int kprog_grad(const fr_kprog& p, double s, double u, double* g)
{
    double val[8];
    int gstart[8], glen[8];
    int sp = 0, ng = 0;
    for (int i = 0; i < p.nops; ++i) {
        const int k = p.ops[i].kind;
        if (k == FR_K_SUM || k == FR_K_PROD) {
            const int b = sp - 1, a = sp - 2;
            if (k == FR_K_PROD) {
                for (int q = 0; q < glen[a]; ++q) g[gstart[a] + q] *= val[b];
                for (int q = 0; q < glen[b]; ++q) g[gstart[b] + q] *= val[a];
                val[a] = val[a] * val[b];
            } else {
                val[a] = val[a] + val[b];
            }
            glen[a] += glen[b];
            --sp;
        } else {
            val[sp] = leaf_eval(p.ops[i], s, u);
            gstart[sp] = ng;
            glen[sp] = leaf_grad(p.ops[i], s, u, g + ng);
            ng += glen[sp];
            ++sp;
        }
    }
    return ng;
}

struct GradArgs {
    fr_kprog prog;
    const double* X;
    int64_t n, ldx, d;
    const double* alpha;
    const double* Kinv;
    int64_t ldk;
    const double* ya;  // device: y . alpha -- the scaled variant divides the alpha alpha^T term by scale = y . alpha / n (optimizer.rs:174); formed
                       // per thread from this word, so that the host does not have to read it back (a synchronisation) in front of the launch
    double nd;         // n as a double
    int scaled;
    double alpha_w;    // weight of the alpha_i alpha_j term: 1; sharded: 1 on rank 0, 0 elsewhere (Kinv is then a rank's PARTIAL of K^-1)
    int ng;
    double* partials;  // [nblocks][ng]
};

__device__ __forceinline__ void grad_sym_tile(int64_t t, int64_t& bi, int64_t& tj)
{
    int64_t b = (int64_t)((sqrt(4.0 * (double)t + 1.0) - 1.0) * 0.5);
    while (b * (b + 1) > t) --b;
    while ((b + 1) * (b + 2) <= t) ++b;
    bi = b;
    tj = t - b * (b + 1);
}

// MODE: which pair statistics the kernel program needs (kprog_needs): squared distance (NEED_S), dot product (NEED_U) or both
// -- as in K1, a kernel without the other accumulator saves its VALU work and the spills of 64 more accumulator registers.
// LEAF >= 0: the program is that single built-in kernel: straight-line epilogue, static accumulator count (as in gram.hip).
template <int MODE, int LEAF>
__global__ __launch_bounds__(256, 2) void grad_reduce_kernel(const GradArgs a)
{
#pragma clang fp contract(off)
    __shared__ double XA[GR_DC][GR_M];
    __shared__ double XB[GR_DC][GR_N];
    __shared__ double red[4][MAXG];
    int64_t ti, tj;
    grad_sym_tile((int64_t)blockIdx.x, ti, tj);
    const int64_t i0 = ti * GR_M, j0 = tj * GR_N;
    const int t = threadIdx.x, r = t & 63, g = t >> 6;
    double s[2][16], u[2][16];
#pragma unroll
    for (int b = 0; b < 16; ++b) {
        s[0][b] = s[1][b] = 0.0;
        u[0][b] = u[1][b] = 0.0;
    }
    for (int64_t c0 = 0; c0 < a.d; c0 += GR_DC) {
        const int dc = (int)((a.d - c0) < GR_DC ? (a.d - c0) : GR_DC);
#pragma unroll
        for (int p = 0; p < GR_DC / 2; ++p) {
            const int row = t & 127, c = (t >> 7) + 2 * p;
            const int64_t gi = i0 + row;
            XA[c][row] = (c < dc && gi < a.n) ? a.X[gi + (c0 + c) * a.ldx] : 0.0;
        }
#pragma unroll
        for (int p = 0; p < GR_DC / 4; ++p) {
            const int row = t & 63, c = (t >> 6) + 4 * p;
            const int64_t gj = j0 + row;
            XB[c][row] = (c < dc && gj < a.n) ? a.X[gj + (c0 + c) * a.ldx] : 0.0;
        }
        __syncthreads();
        for (int c = 0; c < dc; ++c) {
            const double xa0 = XA[c][r], xa1 = XA[c][r + 64];
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                const double xb = XB[c][g * 16 + b];
                if (MODE & NEED_S) {
                    const double d0 = xa0 - xb, d1 = xa1 - xb;
                    s[0][b] = s[0][b] + d0 * d0;
                    s[1][b] = s[1][b] + d1 * d1;
                }
                if (MODE & NEED_U) {
                    u[0][b] = u[0][b] + xa0 * xb;
                    u[1][b] = u[1][b] + xa1 * xb;
                }
            }
        }
        __syncthreads();
    }
    constexpr int NG = LEAF >= 0 ? leaf_ng(LEAF) : MAXG;
    const double inv_scale = a.scaled ? 1.0 / (a.ya[0] / a.nd) : 1.0;  // (the host's two divisions, in the same order)
    double acc[NG];
#pragma unroll
    for (int q = 0; q < NG; ++q) acc[q] = 0.0;
    double ai[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int64_t gi = i0 + r + 64 * h;
        ai[h] = (gi < a.n) ? a.alpha[gi] : 0.0;
    }
    if constexpr (LEAF >= 0) {
        // gradients evaluated unconditionally (padding rows / columns hold zeros: finite inputs), the pair's weight is zero
        // outside the lower triangle; the coefficient's loads are issued for all 32 pairs before the first evaluation
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            const int64_t gj = j0 + g * 16 + b;
            const double aj = gj < a.n ? a.alpha[gj] : 0.0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int64_t gi = i0 + r + 64 * h;
                const bool in = gi < a.n && gj <= gi;  // lower triangle incl. diagonal (algebra/mod.rs:142-151)
                double gv[NG];
                (void)leaf_grad_k<LEAF>(a.prog.ops[0], s[h][b], u[h][b], gv);
                const double kinv = in ? a.Kinv[gi + gj * a.ldk] : 0.0;
                const double w = (gi == gj) ? 1.0 : 2.0;
                const double coef = w * (ai[h] * aj * inv_scale * a.alpha_w - kinv);
#pragma unroll
                for (int q = 0; q < NG; ++q)
                    if (in) acc[q] += coef * gv[q];
            }
        }
    } else {
        // The gradient program is too large to be unrolled 32 times; a rolled loop indexing s[h][b] dynamically would send the
        // accumulators through scratch memory (see gram.hip).  The column loop stays rolled, consumes element 0 and rotates the
        // register arrays; the per-parameter loops are unrolled with a guard so that acc / gv keep static indices.
#pragma unroll 1
        for (int b = 0; b < 16; ++b) {
            const int64_t gj = j0 + g * 16 + b;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int64_t gi = i0 + r + 64 * h;
                if (gi < a.n && gj <= gi) {  // lower triangle incl. diagonal (algebra/mod.rs:142-151)
                    double gv[MAXG];
                    const int ng = kprog_grad(a.prog, s[h][0], u[h][0], gv);
                    const double w = (gi == gj) ? 1.0 : 2.0;
                    const double coef = w * (ai[h] * a.alpha[gj] * inv_scale * a.alpha_w - a.Kinv[gi + gj * a.ldk]);
#pragma unroll
                    for (int q = 0; q < MAXG; ++q)
                        if (q < ng) acc[q] += coef * gv[q];
                }
            }
#pragma unroll
            for (int i = 0; i < 15; ++i) {
                s[0][i] = s[0][i + 1];
                s[1][i] = s[1][i + 1];
                u[0][i] = u[0][i + 1];
                u[1][i] = u[1][i + 1];
            }
        }
    }
    // block reduction of the ng accumulators
#pragma unroll
    for (int q = 0; q < NG; ++q) {
        if (q < a.ng) {
            double v = acc[q];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if ((t & 63) == 0) red[t >> 6][q] = v;
        }
    }
    __syncthreads();
    if (t < a.ng) a.partials[(int64_t)blockIdx.x * a.ng + t] = red[0][t] + red[1][t] + red[2][t] + red[3][t];
}

// out[q] = 1/2 sum_b partials[b][q];  out[ng] = trace(Kinv), out[ng+1] = alpha . alpha
__global__ __launch_bounds__(256) void grad_finish_kernel(const double* partials, int64_t nblocks, int ng, const double* Kinv,
                                                          int64_t ldk, const double* alpha, int64_t n, double* out)
{
    __shared__ double red[4];
    for (int q = 0; q < ng + 2; ++q) {
        double v = 0.0;
        if (q < ng) {
            for (int64_t b = threadIdx.x; b < nblocks; b += blockDim.x) v += partials[b * ng + q];
        } else if (q == ng) {
            for (int64_t i = threadIdx.x; i < n; i += blockDim.x) v += Kinv[i + i * ldk];
        } else {
            for (int64_t i = threadIdx.x; i < n; i += blockDim.x) v += alpha[i] * alpha[i];
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) out[q] = (red[0] + red[1] + red[2] + red[3]) * (q < ng ? 0.5 : 1.0);
        __syncthreads();
    }
}

static int kprog_counts(const fr_kprog& p, int* nb_parameters, int* nb_gradients)
{
    int np = 0, ng = 0;
    for (int i = 0; i < p.nops; ++i) {
        const int k = p.ops[i].kind;
        if (!kind_is_leaf(k)) continue;
        ng += leaf_nvalues(k);
        np += (k == FR_K_MULTIQUADRIC) ? 2 : leaf_nvalues(k);  // kernel.rs:1039-1042 declares 2
    }
    *nb_parameters = np;
    *nb_gradients = ng;
    return FR_OK;
}

}  // namespace fr

using namespace fr;

static int grad_terms_impl(fr_chol* c, const fr_kprog* kernel, const double* y, double noise, int scaled, double* out_grad,
                           double* out_scale);

extern "C" int fr_grad_terms(fr_chol* c, const fr_kprog* kernel, const double* y, double noise, int scaled,
                             double* out_grad, double* out_scale)
{
    if (!c || !out_grad) return FR_INVALID_ARGUMENT;
    FR_LOCK(c->ctx);
    return solve_retry(c->ctx, [&]() -> int { return grad_terms_impl(c, kernel, y, noise, scaled, out_grad, out_scale); });
}

static int grad_terms_impl(fr_chol* c, const fr_kprog* kernel, const double* y, double noise, int scaled, double* out_grad,
                           double* out_scale)
{
    fr_ctx* ctx = c->ctx;
    FR_HIP(ctx, hipSetDevice(ctx->device));
    FR_TRY(kprog_check(ctx, kernel));
    const int64_t n = c->n;
    int np = 0, ng = 0;
    kprog_counts(*kernel, &np, &ng);
    if (ng > MAXG) return set_err(ctx, FR_UNSUPPORTED_KERNEL, "too many kernel parameters for the device gradient");
    if (n == 0) {
        for (int q = 0; q < np + (scaled ? 0 : 1); ++q) out_grad[q] = 0.0;
        if (out_scale) *out_scale = std::nan("");
        return FR_OK;
    }
    if (!y) return set_err(ctx, FR_INVALID_ARGUMENT, "null training-output vector (y)");
    const int64_t ld = round_up(n, kAlign);
    // Sharded: K^-1 = W^T W = sum over the ROWS k of W = L^-1 of the outer products w_k w_k^T, and every
    // term of the gradient that involves K^-1 -- tr(K^-1 G_q), tr(K^-1) -- is linear in it.  So the rows of W are dealt to the
    // ranks in chunks; a rank forms the rows of its chunks (W(K, :)^T = the backward solve  L11^T X = [0; I]  on the leading
    // block that ends with chunk K: the rows of W are zero to the right of their diagonal), accumulates its PARTIAL
    // K^-1 = sum_K X X^T, runs the fused reductions on it (the alpha alpha^T term is counted on rank 0 only), and one all-gather
    // of ng + 2 scalars per rank, summed in rank order on every rank, finishes the job: no matrix ever travels.  Chunks are dealt
    // in a snake over DESCENDING row ranges (the cost of a chunk grows with the square of where it ends).  Ill-conditioned
    // (refined) handles and small factors: every rank computes the whole, no collective.
    const int Wn = ctx->world, me = ctx->rank;
    const bool sharded = Wn > 1 && c->collective && n >= ctx->grad_shard_min && n >= 1024 && !c->refine;
    const int64_t cs = n >= 8192 ? 2048 : 512;  // rows per chunk
    WsGuard wg(ctx), kg(ctx), vg(ctx), pg(ctx), yg(ctx);
    double* W = wg.get(sizeof(double) * (size_t)ld * (size_t)(sharded ? cs : n));
    double* Kinv = kg.get(sizeof(double) * (size_t)ld * (size_t)n);
    double* vec = vg.get(sizeof(double) * (size_t)(ld + (MAXG + 8) * 66));
    double* ydev = yg.get(sizeof(double) * (size_t)ld);
    const int64_t nbk = (n + GR_M - 1) / GR_M;
    const int64_t nblocks = nbk * (nbk + 1);
    double* partials = pg.get(sizeof(double) * (size_t)(nblocks * (ng > 0 ? ng : 1)));
    const bool have = W && Kinv && vec && ydev && partials;
    if (!have) {
        bool all_ok = false;
        if (sharded) (void)comm_agree(ctx, false, &all_ok);  // (the peers must not wait for this rank's partials)
        return FR_OUT_OF_MEMORY;
    }
    double* alpha = vec;
    double* outs = vec + ld;  // [ng] gradient halves, trace, alpha.alpha, y.alpha; sharded: followed by every rank's ng + 2
    // alpha = K^-1 y (optimizer.rs:33, 171) -- first: these are the only persistent kernels of the call, and when sharded a
    // timed-out hand-off must send this rank into its repeat (solve_retry) BEFORE it has taken part in any collective, so that
    // every rank issues each collective of the call exactly once
    {
        const bool dev = is_device_ptr(y);
        FR_HIP(ctx, hipMemcpyAsync(alpha, y, sizeof(double) * (size_t)n, dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                                   ctx->stream));
        if (!dev) FR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    FR_HIP(ctx, hipMemcpyAsync(ydev, alpha, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
    FR_TRY(trsm_lower_fwd(ctx, c, n, alpha, 1, ld, FR_PROF_GEMM_SOLVE));
    FR_TRY(trsm_lower_bwd(ctx, c, n, alpha, 1, ld, FR_PROF_GEMM_SOLVE));
    // scale = y . alpha / n (optimizer.rs:174): y . alpha stays on the device until the end (round 5: reading it back here was a
    // synchronisation in the middle of every optimizer iteration)
    FR_TRY(launch_col_dot(ctx, ydev, ld, alpha, ld, n, 1, outs + ng + 2));
    if (sharded) {
        FR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        FR_TRY(check_status_word(ctx));
        bool all_ok = true;
        FR_TRY(comm_agree(ctx, true, &all_ok));
        if (!all_ok) return set_err(ctx, FR_OUT_OF_MEMORY, "a peer rank could not allocate its gradient workspace");
        const int64_t nc = (n + cs - 1) / cs;
        FR_HIP(ctx, hipMemsetAsync(Kinv, 0, sizeof(double) * (size_t)ld * (size_t)n, ctx->ls));
        for (int64_t t = 0; t < nc; ++t) {  // t-th largest chunk
            const int owner = (int)(((t / Wn) & 1) ? (Wn - 1 - t % Wn) : (t % Wn));
            if (owner != me) continue;
            const int64_t j = nc - 1 - t, k0 = j * cs, k1 = (k0 + cs < n) ? k0 + cs : n, m = k1 - k0;
            if (k0 > 0) FR_TRY(launch_fill(ctx, W, k0, m, ld, 0.0));
            FR_TRY(launch_set_identity(ctx, W + k0, m, ld));
            FR_TRY(trsm_lower_bwd_leading(ctx, c, k1, W, m, ld, FR_PROF_GEMM_SOLVE));
            GemmDesc g;
            g.M = k1; g.N = k1; g.K = m;
            g.A = W; g.lda = ld; g.a_kmajor = false;
            g.B = W; g.ldb = ld; g.b_kmajor = false;
            g.Cin = Kinv; g.ldcin = ld; g.D = Kinv; g.ldd = ld;
            g.alpha = 1.0; g.beta = 1.0; g.lower = true; g.prof_cls = FR_PROF_GEMM_SOLVE;
            FR_TRY(launch_gemm(ctx, g));
        }
    } else {
        // K8: W = L^-1 (strict upper triangle of W is exactly zero), Kinv = W^T W (lower triangle)
        FR_TRY(chol_tri_inverse(ctx, c, W, ld, Kinv, FR_PROF_GEMM_SOLVE));  // (Kinv's buffer is the scratch: it is written next)
        GemmDesc g;
        const bool tri = ctx->tri_inverse != 0 && n > 2048;  // (small cases: few tiles, cut along K instead -- launch_gemm)
        g.dynamic = tri;  // the tiles' contractions differ in length: claimed, not dealt
        g.tri = tri ? 1 : 0;  // W^T (m, k) = W (k, m) is zero for k < m: a lower tile's contraction starts at its row offset
        g.M = n; g.N = n; g.K = n;
        g.A = W; g.lda = ld; g.a_kmajor = true;
        g.B = W; g.ldb = ld; g.b_kmajor = true;
        g.Cin = Kinv; g.ldcin = ld; g.D = Kinv; g.ldd = ld;
        g.alpha = 1.0; g.beta = 0.0; g.lower = true; g.prof_cls = FR_PROF_GEMM_SOLVE;
        FR_TRY(launch_gemm(ctx, g));
    }
    // K2: fused reductions over the lower triangle
    GradArgs a;
    a.prog = *kernel;
    a.X = c->X; a.n = n; a.ldx = c->ld_x; a.d = c->d;
    a.alpha = alpha; a.Kinv = Kinv; a.ldk = ld;
    a.ya = outs + ng + 2;
    a.nd = (double)n;
    a.scaled = scaled ? 1 : 0;
    a.alpha_w = (sharded && me != 0) ? 0.0 : 1.0;
    a.ng = ng;
    a.partials = partials;
    {
        ProfScope ps(ctx, FR_PROF_GRAM, 0.5 * (double)n * (double)n * (3.0 * (double)c->d + 60.0), 4.0 * (double)n * (double)n);
        const int needs = kprog_needs(*kernel);
        const int leaf = kernel->nops == 1 ? kernel->ops[0].kind : -1;
        const dim3 grid((unsigned)nblocks), block(256);
#define FR_GRAD_LEAF(KIND, MODE_)                                                                   \
    case KIND: hipLaunchKernelGGL((grad_reduce_kernel<MODE_, KIND>), grid, block, 0, ctx->ls, a); break;
        switch (leaf) {
            FR_GRAD_LEAF(FR_K_LINEAR, NEED_U)
            FR_GRAD_LEAF(FR_K_POLYNOMIAL, NEED_U)
            FR_GRAD_LEAF(FR_K_SQUAREDEXP, NEED_S)
            FR_GRAD_LEAF(FR_K_EXPONENTIAL, NEED_S)
            FR_GRAD_LEAF(FR_K_MATERN1, NEED_S)
            FR_GRAD_LEAF(FR_K_MATERN2, NEED_S)
            FR_GRAD_LEAF(FR_K_HYPERTAN, NEED_U)
            FR_GRAD_LEAF(FR_K_MULTIQUADRIC, NEED_S)
            FR_GRAD_LEAF(FR_K_RATIONALQUADRATIC, NEED_S)
        default:
            if (needs == NEED_S)
                hipLaunchKernelGGL((grad_reduce_kernel<NEED_S, -1>), grid, block, 0, ctx->ls, a);
            else if (needs == NEED_U)
                hipLaunchKernelGGL((grad_reduce_kernel<NEED_U, -1>), grid, block, 0, ctx->ls, a);
            else
                hipLaunchKernelGGL((grad_reduce_kernel<NEED_S | NEED_U, -1>), grid, block, 0, ctx->ls, a);
        }
#undef FR_GRAD_LEAF
        FR_HIP(ctx, hipGetLastError());
        hipLaunchKernelGGL(grad_finish_kernel, dim3(1), dim3(256), 0, ctx->ls, (const double*)partials, nblocks, ng,
                           (const double*)Kinv, ld, (const double*)alpha, n, outs);
        FR_HIP(ctx, hipGetLastError());
    }
    double h[MAXG + 3];
    double h_ya = 0.0;
    if (sharded) {
        // every rank's partials, summed in rank order (the same bits on every rank); alpha . alpha is rank 0's
        if (Wn > 64) return set_err(ctx, FR_INVALID_ARGUMENT, "more than 64 ranks");
        const size_t cnt = (size_t)(ng + 2);
        double* all = outs + (MAXG + 8);
        FR_TRY(comm_allgather(ctx, outs, all, cnt));
        std::vector<double> hh(cnt * (size_t)Wn);
        FR_TRY(comm_d2h(ctx, hh.data(), all, sizeof(double) * hh.size(), ctx->stream, "the all-gather of the gradient partials"));
        FR_TRY(comm_stream_sync(ctx, ctx->stream, "the all-gather of the gradient partials"));
        for (size_t q = 0; q < cnt; ++q) {
            h[q] = 0.0;
            for (int r = 0; r < Wn; ++r) h[q] += hh[(size_t)r * cnt + q];
        }
        h[ng + 1] = hh[(size_t)ng + 1];
        FR_HIP(ctx, hipMemcpy(&h_ya, outs + ng + 2, sizeof(double), hipMemcpyDeviceToHost));
    } else {
        FR_HIP(ctx, hipMemcpyAsync(h, outs, sizeof(double) * (size_t)(ng + 3), hipMemcpyDeviceToHost, ctx->stream));
        FR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        h_ya = h[ng + 2];
    }
    FR_TRY(check_status_word(ctx));  // (the alpha solves are persistent kernels)
    const double scale = h_ya / (double)n;
    // one entry per DECLARED parameter: the reference allocates nb_parameters() matrices and zips them POSITIONALLY with
    // the gradient vector (algebra/mod.rs:135-151), so trailing entries without a value stay NaN (Multiquadric)
    for (int q = 0; q < np; ++q) out_grad[q] = (q < ng) ? h[q] : std::nan("");
    if (!scaled) out_grad[np] = noise * (h[ng + 1] - h[ng]);  // optimizer.rs:54-57
    if (out_scale) *out_scale = scale;
    return FR_OK;
}
