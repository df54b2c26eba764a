// gp.hip -- the predict family of src/gaussian_process/mod.rs fused on the device: the n x m cross-Gram K*
// is produced in HBM by K1 and never leaves it; the solves are the GEMM-recast TRSMs of chol.hip; only the
// m-vectors (or the m x m posterior) travel back.
//
//   likelihood              mod.rs:196-220
//   predict                 mod.rs:226-244
//   predict_variance        mod.rs:248-273
//   predict_mean_variance   mod.rs:290-326
//   predict_covariance      mod.rs:329-350
//   sample_at               mod.rs:371-392 + MultivariateNormal::new (multivariate_normal.rs:54-59)
#include "fr_internal.hpp"

namespace fr {

int potrf_matrix_ws(fr_ctx* ctx, double* A, int64_t ld, int64_t n, int mode, double sub, int64_t* fail_col);

struct QueryCtx {
    fr_chol* c;
    fr_ctx* ctx;
    Staged xq;
    WsGuard kstar;
    double* K = nullptr;  // n x m cross-covariance
    int64_t ldk = 0;
    int64_t m = 0;
    QueryCtx(fr_chol* c_) : c(c_), ctx(c_->ctx), xq(c_->ctx), kstar(c_->ctx) {}
    // make_covariance_matrix(train, inputs) (mod.rs:234, 256-257, 296-297, 337-338, 377-378)
    int init(const fr_kprog* kernel, const double* Xq, int64_t m_, int64_t ldq, bool with_gram = true)
    {
        FR_HIP(ctx, hipSetDevice(ctx->device));
        FR_TRY(kprog_check(ctx, kernel));
        if (m_ < 0) return set_err(ctx, FR_SHAPE, "negative query count");
        m = m_;
        FR_TRY(xq.in(Xq, m, c->d, ldq));
        ldk = round_up(c->n > 0 ? c->n : 1, kAlign);
        if (!with_gram) return FR_OK;  // (the caller fuses the covariance into its product: launch_gram_cross_dot)
        K = kstar.get(sizeof(double) * (size_t)ldk * (size_t)(m > 0 ? m : 1));
        if (!K) return FR_OUT_OF_MEMORY;
        return launch_gram_cross(ctx, *kernel, c->X, c->n, c->ld_x, xq.dev, m, xq.ld, c->d, K, ldk);
    }
};

static int stage_vec_in(fr_ctx* ctx, Staged& s, const double* v, int64_t n)
{
    if (n > 0 && !v) return set_err(ctx, FR_INVALID_ARGUMENT, "null training-output vector (y) for a factor of %lld rows", (long long)n);
    return s.in(v, n, 1, n > 0 ? n : 1);
}
static int stage_vec_out(fr_ctx* ctx, Staged& s, double* v, int64_t n) { return s.out(v, n, 1, n > 0 ? n : 1); }

// out = prior_q (or zeros)
static int init_with_prior(fr_ctx* ctx, double* out_dev, const double* prior_q, int64_t m)
{
    if (m <= 0) return FR_OK;
    if (!prior_q) return launch_fill(ctx, out_dev, m, 1, m, 0.0);
    const bool dev = is_device_ptr(prior_q);
    FR_HIP(ctx, hipMemcpyAsync(out_dev, prior_q, sizeof(double) * (size_t)m,
                               dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
    if (!dev) FR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FR_OK;
}

static int zero_diag_status(fr_chol* c, const char* what)
{
    fr_ctx* ctx = c->ctx;
    if (c->diag_gen != c->gen) {
        FR_HIP(ctx, hipMemsetAsync(c->info + 2, 0, sizeof(int64_t), ctx->stream));
        FR_TRY(launch_diag_check_zero(ctx, c->A, c->n, c->ld_a, c->info + 2));
        int64_t flag = 0;
        FR_HIP(ctx, hipMemcpyAsync(&flag, c->info + 2, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
        FR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        c->diag_zero = flag != 0;
        c->diag_gen = c->gen;
    }
    if (c->diag_zero) return set_err(ctx, FR_SINGULAR_SOLVE, "%s : solve failed", what);
    return FR_OK;
}

// alpha = K^-1 targets, recomputed when the factor changed since it was last solved (two single-column solves: trsv.hip)
static int ensure_alpha(fr_chol* c)
{
    fr_ctx* ctx = c->ctx;
    if (c->targets_n != c->n || !c->yt)
        return set_err(ctx, FR_INVALID_ARGUMENT,
                       "no training outputs cached for the current %lld rows: call fr_chol_set_targets after the factor's row "
                       "count changed (or pass y)", (long long)c->n);
    if (c->alpha_gen == c->gen) return FR_OK;
    if (c->n > 0) {
        FR_HIP(ctx, hipMemcpyAsync(c->alpha, c->yt, sizeof(double) * (size_t)c->n, hipMemcpyDeviceToDevice, ctx->stream));
        const int64_t ld = c->targets_cap;
        FR_TRY(trsm_lower_fwd(ctx, c, c->n, c->alpha, 1, ld, FR_PROF_GEMM_SOLVE));
        FR_TRY(trsm_lower_bwd(ctx, c, c->n, c->alpha, 1, ld, FR_PROF_GEMM_SOLVE));
        // alpha is marked current only once the solves are known to have completed (a timed-out hand-off would otherwise
        // leave garbage behind a valid generation tag); once per change of the factor, so the synchronisation is free
        FR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        FR_TRY(check_status_word(ctx));
    }
    c->alpha_gen = c->gen;
    return FR_OK;
}

}  // namespace fr

using namespace fr;

static int fr_likelihood_impl(fr_chol* c, const fr_kprog* kernel, const double* y, double noise, double* out)
{
    if (!c || !out) return FR_INVALID_ARGUMENT;
    FR_LOCK(c->ctx);
    fr_ctx* ctx = c->ctx;
    FR_HIP(ctx, hipSetDevice(ctx->device));
    FR_TRY(kprog_check(ctx, kernel));
    const int64_t n = c->n;
    if (n > 0 && !y) return set_err(ctx, FR_INVALID_ARGUMENT, "null training-output vector (y)");
    FR_TRY(zero_diag_status(c, "likelihood"));  // mod.rs:203
    WsGuard w(ctx);
    const int64_t ld = round_up(n > 0 ? n : 1, kAlign);
    double* buf = w.get(sizeof(double) * (size_t)(2 * ld + 8));
    if (!buf) return FR_OUT_OF_MEMORY;
    double* ol = buf;
    double* diag = buf + ld;
    double* scal = buf + 2 * ld;  // [0] data_fit, [1] complexity
    if (n > 0) {
        const bool dev = is_device_ptr(y);
        FR_HIP(ctx, hipMemcpyAsync(ol, y, sizeof(double) * (size_t)n, dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                                   ctx->stream));
        if (!dev) FR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    FR_TRY(trsm_lower_fwd(ctx, c, n, ol, 1, ld, FR_PROF_GEMM_SOLVE));               // ol = L^-1 y          :203
    FR_HIP(ctx, hipMemsetAsync(scal, 0, sizeof(double) * 2, ctx->stream));
    if (n > 0) {
        FR_TRY(launch_col_norm2(ctx, ol, n, 1, ld, scal));                           // data_fit             :204
        FR_TRY(launch_gram_diag(ctx, *kernel, c->X, n, c->ld_x, c->d, noise * noise, diag));  // k(r,r)+noise^2  :211
        FR_TRY(launch_sum_log_abs(ctx, diag, n, scal + 1));                          // sum ln|.|            :212-213
    }
    double h[2] = {0.0, 0.0};
    FR_HIP(ctx, hipMemcpyAsync(h, scal, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    FR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    FR_TRY(check_status_word(ctx));
    const double normalization_constant = (double)n * std::log(2.0 * M_PI);  // :217
    *out = -(h[0] + h[1] + normalization_constant) / 2.0;                    // :219
    return FR_OK;
}

static int fr_predict_mean_impl(fr_chol* c, const fr_kprog* kernel, const double* y, const double* Xq, int64_t m, int64_t ldq,
                    const double* prior_q, double* out_mean)
{
    if (!c) return FR_INVALID_ARGUMENT;
    FR_LOCK(c->ctx);
    fr_ctx* ctx = c->ctx;
    QueryCtx q(c);
    FR_TRY(q.init(kernel, Xq, m, ldq, y != nullptr));
    Staged ys(ctx), mean(ctx);
    if (!y) {
        // cached alpha = K^-1 y (fr_chol_set_targets): prior + K*^T alpha -- no solve at all once alpha is current, and the
        // n x m cross-covariance is multiplied by alpha tile by tile as K1 produces it: it never reaches memory (todo.md:10)
        FR_TRY(ensure_alpha(c));
        FR_TRY(stage_vec_out(ctx, mean, out_mean, m));
        FR_TRY(init_with_prior(ctx, mean.dev, prior_q, m));
        FR_TRY(launch_gram_cross_dot(ctx, *kernel, c->X, c->n, c->ld_x, q.xq.dev, m, q.xq.ld, c->d, c->alpha, mean.dev));
        return mean.commit();
    }
    FR_TRY(stage_vec_in(ctx, ys, y, c->n));
    FR_TRY(stage_vec_out(ctx, mean, out_mean, m));
    FR_TRY(init_with_prior(ctx, mean.dev, prior_q, m));
    if (ctx->predict_assoc == 1) {
        // same value associated the cheap way: prior + K*^T (K^-1 y) -- two n x 1 solves instead of two n x m ones
        WsGuard w(ctx);
        const int64_t lda = round_up(c->n > 0 ? c->n : 1, kAlign);
        double* alpha = w.get(sizeof(double) * (size_t)lda);
        if (!alpha) return FR_OUT_OF_MEMORY;
        FR_TRY(launch_copy(ctx, ys.dev, lda, alpha, lda, c->n, 1));
        FR_TRY(trsm_lower_fwd(ctx, c, c->n, alpha, 1, lda, FR_PROF_GEMM_SOLVE));
        FR_TRY(trsm_lower_bwd(ctx, c, c->n, alpha, 1, lda, FR_PROF_GEMM_SOLVE));
        FR_TRY(launch_gemv_t(ctx, q.K, c->n, m, q.ldk, alpha, 1.0, 1.0, mean.dev));
        return mean.commit();
    }
    // weights = K^-1 K*   (solve_mut, mod.rs:235)
    FR_TRY(trsm_lower_fwd(ctx, c, c->n, q.K, m, q.ldk, FR_PROF_GEMM_SOLVE));
    FR_TRY(trsm_lower_bwd(ctx, c, c->n, q.K, m, q.ldk, FR_PROF_GEMM_SOLVE));
    // prior.gemm_tr(1, weights, y, 1)   (mod.rs:238-241)
    FR_TRY(launch_gemv_t(ctx, q.K, c->n, m, q.ldk, ys.dev, 1.0, 1.0, mean.dev));
    return mean.commit();
}

static int fr_predict_variance_impl(fr_chol* c, const fr_kprog* kernel, const double* Xq, int64_t m, int64_t ldq, double* out_var)
{
    if (!c) return FR_INVALID_ARGUMENT;
    FR_LOCK(c->ctx);
    fr_ctx* ctx = c->ctx;
    QueryCtx q(c);
    FR_TRY(q.init(kernel, Xq, m, ldq));
    FR_TRY(zero_diag_status(c, "predict_covariance"));  // mod.rs:263
    Staged var(ctx);
    FR_TRY(stage_vec_out(ctx, var, out_var, m));
    // kl = L^-1 K*  (mod.rs:260-263); var_i = k(x_i, x_i) - ||kl[:, i]||^2  (:266-270): the solve, then ONE epilogue launch
    FR_TRY(trsm_lower_fwd(ctx, c, c->n, q.K, m, q.ldk, FR_PROF_GEMM_SOLVE));
    FR_TRY(launch_variance_epilogue(ctx, *kernel, q.xq.dev, m, q.xq.ld, c->d, q.K, q.ldk, q.K, q.ldk, c->n, var.dev));
    return var.commit();
}

static int fr_predict_mean_variance_impl(fr_chol* c, const fr_kprog* kernel, const double* y, const double* Xq, int64_t m,
                             int64_t ldq, const double* prior_q, double* out_mean, double* out_var)
{
    if (!c) return FR_INVALID_ARGUMENT;
    FR_LOCK(c->ctx);
    fr_ctx* ctx = c->ctx;
    QueryCtx q(c);
    FR_TRY(q.init(kernel, Xq, m, ldq));
    Staged ys(ctx), mean(ctx), var(ctx);
    FR_TRY(stage_vec_in(ctx, ys, y, c->n));
    FR_TRY(stage_vec_out(ctx, mean, out_mean, m));
    FR_TRY(stage_vec_out(ctx, var, out_var, m));
    WsGuard w(ctx);
    double* W = w.get(sizeof(double) * (size_t)q.ldk * (size_t)(m > 0 ? m : 1));
    if (!W) return FR_OUT_OF_MEMORY;
    // weights = covmat_cholesky.solve(&cov_train_inputs)  (clone + solve_mut, mod.rs:298)
    FR_TRY(launch_copy(ctx, q.K, q.ldk, W, q.ldk, c->n, m));
    FR_TRY(trsm_lower_fwd(ctx, c, c->n, W, m, q.ldk, FR_PROF_GEMM_SOLVE));
    FR_TRY(trsm_lower_bwd(ctx, c, c->n, W, m, q.ldk, FR_PROF_GEMM_SOLVE));
    FR_TRY(init_with_prior(ctx, mean.dev, prior_q, m));
    FR_TRY(launch_gemv_t(ctx, W, c->n, m, q.ldk, ys.dev, 1.0, 1.0, mean.dev));  // :306
    // var_i = k(x_i,x_i) - K*[:,i] . W[:,i]   (:313-319)
    FR_TRY(launch_variance_epilogue(ctx, *kernel, q.xq.dev, m, q.xq.ld, c->d, q.K, q.ldk, W, q.ldk, c->n, var.dev));
    FR_TRY(mean.commit());
    return var.commit();
}

static int fr_predict_covariance_impl(fr_chol* c, const fr_kprog* kernel, const double* Xq, int64_t m, int64_t ldq, double* out_cov,
                          int64_t ldc)
{
    if (!c) return FR_INVALID_ARGUMENT;
    FR_LOCK(c->ctx);
    fr_ctx* ctx = c->ctx;
    QueryCtx q(c);
    FR_TRY(q.init(kernel, Xq, m, ldq));
    FR_TRY(zero_diag_status(c, "predict_covariance"));  // mod.rs:345
    Staged cov(ctx);
    FR_TRY(cov.out(out_cov, m, m, ldc));
    // cov_inputs_inputs (mod.rs:339), kl = L^-1 K* (:342-345), cov -= kl^T kl (:348)
    FR_TRY(launch_gram_cross(ctx, *kernel, q.xq.dev, m, q.xq.ld, q.xq.dev, m, q.xq.ld, c->d, cov.dev, cov.ld));
    FR_TRY(trsm_lower_fwd(ctx, c, c->n, q.K, m, q.ldk, FR_PROF_GEMM_SOLVE));
    GemmDesc g;
    g.M = m; g.N = m; g.K = c->n;
    g.A = q.K; g.lda = q.ldk; g.a_kmajor = true;
    g.B = q.K; g.ldb = q.ldk; g.b_kmajor = true;
    g.Cin = cov.dev; g.ldcin = cov.ld; g.D = cov.dev; g.ldd = cov.ld;
    g.alpha = -1.0; g.beta = 1.0; g.lower = false; g.prof_cls = FR_PROF_GEMM_SOLVE;
    FR_TRY(launch_gemm(ctx, g));
    return cov.commit();
}

static int fr_posterior_impl(fr_chol* c, const fr_kprog* kernel, const double* y, const double* Xq, int64_t m, int64_t ldq,
                 const double* prior_q, double* out_mean, double* out_cov, int64_t ldc, double* out_cov_l, int64_t ldl)
{
    if (!c) return FR_INVALID_ARGUMENT;
    FR_LOCK(c->ctx);
    fr_ctx* ctx = c->ctx;
    QueryCtx q(c);
    FR_TRY(q.init(kernel, Xq, m, ldq));
    Staged ys(ctx), mean(ctx), covl(ctx);
    FR_TRY(stage_vec_in(ctx, ys, y, c->n));
    FR_TRY(stage_vec_out(ctx, mean, out_mean, m));
    FR_TRY(covl.out(out_cov_l, m, m, ldl));
    WsGuard w(ctx);
    double* W = w.get(sizeof(double) * (size_t)q.ldk * (size_t)(m > 0 ? m : 1));
    if (!W) return FR_OUT_OF_MEMORY;
    FR_TRY(launch_copy(ctx, q.K, q.ldk, W, q.ldk, c->n, m));  // weights = solve(K*)  mod.rs:379
    FR_TRY(trsm_lower_fwd(ctx, c, c->n, W, m, q.ldk, FR_PROF_GEMM_SOLVE));
    FR_TRY(trsm_lower_bwd(ctx, c, c->n, W, m, q.ldk, FR_PROF_GEMM_SOLVE));
    // cov = K** - K*^T W   (mod.rs:382-383), assembled directly in the buffer that will be factored
    FR_TRY(launch_gram_cross(ctx, *kernel, q.xq.dev, m, q.xq.ld, q.xq.dev, m, q.xq.ld, c->d, covl.dev, covl.ld));
    GemmDesc g;
    g.M = m; g.N = m; g.K = c->n;
    g.A = q.K; g.lda = q.ldk; g.a_kmajor = true;
    g.B = W; g.ldb = q.ldk; g.b_kmajor = true;
    g.Cin = covl.dev; g.ldcin = covl.ld; g.D = covl.dev; g.ldd = covl.ld;
    g.alpha = -1.0; g.beta = 1.0; g.lower = false; g.prof_cls = FR_PROF_GEMM_SOLVE;
    FR_TRY(launch_gemm(ctx, g));
    // mean = prior + W^T y   (mod.rs:387-388)
    FR_TRY(init_with_prior(ctx, mean.dev, prior_q, m));
    FR_TRY(launch_gemv_t(ctx, W, c->n, m, q.ldk, ys.dev, 1.0, 1.0, mean.dev));
    // (the covariance goes to its own device image here and travels to the host at the END, with the mean and the factor: a
    // commit in this place -- a device-to-host copy and a synchronisation -- left the GPU idle for ~110 us in front of the
    // factorisation of every sample_at(256), round 5's kernel trace of configs[4])
    Staged cov(ctx);
    if (out_cov) {
        FR_TRY(cov.out(out_cov, m, m, ldc));
        FR_TRY(launch_copy(ctx, covl.dev, covl.ld, cov.dev, cov.ld, m, m));
    }
    // MultivariateNormal::new: covariance.cholesky().expect(..).unpack()   (multivariate_normal.rs:56-57)
    int64_t fail_col = -1;
    ctx->refine_now = c->refine;  // an ill-conditioned training factor usually means an ill-conditioned posterior
    const int pst = potrf_matrix_ws(ctx, covl.dev, covl.ld, m, 0, 0.0, &fail_col);
    ctx->refine_now = false;
    FR_TRY(pst);
    FR_TRY(launch_tri_fill(ctx, covl.dev, m, covl.ld, 0.0));
    FR_TRY(mean.commit());
    if (out_cov) FR_TRY(cov.commit());
    FR_TRY(covl.commit());
    if (fail_col >= 0)
        return set_err(ctx, FR_NOT_POSITIVE_DEFINITE, "MultivariateNormal: Cholesky decomposition failed! (column %lld)",
                       (long long)fail_col);
    return FR_OK;
}

extern "C" {

int fr_chol_set_targets(fr_chol* c, const double* y)
{
    if (!c) return FR_INVALID_ARGUMENT;
    FR_LOCK(c->ctx);
    fr_ctx* ctx = c->ctx;
    FR_HIP(ctx, hipSetDevice(ctx->device));
    if (c->n > 0 && !y) return set_err(ctx, FR_INVALID_ARGUMENT, "null training-output vector (y)");
    if (c->targets_cap < c->capacity || !c->yt) {
        FR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (c->yt) (void)hipFree(c->yt);
        if (c->alpha) (void)hipFree(c->alpha);
        c->yt = c->alpha = nullptr;
        c->targets_cap = 0;
        const int64_t cap = round_up(c->capacity > 0 ? c->capacity : 1, kAlign);
        FR_HIP(ctx, dev_malloc(ctx, (void**)&c->yt, sizeof(double) * (size_t)cap));
        FR_HIP(ctx, dev_malloc(ctx, (void**)&c->alpha, sizeof(double) * (size_t)cap));
        c->targets_cap = cap;
    }
    if (c->n > 0) {
        const bool dev = is_device_ptr(y);
        FR_HIP(ctx, hipMemcpyAsync(c->yt, y, sizeof(double) * (size_t)c->n, dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                                   ctx->stream));
        if (!dev) FR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    c->targets_n = c->n;
    c->alpha_gen = 0;  // solved lazily by the first predict
    return FR_OK;
}

int fr_likelihood(fr_chol* c, const fr_kprog* kernel, const double* y, double noise, double* out)
{
    if (!c) return FR_INVALID_ARGUMENT;
    FR_LOCK(c->ctx);
    return solve_retry(c->ctx, [&]() -> int { return fr_likelihood_impl(c, kernel, y, noise, out); });
}

int fr_predict_mean(fr_chol* c, const fr_kprog* kernel, const double* y, const double* Xq, int64_t m, int64_t ldq,
                    const double* prior_q, double* out_mean)
{
    if (!c) return FR_INVALID_ARGUMENT;
    FR_LOCK(c->ctx);
    return solve_retry(c->ctx, [&]() -> int { return fr_predict_mean_impl(c, kernel, y, Xq, m, ldq, prior_q, out_mean); });
}

int fr_predict_variance(fr_chol* c, const fr_kprog* kernel, const double* Xq, int64_t m, int64_t ldq, double* out_var)
{
    if (!c) return FR_INVALID_ARGUMENT;
    FR_LOCK(c->ctx);
    return solve_retry(c->ctx, [&]() -> int { return fr_predict_variance_impl(c, kernel, Xq, m, ldq, out_var); });
}

int fr_predict_mean_variance(fr_chol* c, const fr_kprog* kernel, const double* y, const double* Xq, int64_t m,
                             int64_t ldq, const double* prior_q, double* out_mean, double* out_var)
{
    if (!c) return FR_INVALID_ARGUMENT;
    FR_LOCK(c->ctx);
    return solve_retry(c->ctx, [&]() -> int { return fr_predict_mean_variance_impl(c, kernel, y, Xq, m, ldq, prior_q, out_mean, out_var); });
}

int fr_predict_covariance(fr_chol* c, const fr_kprog* kernel, const double* Xq, int64_t m, int64_t ldq, double* out_cov,
                          int64_t ldc)
{
    if (!c) return FR_INVALID_ARGUMENT;
    FR_LOCK(c->ctx);
    return solve_retry(c->ctx, [&]() -> int { return fr_predict_covariance_impl(c, kernel, Xq, m, ldq, out_cov, ldc); });
}

int fr_posterior(fr_chol* c, const fr_kprog* kernel, const double* y, const double* Xq, int64_t m, int64_t ldq,
                 const double* prior_q, double* out_mean, double* out_cov, int64_t ldc, double* out_cov_l, int64_t ldl)
{
    if (!c) return FR_INVALID_ARGUMENT;
    FR_LOCK(c->ctx);
    return solve_retry(c->ctx, [&]() -> int { return fr_posterior_impl(c, kernel, y, Xq, m, ldq, prior_q, out_mean, out_cov, ldc, out_cov_l, ldl); });
}

int fr_gemm(fr_ctx* ctx, int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, double alpha, const double* A,
            int64_t lda, const double* B, int64_t ldb, double beta, double* C, int64_t ldc)
{
    if (!ctx) return FR_INVALID_ARGUMENT;
    FR_LOCK(ctx);
    FR_HIP(ctx, hipSetDevice(ctx->device));
    if (M < 0 || N < 0 || K < 0) return set_err(ctx, FR_SHAPE, "negative GEMM dimension");
    Staged a(ctx), b(ctx), c(ctx);
    FR_TRY(trans_a ? a.in(A, K, M, lda) : a.in(A, M, K, lda));
    FR_TRY(trans_b ? b.in(B, N, K, ldb) : b.in(B, K, N, ldb));
    if (beta != 0.0)
        FR_TRY(c.inout(C, M, N, ldc));
    else
        FR_TRY(c.out(C, M, N, ldc));
    GemmDesc g;
    g.M = M; g.N = N; g.K = K;
    g.A = a.dev; g.lda = a.ld; g.a_kmajor = trans_a != 0;   // stored K x M: element (m,k) at A[k + m*lda]
    g.B = b.dev; g.ldb = b.ld; g.b_kmajor = trans_b == 0;   // stored K x N: element (k,n) at B[k + n*ldb]
    g.Cin = c.dev; g.ldcin = c.ld; g.D = c.dev; g.ldd = c.ld;
    g.alpha = alpha; g.beta = beta; g.prof_cls = FR_PROF_GEMM_SOLVE;
    g.lower = false;
    FR_TRY(launch_gemm(ctx, g));
    return c.commit();
}

}  // extern "C"
