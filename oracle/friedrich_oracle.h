/*
 * friedrich_oracle.h -- CPU restatement of friedrich 0.5.1's dense-linear-algebra hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker / the reported CPU baseline.  The product path (friedrich_amd/) never
 * links, imports or calls it.
 *
 * PARITY UNPINNED: the reference (Rust, nalgebra 0.31.4) cannot be built in this image (no
 * rustc/cargo, nalgebra source not vendored) and its test-suite holds no numeric assertion,
 * golden vector or fixture for any function on this path.  This
 * restatement follows the reference sources statement by statement (file:line cited at each
 * function) and nalgebra 0.31.4's published algorithms from memory; it is cross-checked against
 * scipy/LAPACK and 50-digit mpmath in tests/test_oracle_*.py, not against reference outputs.
 *
 * All matrices are column-major f64 with an explicit leading dimension (nalgebra DMatrix layout;
 * EMatrix::as_matrix() yields ld = capacity != nrows, src/algebra/extendable_matrix.rs:52-55).
 * Compile with -ffp-contract=off: nalgebra/rustc never fuse mul+add.
 */
#ifndef FRIEDRICH_ORACLE_H
#define FRIEDRICH_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Kernel "program": reverse-polish list of leaves and Sum/Prod combinators.
 * Same POD layout as fr_kprog in include/friedrich_amd.h (kept in sync by tests/test_abi.py). */
enum {
    FRO_K_LINEAR = 0,            /* src/parameters/kernel.rs:342-402  params [c]            */
    FRO_K_POLYNOMIAL = 1,        /* :411-485  params [alpha, c, d]                            */
    FRO_K_SQUAREDEXP = 2,        /* :496-601  params [ls, ampl]   (alias Gaussian :496)       */
    FRO_K_EXPONENTIAL = 3,       /* :612-706  params [ls, ampl]                               */
    FRO_K_MATERN1 = 4,           /* :717-813  params [ls, ampl]                               */
    FRO_K_MATERN2 = 5,           /* :824-925  params [ls, ampl]                               */
    FRO_K_HYPERTAN = 6,          /* :934-1001 params [alpha, c]                               */
    FRO_K_MULTIQUADRIC = 7,      /* :1010-1070 params [c]                                     */
    FRO_K_RATIONALQUADRATIC = 8, /* :1079-1157 params [alpha, ls]                             */
    FRO_K_SUM = 100,             /* :132-211  pops two, pushes k1 + k2                        */
    FRO_K_PROD = 101             /* :221-307  pops two, pushes k1 * k2                        */
};

#define FRO_KPROG_MAX_OPS 15

typedef struct {
    int32_t kind;
    int32_t nparams;
    double params[3];
} fro_kernel_op;

typedef struct {
    int32_t nops;
    int32_t reserved;
    fro_kernel_op ops[FRO_KPROG_MAX_OPS];
} fro_kprog;

/* ---- per-pair kernel math (src/parameters/kernel.rs) ---- */
int fro_kprog_validate(const fro_kprog* p);
int fro_kprog_nb_parameters(const fro_kprog* p); /* Kernel::nb_parameters (Multiquadric says 2, :1041) */
int fro_kprog_nb_gradients(const fro_kprog* p);  /* length of Kernel::gradient's Vec                   */
int fro_kprog_is_scalable(const fro_kprog* p);   /* :33-36,152,241,544...                              */
int fro_kprog_get_parameters(const fro_kprog* p, double* out);
int fro_kprog_set_parameters(fro_kprog* p, const double* params, int n);
int fro_kprog_rescale(fro_kprog* p, double scale);
double fro_kernel(const fro_kprog* p, const double* x1, int64_t inc1, const double* x2, int64_t inc2, int64_t d);
int fro_kernel_gradient(const fro_kprog* p, const double* x1, int64_t inc1, const double* x2, int64_t inc2,
                        int64_t d, double* out);

/* heuristics kernel.rs:94-119 */
double fro_fit_bandwidth_mean(const double* X, int64_t n, int64_t ldx, int64_t d);
double fro_variance(const double* y, int64_t n); /* nalgebra variance(): population, E[x^2]-E[x]^2 */
double fro_mean(const double* y, int64_t n);
void fro_heuristic_fit(fro_kprog* p, const double* X, int64_t n, int64_t ldx, int64_t d, const double* y);

/* ---- src/algebra/mod.rs ---- */
/* :41-54 */
void fro_make_covariance_matrix(const fro_kprog* p, const double* A, int64_t n1, int64_t lda, const double* B,
                                int64_t n2, int64_t ldb, int64_t d, double* out, int64_t ldo);
/* :59-92.  out: n x n, upper triangle left NaN (:67).  Returns 0, or 1+j when column j's pivot
 * could not be taken (the reference panics, :85/:90).  subst_idx (may be NULL) receives the ordered
 * list of columns where the substitute fired; *n_subst their count. */
int fro_make_cholesky_cov_matrix(const fro_kprog* p, const double* X, int64_t n, int64_t ldx, int64_t d,
                                 double noise, int has_eps, double eps, double* out, int64_t ldo,
                                 int64_t* n_subst, int64_t* subst_idx);
/* nalgebra Cholesky::new_internal, in place on the lower triangle. */
int fro_cholesky(double* A, int64_t n, int64_t lda, int has_sub, double sub, int64_t* n_subst,
                 int64_t* subst_idx);
/* Host threads for the loops whose iterations are independent in the reference (right-hand sides of a solve, Gram
 * entries, rows of an axpy).  n <= 0: all online cores.  Default 1.  Results are bit-identical for every thread count. */
void fro_set_threads(int n);
int fro_get_threads(void);
/* fro_cholesky / fro_make_cholesky_cov_matrix restricted to the leading ncols columns (out is n x ncols; the strict upper
 * part of those columns is NaN), blocked and threaded with the reference's per-element operation order: bit-identical
 * to the leading columns of the single-thread functions. */
int fro_cholesky_cols_mt(double* A, int64_t n, int64_t lda, int64_t ncols, int has_sub, double sub, int64_t* n_subst,
                         int64_t* subst_idx);
int fro_make_cholesky_cov_matrix_cols_mt(const fro_kprog* p, const double* X, int64_t n, int64_t ldx, int64_t d,
                                         double noise, int has_eps, double eps, int64_t ncols, double* out, int64_t ldo,
                                         int64_t* n_subst, int64_t* subst_idx);
/* :97-126 + Cholesky::insert_column (A.3).  L: buffer with ldl >= n_old+nb_new holding the n_old factor;
 * on return holds the (n_old+nb_new) factor.  No epsilon, no failure check (plain sqrt => NaN). */
void fro_add_rows_cholesky_cov_matrix(const fro_kprog* p, double* L, int64_t ldl, const double* Xall,
                                      int64_t n_all, int64_t ldx, int64_t d, int64_t nb_new, double noise);
/* :129-155.  out: nb_parameters() matrices of n x n (ld n), contiguous; both triangles written. */
void fro_make_gradient_covariance_matrices(const fro_kprog* p, const double* X, int64_t n, int64_t ldx,
                                           int64_t d, double* out);

/* ---- nalgebra triangular solves (Appendix A.2) ---- */
int fro_solve_lower(const double* L, int64_t n, int64_t ldl, double* B, int64_t m, int64_t ldb);    /* checked: -1 on zero diag */
void fro_ad_solve_lower(const double* L, int64_t n, int64_t ldl, double* B, int64_t m, int64_t ldb); /* L^T x = b, dot-oriented   */
void fro_chol_solve(const double* L, int64_t n, int64_t ldl, double* B, int64_t m, int64_t ldb);     /* solve_mut                */
void fro_chol_inverse(const double* L, int64_t n, int64_t ldl, double* out, int64_t ldo);            /* identity + solve_mut     */

/* ---- src/gaussian_process/mod.rs (prior values are supplied by the caller) ---- */
double fro_likelihood(const fro_kprog* p, const double* L, int64_t n, int64_t ldl, const double* X, int64_t ldx,
                      int64_t d, const double* y, double noise);                                   /* :196-220 */
void fro_predict(const fro_kprog* p, const double* L, int64_t n, int64_t ldl, const double* X, int64_t ldx,
                 int64_t d, const double* y, const double* Xq, int64_t m, int64_t ldq, const double* prior_q,
                 double* mean);                                                                    /* :226-244 */
int fro_predict_variance(const fro_kprog* p, const double* L, int64_t n, int64_t ldl, const double* X,
                         int64_t ldx, int64_t d, const double* Xq, int64_t m, int64_t ldq, double* var); /* :248-273 */
void fro_predict_mean_variance(const fro_kprog* p, const double* L, int64_t n, int64_t ldl, const double* X,
                               int64_t ldx, int64_t d, const double* y, const double* Xq, int64_t m, int64_t ldq,
                               const double* prior_q, double* mean, double* var);                  /* :290-326 */
int fro_predict_covariance(const fro_kprog* p, const double* L, int64_t n, int64_t ldl, const double* X,
                           int64_t ldx, int64_t d, const double* Xq, int64_t m, int64_t ldq, double* cov,
                           int64_t ldc);                                                           /* :329-350 */
/* :371-392 + multivariate_normal.rs:54-59.  cov_l: m x m, cholesky().unpack() (upper zeroed).
 * Returns 0 or 1+j if the m x m Cholesky fails (reference: expect panic, multivariate_normal.rs:57). */
int fro_sample_at(const fro_kprog* p, const double* L, int64_t n, int64_t ldl, const double* X, int64_t ldx,
                  int64_t d, const double* y, const double* Xq, int64_t m, int64_t ldq, const double* prior_q,
                  double* mean, double* cov, double* cov_l);
/* multivariate_normal.rs:68-73 with the normal draws z supplied by the caller */
void fro_mvn_sample(const double* mean, const double* cov_l, int64_t m, const double* z, double* out);

/* ---- src/gaussian_process/optimizer.rs ---- */
/* :24-60 : out_grad has nb_parameters()+1 entries (last = noise) */
void fro_gradient_marginal_likelihood(const fro_kprog* p, const double* L, int64_t n, int64_t ldl, const double* X,
                                      int64_t ldx, int64_t d, const double* y, double noise, double* out_grad);
/* :159-203 */
void fro_scaled_gradient_marginal_likelihood(const fro_kprog* p, const double* L, int64_t n, int64_t ldl,
                                             const double* X, int64_t ldx, int64_t d, const double* y,
                                             double* out_scale, double* out_grad);
/* :69-149 / :211-283.  L (n x n, ld ldl) holds the current factor and is replaced by the re-fitted one.
 * Wall-clock budget (max_time) is not modelled.  Returns the number of iterations run, or -(1+j) if a
 * re-factorisation failed at column j. */
int fro_optimize_parameters(fro_kprog* p, double* noise, double* L, int64_t n, int64_t ldl, const double* X,
                            int64_t ldx, int64_t d, const double* y, int has_eps, double eps, int max_iter,
                            double convergence_fraction);
int fro_scaled_optimize_parameters(fro_kprog* p, double* noise, double* L, int64_t n, int64_t ldl, const double* X,
                                   int64_t ldx, int64_t d, const double* y, int has_eps, double eps, int max_iter,
                                   double convergence_fraction);

#ifdef __cplusplus
}
#endif
#endif
