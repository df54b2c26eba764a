// friedrich.hpp -- TEST HARNESS (not a product component): a C++ stand-in for the reference's own host code, so that the
// reference's doctests / src/main.rs can be replayed against the C ABI (include/friedrich_amd.h) in an image without a
// Rust toolchain.  In a real integration these layers (builder, priors, Input conversion, ADAM scalars: SURVEY.md
// section 2 rows 6, 8, 9 -- OUT OF SCOPE) stay friedrich's unchanged Rust; see INTEGRATION.md for the Rust side.
//
// The reference is a Rust crate and this image has no Rust toolchain, so the host side that a Rust maintainer would
// write is mirrored here in C++ with the reference's names, argument meaning and error behaviour:
//
//   friedrich::GaussianProcess<Kernel, Prior>      src/gaussian_process/mod.rs:59-445
//   friedrich::GaussianProcessBuilder<Kernel,Prior> src/gaussian_process/builder.rs:35-214
//   friedrich::MultivariateNormal                   src/gaussian_process/multivariate_normal.rs:44-73
//   kernels (Linear ... RationalQuadratic, KernelSum, KernelProd)   src/parameters/kernel.rs
//   priors  (ZeroPrior, ConstantPrior, LinearPrior)                  src/parameters/prior.rs
//
// Everything numerical is delegated to libfriedrich_amd.so: this header holds the host logic only (input
// conversion, prior evaluation, ADAM scalars, builder defaults).  The reference panics; this mirror throws
// std::runtime_error carrying the reference's panic text.  Inputs follow `conversion::Input` (conversion/mod.rs):
// a `std::vector<double>` is ONE sample (1 x d), a `std::vector<std::vector<double>>` is one sample per row.
#pragma once

#include <chrono>
#include <cmath>
#include <cstdint>
#include <memory>
#include <random>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "friedrich_amd.h"

namespace friedrich {

// ---- column-major matrix (nalgebra DMatrix stand-in) ------------------------------------------------------------------
struct DMatrix {
    int64_t rows = 0, cols = 0;
    std::vector<double> data;  // column-major, ld == rows
    DMatrix() = default;
    DMatrix(int64_t r, int64_t c, double v = 0.0) : rows(r), cols(c), data((size_t)(r * c), v) {}
    double& operator()(int64_t r, int64_t c) { return data[(size_t)(r + c * rows)]; }
    double operator()(int64_t r, int64_t c) const { return data[(size_t)(r + c * rows)]; }
    const double* ptr() const { return data.data(); }
    double* ptr() { return data.data(); }
    int64_t ld() const { return rows > 0 ? rows : 1; }
};
using DVector = std::vector<double>;

// conversion::Input (conversion/mod.rs:95-146)
inline DMatrix to_dmatrix(const std::vector<double>& row)
{
    DMatrix m(1, (int64_t)row.size());
    for (size_t c = 0; c < row.size(); ++c) m(0, (int64_t)c) = row[c];
    return m;
}
inline DMatrix to_dmatrix(const std::vector<std::vector<double>>& rows)
{
    if (rows.empty()) throw std::runtime_error("assertion failed: !m.is_empty()");  // conversion/mod.rs:130
    const int64_t n = (int64_t)rows.size(), d = (int64_t)rows[0].size();
    DMatrix m(n, d);
    for (int64_t r = 0; r < n; ++r) {
        if ((int64_t)rows[(size_t)r].size() != d) throw std::runtime_error("inconsistent row width");
        for (int64_t c = 0; c < d; ++c) m(r, c) = rows[(size_t)r][(size_t)c];
    }
    return m;
}
inline DMatrix to_dmatrix(const DMatrix& m) { return m; }

// f64::powi (compiler-rt __powidf2: square-and-multiply), used by the ADAM bias correction (optimizer.rs:116-117)
inline double powi(double a, int b)
{
    const bool recip = b < 0;
    double r = 1.0;
    while (true) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return recip ? 1.0 / r : r;
}

// ---- process-wide device context ---------------------------------------------------------------------------------------
inline fr_ctx* default_context()
{
    static fr_ctx* ctx = [] {
        fr_ctx* c = nullptr;
        const int st = fr_ctx_create(&c, -1);
        if (st != FR_OK) throw std::runtime_error("friedrich_amd: no gfx950 device (fr_ctx_create failed)");
        return c;
    }();
    return ctx;
}
inline void check(fr_ctx* ctx, int st)
{
    if (st != FR_OK) throw std::runtime_error(std::string("friedrich_amd: ") + fr_last_error(ctx));
}

// ---- kernels (src/parameters/kernel.rs) --------------------------------------------------------------------------------
inline void push_leaf(fr_kprog& p, int kind, std::initializer_list<double> params)
{
    if (p.nops >= FR_KPROG_MAX_OPS) throw std::runtime_error("kernel program too long");
    fr_kernel_op& op = p.ops[p.nops++];
    op.kind = kind;
    op.nparams = (int32_t)params.size();
    int i = 0;
    for (double v : params) op.params[i++] = v;
    for (; i < 3; ++i) op.params[i] = 0.0;
}
inline void push_op(fr_kprog& p, int kind)
{
    if (p.nops >= FR_KPROG_MAX_OPS) throw std::runtime_error("kernel program too long");
    fr_kernel_op& op = p.ops[p.nops++];
    op.kind = kind;
    op.nparams = 0;
    op.params[0] = op.params[1] = op.params[2] = 0.0;
}

// Every kernel offers the `Kernel` trait surface (kernel.rs:22-86) plus device_program.
#define FRIEDRICH_SCALABLE_KERNEL(NAME, KIND)                                                              \
    struct NAME {                                                                                          \
        double ls = 1.0, ampl = 1.0;                                                                       \
        NAME() = default;                                                                                  \
        NAME(double ls_, double ampl_) : ls(ls_), ampl(ampl_) {}                                           \
        size_t nb_parameters() const { return 2; }                                                         \
        bool is_scalable() const { return true; }                                                          \
        void rescale(double scale) { ampl *= scale; }                                                      \
        std::vector<double> get_parameters() const { return {ls, ampl}; }                                  \
        void set_parameters(const double* p, size_t) { ls = p[0]; ampl = p[1]; }                           \
        void device_program(fr_kprog& p) const { push_leaf(p, KIND, {ls, ampl}); }                         \
        void heuristic_fit(double bandwidth, double amplitude) { ls = bandwidth; ampl = amplitude; }       \
        static constexpr bool has_heuristic = true;                                                        \
    }
FRIEDRICH_SCALABLE_KERNEL(SquaredExp, FR_K_SQUAREDEXP);    // kernel.rs:496-601
FRIEDRICH_SCALABLE_KERNEL(Exponential, FR_K_EXPONENTIAL);  // :612-706
FRIEDRICH_SCALABLE_KERNEL(Matern1, FR_K_MATERN1);          // :717-813
FRIEDRICH_SCALABLE_KERNEL(Matern2, FR_K_MATERN2);          // :824-925
using Gaussian = SquaredExp;                               // :496

struct Linear {  // kernel.rs:342-402
    double c = 0.0;
    Linear() = default;
    explicit Linear(double c_) : c(c_) {}
    size_t nb_parameters() const { return 1; }
    bool is_scalable() const { return false; }
    void rescale(double) { throw std::runtime_error("You tried to rescale a Kernel that is not Scalable!"); }
    std::vector<double> get_parameters() const { return {c}; }
    void set_parameters(const double* p, size_t) { c = p[0]; }
    void device_program(fr_kprog& p) const { push_leaf(p, FR_K_LINEAR, {c}); }
    void heuristic_fit(double, double) {}
    static constexpr bool has_heuristic = false;
};
struct Polynomial {  // :411-485
    double alpha = 1.0, c = 0.0, d = 1.0;
    Polynomial() = default;
    Polynomial(double a, double c_, double d_) : alpha(a), c(c_), d(d_) {}
    size_t nb_parameters() const { return 3; }
    bool is_scalable() const { return false; }
    void rescale(double) { throw std::runtime_error("You tried to rescale a Kernel that is not Scalable!"); }
    std::vector<double> get_parameters() const { return {alpha, c, d}; }
    void set_parameters(const double* p, size_t) { alpha = p[0]; c = p[1]; d = p[2]; }
    void device_program(fr_kprog& p) const { push_leaf(p, FR_K_POLYNOMIAL, {alpha, c, d}); }
    void heuristic_fit(double, double) {}
    static constexpr bool has_heuristic = false;
};
struct HyperTan {  // :934-1001
    double alpha = 1.0, c = 0.0;
    HyperTan() = default;
    HyperTan(double a, double c_) : alpha(a), c(c_) {}
    size_t nb_parameters() const { return 2; }
    bool is_scalable() const { return false; }
    void rescale(double) { throw std::runtime_error("You tried to rescale a Kernel that is not Scalable!"); }
    std::vector<double> get_parameters() const { return {alpha, c}; }
    void set_parameters(const double* p, size_t) { alpha = p[0]; c = p[1]; }
    void device_program(fr_kprog& p) const { push_leaf(p, FR_K_HYPERTAN, {alpha, c}); }
    void heuristic_fit(double, double) {}
    static constexpr bool has_heuristic = false;
};
struct Multiquadric {  // :1010-1070 (nb_parameters() == 2 and set_parameters reads index 1, as the reference)
    double c = 0.0;
    Multiquadric() = default;
    explicit Multiquadric(double c_) : c(c_) {}
    size_t nb_parameters() const { return 2; }
    bool is_scalable() const { return false; }
    void rescale(double) { throw std::runtime_error("You tried to rescale a Kernel that is not Scalable!"); }
    std::vector<double> get_parameters() const { return {c}; }
    void set_parameters(const double* p, size_t n)
    {
        if (n < 2) throw std::out_of_range("index out of bounds: the len is 1 but the index is 1");  // :1068
        c = p[1];
    }
    void device_program(fr_kprog& p) const { push_leaf(p, FR_K_MULTIQUADRIC, {c}); }
    void heuristic_fit(double, double) {}
    static constexpr bool has_heuristic = false;
};
struct RationalQuadratic {  // :1079-1157
    double alpha = 1.0, ls = 1.0;
    RationalQuadratic() = default;
    RationalQuadratic(double a, double l) : alpha(a), ls(l) {}
    size_t nb_parameters() const { return 2; }
    bool is_scalable() const { return false; }
    void rescale(double) { throw std::runtime_error("You tried to rescale a Kernel that is not Scalable!"); }
    std::vector<double> get_parameters() const { return {alpha, ls}; }
    void set_parameters(const double* p, size_t) { alpha = p[0]; ls = p[1]; }
    void device_program(fr_kprog& p) const { push_leaf(p, FR_K_RATIONALQUADRATIC, {alpha, ls}); }
    void heuristic_fit(double, double) {}
    static constexpr bool has_heuristic = false;
};

template <class T, class U>
struct KernelSum {  // kernel.rs:132-211
    T k1;
    U k2;
    size_t nb_parameters() const { return k1.nb_parameters() + k2.nb_parameters(); }
    bool is_scalable() const { return k1.is_scalable() && k2.is_scalable(); }
    void rescale(double s) { k1.rescale(s); k2.rescale(s); }
    std::vector<double> get_parameters() const
    {
        auto p = k1.get_parameters();
        auto q = k2.get_parameters();
        p.insert(p.end(), q.begin(), q.end());
        return p;
    }
    void set_parameters(const double* p, size_t n)
    {
        const size_t n1 = k1.nb_parameters();
        k1.set_parameters(p, n1 < n ? n1 : n);
        k2.set_parameters(p + n1, n > n1 ? n - n1 : 0);
    }
    void device_program(fr_kprog& p) const { k1.device_program(p); k2.device_program(p); push_op(p, FR_K_SUM); }
    void heuristic_fit(double b, double a) { k1.heuristic_fit(b, a); k2.heuristic_fit(b, a); }
    static constexpr bool has_heuristic = T::has_heuristic || U::has_heuristic;
};
template <class T, class U>
struct KernelProd {  // kernel.rs:221-307
    T k1;
    U k2;
    size_t nb_parameters() const { return k1.nb_parameters() + k2.nb_parameters(); }
    bool is_scalable() const { return k1.is_scalable() || k2.is_scalable(); }
    void rescale(double s) { if (k1.is_scalable()) k1.rescale(s); else k2.rescale(s); }
    std::vector<double> get_parameters() const
    {
        auto p = k1.get_parameters();
        auto q = k2.get_parameters();
        p.insert(p.end(), q.begin(), q.end());
        return p;
    }
    void set_parameters(const double* p, size_t n)
    {
        const size_t n1 = k1.nb_parameters();
        k1.set_parameters(p, n1 < n ? n1 : n);
        k2.set_parameters(p + n1, n > n1 ? n - n1 : 0);
    }
    void device_program(fr_kprog& p) const { k1.device_program(p); k2.device_program(p); push_op(p, FR_K_PROD); }
    void heuristic_fit(double b, double a) { k1.heuristic_fit(b, a); k2.heuristic_fit(b, a); }
    static constexpr bool has_heuristic = T::has_heuristic || U::has_heuristic;
};
// KernelArith (kernel.rs:312-332): `+` / `*` build KernelSum / KernelProd; constrained to types offering the Kernel surface
template <class T, class = void> struct is_kernel : std::false_type {};
template <class T>
struct is_kernel<T, std::void_t<decltype(std::declval<const T&>().device_program(std::declval<fr_kprog&>())),
                                decltype(std::declval<const T&>().nb_parameters())>> : std::true_type {};
template <class T, class U, class = std::enable_if_t<is_kernel<T>::value && is_kernel<U>::value>>
KernelSum<T, U> operator+(const T& a, const U& b) { return {a, b}; }
template <class T, class U, class = std::enable_if_t<is_kernel<T>::value && is_kernel<U>::value>>
KernelProd<T, U> operator*(const T& a, const U& b) { return {a, b}; }

template <class K>
fr_kprog program_of(const K& k)
{
    fr_kprog p;
    p.nops = 0;
    p.reserved = 0;
    k.device_program(p);
    return p;
}

// ---- priors (src/parameters/prior.rs) ----------------------------------------------------------------------------------
struct ZeroPrior {  // :43-56
    static ZeroPrior default_(size_t) { return {}; }
    DVector prior(const DMatrix& x) const { return DVector((size_t)x.rows, 0.0); }
    void fit(const DMatrix&, const DVector&) {}
};
struct ConstantPrior {  // :66-99
    double c = 0.0;
    ConstantPrior() = default;
    explicit ConstantPrior(double c_) : c(c_) {}
    static ConstantPrior default_(size_t) { return ConstantPrior(0.0); }
    DVector prior(const DMatrix& x) const { return DVector((size_t)x.rows, c); }
    void fit(const DMatrix&, const DVector& y)
    {
        double s = 0.0;
        for (double v : y) s += v;
        c = y.empty() ? 0.0 : s / (double)y.size();  // training_outputs.mean() :97
    }
};
struct LinearPrior {  // :108-160
    DVector weights;
    double intercept = 0.0;
    LinearPrior() = default;
    LinearPrior(DVector w, double b) : weights(std::move(w)), intercept(b) {}
    static LinearPrior default_(size_t d) { return LinearPrior(DVector(d, 0.0), 0.0); }
    DVector prior(const DMatrix& x) const
    {
        DVector out((size_t)x.rows, intercept);
        for (int64_t c = 0; c < x.cols; ++c)
            for (int64_t r = 0; r < x.rows; ++r) out[(size_t)r] += x(r, c) * weights[(size_t)c];
        return out;
    }
    // least squares on [1 | X] with the reference's SVD-solve numerics (:139-159): through the ABI (fr_linear_prior_fit:
    // tall-skinny QR on the device + SVD of the small triangle)
    void fit(const DMatrix& x, const DVector& y)
    {
        fr_ctx* ctx = nullptr;
        if (fr_ctx_create(&ctx, -1) != FR_OK) throw std::runtime_error("Linear prior fit : no device.");
        weights.assign((size_t)x.cols, 0.0);
        const int st = fr_linear_prior_fit(ctx, x.data.data(), x.rows, x.rows > 0 ? x.rows : 1, x.cols, y.data(), weights.data(),
                                           &intercept);
        fr_ctx_destroy(ctx);
        if (st != FR_OK) throw std::runtime_error("Linear prior fit : solve failed.");
    }
};

// ---- MultivariateNormal (multivariate_normal.rs:44-73) -------------------------------------------------------------------
struct MultivariateNormal {
    DVector mean_;
    DMatrix cholesky_covariance;  // cholesky(cov).unpack(): strict upper triangle zeroed
    const DVector& mean() const { return mean_; }
    // mean + L * z with z ~ N(0, 1) drawn from the caller's generator (:68-73)
    template <class RNG>
    DVector sample(RNG& rng) const
    {
        std::normal_distribution<double> normal(0.0, 1.0);
        const int64_t m = (int64_t)mean_.size();
        DVector z((size_t)m);
        for (auto& v : z) v = normal(rng);
        return sample_with(z);
    }
    DVector sample_with(const DVector& z) const
    {
        const int64_t m = (int64_t)mean_.size();
        DVector out = mean_;
        for (int64_t c = 0; c < m; ++c)
            for (int64_t r = c; r < m; ++r) out[(size_t)r] += cholesky_covariance(r, c) * z[(size_t)c];
        return out;
    }
};

// ---- GaussianProcess (src/gaussian_process/mod.rs) -------------------------------------------------------------------------
template <class KernelType, class PriorType>
class GaussianProcessBuilder;

template <class KernelType = Gaussian, class PriorType = ConstantPrior>
class GaussianProcess {
  public:
    PriorType prior;
    KernelType kernel;
    double noise;
    bool has_cholesky_epsilon;
    double cholesky_epsilon;

    // GaussianProcess::new (mod.rs:142-167)
    template <class In>
    GaussianProcess(PriorType prior_, KernelType kernel_, double noise_, bool has_eps, double eps, const In& training_inputs,
                    const DVector& training_outputs)
        : prior(std::move(prior_)), kernel(std::move(kernel_)), noise(noise_), has_cholesky_epsilon(has_eps),
          cholesky_epsilon(eps), ctx_(default_context())
    {
        if (!(noise >= 0.0))
            throw std::runtime_error("The noise parameter should non-negative but we tried to set it to " + std::to_string(noise));
        X_ = to_dmatrix(training_inputs);
        if (X_.rows != (int64_t)training_outputs.size()) throw std::runtime_error("assertion failed: `(left == right)`");  // :153
        const DVector p = prior.prior(X_);
        y_ = training_outputs;
        for (size_t i = 0; i < y_.size(); ++i) y_[i] -= p[i];  // :156
        const fr_kprog prog = program_of(kernel);
        fr_chol* h = nullptr;
        const int st = fr_chol_from_inputs(ctx_, &prog, X_.ptr(), X_.rows, X_.ld(), X_.cols, noise, has_eps ? 1 : 0, eps, 0, &h);
        chol_.reset(h);
        raise_factor_status(st);
    }
    GaussianProcess(GaussianProcess&&) = default;
    GaussianProcess& operator=(GaussianProcess&&) = default;

    // GaussianProcess::default (mod.rs:96-102) and ::builder (:129-135)
    template <class In>
    static GaussianProcess<Gaussian, ConstantPrior> default_(const In& training_inputs, const DVector& training_outputs);
    template <class In>
    static GaussianProcessBuilder<Gaussian, ConstantPrior> builder(const In& training_inputs, const DVector& training_outputs);

    // add_samples (mod.rs:173-190)
    template <class In>
    void add_samples(const In& inputs, const DVector& outputs)
    {
        const DMatrix in = to_dmatrix(inputs);
        if (in.rows != (int64_t)outputs.size()) throw std::runtime_error("assertion failed: `(left == right)`");  // :177
        if (in.cols != X_.cols) throw std::runtime_error("assertion failed: `(left == right)`");                  // :178
        const DVector p = prior.prior(in);
        DMatrix all(X_.rows + in.rows, X_.cols);
        for (int64_t c = 0; c < X_.cols; ++c) {
            for (int64_t r = 0; r < X_.rows; ++r) all(r, c) = X_(r, c);
            for (int64_t r = 0; r < in.rows; ++r) all(X_.rows + r, c) = in(r, c);
        }
        for (int64_t r = 0; r < in.rows; ++r) y_.push_back(outputs[(size_t)r] - p[(size_t)r]);  // :180-182
        X_ = std::move(all);
        const fr_kprog prog = program_of(kernel);
        check(ctx_, fr_chol_add_rows(chol_.get(), &prog, X_.ptr(), X_.rows, X_.ld(), X_.cols, in.rows, noise));  // :185-189
    }

    // likelihood (mod.rs:196-220)
    double likelihood() const
    {
        const fr_kprog prog = program_of(kernel);
        double out = 0.0;
        const int st = fr_likelihood(chol_.get(), &prog, y_.data(), noise, &out);
        if (st == FR_SINGULAR_SOLVE) throw std::runtime_error("likelihood : solve failed");
        check(ctx_, st);
        return out;
    }

    // predict (mod.rs:226-244)
    template <class In>
    DVector predict(const In& inputs) const
    {
        const DMatrix q = query(inputs);
        const DVector p = prior.prior(q);
        DVector mean((size_t)q.rows);
        const fr_kprog prog = program_of(kernel);
        check(ctx_, fr_predict_mean(chol_.get(), &prog, y_.data(), q.ptr(), q.rows, q.ld(), p.data(), mean.data()));
        return mean;
    }
    // predict_variance (mod.rs:248-273)
    template <class In>
    DVector predict_variance(const In& inputs) const
    {
        const DMatrix q = query(inputs);
        DVector var((size_t)q.rows);
        const fr_kprog prog = program_of(kernel);
        const int st = fr_predict_variance(chol_.get(), &prog, q.ptr(), q.rows, q.ld(), var.data());
        if (st == FR_SINGULAR_SOLVE) throw std::runtime_error("predict_covariance : solve failed");  // :263
        check(ctx_, st);
        return var;
    }
    // predict_mean_variance (mod.rs:290-326)
    template <class In>
    std::pair<DVector, DVector> predict_mean_variance(const In& inputs) const
    {
        const DMatrix q = query(inputs);
        const DVector p = prior.prior(q);
        DVector mean((size_t)q.rows), var((size_t)q.rows);
        const fr_kprog prog = program_of(kernel);
        check(ctx_, fr_predict_mean_variance(chol_.get(), &prog, y_.data(), q.ptr(), q.rows, q.ld(), p.data(), mean.data(),
                                             var.data()));
        return {mean, var};
    }
    // predict_covariance (mod.rs:329-350)
    template <class In>
    DMatrix predict_covariance(const In& inputs) const
    {
        const DMatrix q = query(inputs);
        DMatrix cov(q.rows, q.rows);
        const fr_kprog prog = program_of(kernel);
        const int st = fr_predict_covariance(chol_.get(), &prog, q.ptr(), q.rows, q.ld(), cov.ptr(), cov.ld());
        if (st == FR_SINGULAR_SOLVE) throw std::runtime_error("predict_covariance : solve failed");  // :345
        check(ctx_, st);
        return cov;
    }
    // sample_at (mod.rs:371-392)
    template <class In>
    MultivariateNormal sample_at(const In& inputs) const
    {
        const DMatrix q = query(inputs);
        const DVector p = prior.prior(q);
        MultivariateNormal mvn;
        mvn.mean_.resize((size_t)q.rows);
        mvn.cholesky_covariance = DMatrix(q.rows, q.rows);
        const fr_kprog prog = program_of(kernel);
        const int st = fr_posterior(chol_.get(), &prog, y_.data(), q.ptr(), q.rows, q.ld(), p.data(), mvn.mean_.data(), nullptr, 1,
                                    mvn.cholesky_covariance.ptr(), mvn.cholesky_covariance.ld());
        if (st == FR_NOT_POSITIVE_DEFINITE)
            throw std::runtime_error("MultivariateNormal: Cholesky decomposition failed!");  // multivariate_normal.rs:57
        check(ctx_, st);
        return mvn;
    }

    // fit_parameters (mod.rs:406-445)
    void fit_parameters(bool fit_prior, bool fit_kernel, size_t max_iter, double convergence_fraction,
                        std::chrono::duration<double> max_time)
    {
        if (fit_prior) {
            DVector p = prior.prior(X_);
            DVector full = y_;
            for (size_t i = 0; i < full.size(); ++i) full[i] += p[i];  // :416-417
            prior.fit(X_, full);
            p = prior.prior(X_);
            for (size_t i = 0; i < full.size(); ++i) y_[i] = full[i] - p[i];  // :419-420
            if (!fit_kernel) refactor();                                      // :423-430
        }
        if (fit_kernel) {
            if (kernel.is_scalable())
                scaled_optimize_parameters(max_iter, convergence_fraction, max_time);  // :436-439
            else
                optimize_parameters(max_iter, convergence_fraction, max_time);
        }
    }

    int64_t nb_samples() const { return X_.rows; }
    size_t last_fit_iterations() const { return iterations_; }
    std::vector<int64_t> substituted_columns() const
    {
        int64_t ns = 0;
        fr_chol_info(chol_.get(), nullptr, nullptr, nullptr, &ns, nullptr);
        std::vector<int64_t> idx((size_t)ns);
        fr_chol_substitutions(chol_.get(), idx.data(), ns);
        return idx;
    }

  private:
    struct CholDeleter {
        void operator()(fr_chol* c) const { fr_chol_free(c); }
    };
    fr_ctx* ctx_;
    DMatrix X_;  // EMatrix stand-in (host copy; the device keeps its own inside fr_chol)
    DVector y_;  // residual training outputs (EVector)
    std::unique_ptr<fr_chol, CholDeleter> chol_;
    size_t iterations_ = 0;

    template <class In>
    DMatrix query(const In& inputs) const
    {
        DMatrix q = to_dmatrix(inputs);
        if (q.cols != X_.cols) throw std::runtime_error("assertion failed: `(left == right)`");  // :231, 253, 293, 334, 374
        return q;
    }
    void raise_factor_status(int st) const
    {
        if (st == FR_NOT_POSITIVE_DEFINITE) {
            if (has_cholesky_epsilon)  // algebra/mod.rs:85
                throw std::runtime_error("Cholesky decomposition failed even though we used `cholesky_epsilon` value of " +
                                         std::to_string(cholesky_epsilon));
            throw std::runtime_error(  // algebra/mod.rs:90
                "Cholesky decomposition failed, consider setting `cholesky_epsilon` via `GaussianProcessBuilder`");
        }
        check(ctx_, st);
    }
    void refactor()
    {
        const fr_kprog prog = program_of(kernel);
        raise_factor_status(fr_chol_refactor(chol_.get(), &prog, noise, has_cholesky_epsilon ? 1 : 0, cholesky_epsilon));
    }

    // optimizer.rs:69-149
    void optimize_parameters(size_t max_iter, double convergence_fraction, std::chrono::duration<double> max_time)
    {
        const double beta1 = 0.9, beta2 = 0.999, epsilon = 1e-8, learning_rate = 0.1;
        std::vector<double> parameters = kernel.get_parameters();
        for (auto& p : parameters)
            if (p == 0.0) p = epsilon;
        parameters.push_back(std::log(noise));  // :98
        std::vector<double> mean_grad(parameters.size(), 0.0), var_grad(parameters.size(), 0.0);
        const auto time_start = std::chrono::steady_clock::now();
        iterations_ = 0;
        for (size_t i = 1; i <= max_iter; ++i) {
            iterations_ = i;
            std::vector<double> gradients(kernel.nb_parameters() + 1);
            const fr_kprog prog = program_of(kernel);
            check(ctx_, fr_grad_terms(chol_.get(), &prog, y_.data(), noise, 0, gradients.data(), nullptr));
            gradients.back() *= noise;  // :106-110
            bool had_significant_progress = false;
            for (size_t p = 0; p < parameters.size(); ++p) {  // :113-122
                mean_grad[p] = beta1 * mean_grad[p] + (1.0 - beta1) * gradients[p];
                var_grad[p] = beta2 * var_grad[p] + (1.0 - beta2) * gradients[p] * gradients[p];
                const double bias_corrected_mean = mean_grad[p] / (1.0 - powi(beta1, (int)i));
                const double bias_corrected_variance = var_grad[p] / (1.0 - powi(beta2, (int)i));
                const double delta = learning_rate * bias_corrected_mean / (std::sqrt(bias_corrected_variance) + epsilon);
                had_significant_progress |= std::fabs(delta) > convergence_fraction;
                parameters[p] *= 1.0 + delta;
            }
            kernel.set_parameters(parameters.data(), parameters.size());  // :125
            noise = std::exp(parameters.back());                          // :126-130
            refactor();                                                   // :133-136
            if (!had_significant_progress || (std::chrono::steady_clock::now() - time_start) > max_time) break;  // :138
        }
    }

    // optimizer.rs:211-283
    void scaled_optimize_parameters(size_t max_iter, double convergence_fraction, std::chrono::duration<double> max_time)
    {
        const double beta1 = 0.9, beta2 = 0.999, epsilon = 1e-8, learning_rate = 0.1;
        std::vector<double> parameters = kernel.get_parameters();
        for (auto& p : parameters)
            if (p == 0.0) p = epsilon;
        std::vector<double> mean_grad(parameters.size(), 0.0), var_grad(parameters.size(), 0.0);
        const auto time_start = std::chrono::steady_clock::now();
        iterations_ = 0;
        for (size_t i = 1; i <= max_iter; ++i) {
            iterations_ = i;
            std::vector<double> gradients(kernel.nb_parameters() + 1);
            double scale = 1.0;
            const fr_kprog prog = program_of(kernel);
            check(ctx_, fr_grad_terms(chol_.get(), &prog, y_.data(), noise, 1, gradients.data(), &scale));  // :246
            bool had_significant_progress = false;
            for (size_t p = 0; p < parameters.size(); ++p) {  // :249-258
                mean_grad[p] = beta1 * mean_grad[p] + (1.0 - beta1) * gradients[p];
                var_grad[p] = beta2 * var_grad[p] + (1.0 - beta2) * gradients[p] * gradients[p];
                const double bias_corrected_mean = mean_grad[p] / (1.0 - powi(beta1, (int)i));
                const double bias_corrected_variance = var_grad[p] / (1.0 - powi(beta2, (int)i));
                const double delta = learning_rate * bias_corrected_mean / (std::sqrt(bias_corrected_variance) + epsilon);
                had_significant_progress |= std::fabs(delta) > convergence_fraction;
                parameters[p] *= 1.0 + delta;
            }
            kernel.set_parameters(parameters.data(), parameters.size());  // :261
            kernel.rescale(scale);                                        // :262
            noise *= scale;                                               // :263 (noise, not noise^2 -- as the reference)
            parameters = kernel.get_parameters();                         // :264
            refactor();                                                   // :267-270
            if (!had_significant_progress || (std::chrono::steady_clock::now() - time_start) > max_time) break;  // :272
        }
    }

    template <class K2, class P2>
    friend class GaussianProcessBuilder;
};

// ---- GaussianProcessBuilder (src/gaussian_process/builder.rs) ---------------------------------------------------------------
template <class KernelType, class PriorType>
class GaussianProcessBuilder {
  public:
    // builder.rs:66-95
    template <class In>
    GaussianProcessBuilder(const In& training_inputs, const DVector& training_outputs)
        : X_(to_dmatrix(training_inputs)), y_(training_outputs), prior_(PriorType::default_((size_t)X_.cols)), kernel_()
    {
        // noise = 0.1 * sqrt(row_variance(y)[0]): population variance, E[x^2] - E[x]^2 (nalgebra 0.31)
        double s2 = 0.0, s1 = 0.0;
        for (double v : y_) {
            s2 += v * v;
            s1 += v;
        }
        const double inv = y_.empty() ? 0.0 : 1.0 / (double)y_.size();
        const double var = s2 * inv - (s1 * inv) * (s1 * inv);
        variance_ = var;
        noise_ = 0.1 * std::sqrt(var);  // :73
    }
    template <class NewPrior>
    GaussianProcessBuilder<KernelType, NewPrior> set_prior(NewPrior p) const  // :102-118
    {
        GaussianProcessBuilder<KernelType, NewPrior> b(X_, y_, std::move(p), kernel_);
        copy_settings(b);
        return b;
    }
    GaussianProcessBuilder& set_noise(double noise)  // :122-126
    {
        if (!(noise >= 0.0))
            throw std::runtime_error("The noise parameter should non-negative but we tried to set it to " + std::to_string(noise));
        noise_ = noise;
        return *this;
    }
    template <class NewKernel>
    GaussianProcessBuilder<NewKernel, PriorType> set_kernel(NewKernel k) const  // :130-145
    {
        GaussianProcessBuilder<NewKernel, PriorType> b(X_, y_, prior_, std::move(k));
        copy_settings(b);
        return b;
    }
    GaussianProcessBuilder& set_cholesky_epsilon(bool has, double eps = 0.0)  // :156-159 (Option<f64>)
    {
        has_eps_ = has;
        eps_ = eps;
        return *this;
    }
    GaussianProcessBuilder& set_fit_parameters(size_t max_iter, double convergence_fraction)  // :165-168
    {
        max_iter_ = max_iter;
        convergence_fraction_ = convergence_fraction;
        return *this;
    }
    GaussianProcessBuilder& fit_kernel() { should_fit_kernel_ = true; return *this; }  // :172-175
    GaussianProcessBuilder& fit_prior() { should_fit_prior_ = true; return *this; }    // :179-182

    // train (builder.rs:189-214)
    GaussianProcess<KernelType, PriorType> train()
    {
        if (should_fit_kernel_ && KernelType::has_heuristic) {
            // heuristic_fit (kernel.rs:594-600): ls = mean pairwise distance (device reduction), ampl = var(y)
            double bandwidth = 0.0;
            fr_ctx* ctx = default_context();
            check(ctx, fr_mean_pairwise_distance(ctx, X_.ptr(), X_.rows, X_.ld(), X_.cols, &bandwidth));
            kernel_.heuristic_fit(bandwidth, variance_);
        }
        GaussianProcess<KernelType, PriorType> gp(prior_, kernel_, noise_, has_eps_, eps_, X_, y_);
        gp.fit_parameters(should_fit_prior_, should_fit_kernel_, max_iter_, convergence_fraction_, max_time_);
        return gp;
    }

    GaussianProcessBuilder(DMatrix X, DVector y, PriorType p, KernelType k)
        : X_(std::move(X)), y_(std::move(y)), prior_(std::move(p)), kernel_(std::move(k))
    {
    }

  private:
    DMatrix X_;
    DVector y_;
    PriorType prior_;
    KernelType kernel_;
    double noise_ = 0.0, variance_ = 0.0;
    bool has_eps_ = false;
    double eps_ = 0.0;
    bool should_fit_kernel_ = false, should_fit_prior_ = false;
    size_t max_iter_ = 100;
    double convergence_fraction_ = 0.05;
    std::chrono::duration<double> max_time_ = std::chrono::seconds(3600);

    template <class B>
    void copy_settings(B& b) const
    {
        b.noise_ = noise_;
        b.variance_ = variance_;
        b.has_eps_ = has_eps_;
        b.eps_ = eps_;
        b.should_fit_kernel_ = should_fit_kernel_;
        b.should_fit_prior_ = should_fit_prior_;
        b.max_iter_ = max_iter_;
        b.convergence_fraction_ = convergence_fraction_;
        b.max_time_ = max_time_;
    }
    template <class K2, class P2>
    friend class GaussianProcessBuilder;
};

template <class K, class P>
template <class In>
GaussianProcess<Gaussian, ConstantPrior> GaussianProcess<K, P>::default_(const In& training_inputs, const DVector& training_outputs)
{
    return GaussianProcessBuilder<Gaussian, ConstantPrior>(training_inputs, training_outputs).fit_kernel().fit_prior().train();
}
template <class K, class P>
template <class In>
GaussianProcessBuilder<Gaussian, ConstantPrior> GaussianProcess<K, P>::builder(const In& training_inputs,
                                                                               const DVector& training_outputs)
{
    return GaussianProcessBuilder<Gaussian, ConstantPrior>(training_inputs, training_outputs);
}

}  // namespace friedrich
