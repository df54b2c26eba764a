"""ctypes front-end of the CPU oracle (oracle/friedrich_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg,
never by the product package.  PARITY UNPINNED (see friedrich_oracle.h).

Kernel specs are nested tuples shared with the product binding so a test can hand the same spec to both:
    ("squared_exp", ls, ampl) | ("matern2", ls, ampl) | ("linear", c) | ("polynomial", alpha, c, d) |
    ("exponential", ls, ampl) | ("matern1", ls, ampl) | ("hyper_tan", alpha, c) | ("multiquadric", c) |
    ("rational_quadratic", alpha, ls) | ("sum", spec, spec) | ("prod", spec, spec)

`OracleGP` restates the host logic of src/gaussian_process/{mod,builder}.rs on top of the C primitives.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfriedrich_oracle.so")

MAX_OPS = 15
LEAF_KINDS = {
    "linear": (0, 1),
    "polynomial": (1, 3),
    "squared_exp": (2, 2),
    "gaussian": (2, 2),
    "exponential": (3, 2),
    "matern1": (4, 2),
    "matern2": (5, 2),
    "hyper_tan": (6, 2),
    "multiquadric": (7, 1),
    "rational_quadratic": (8, 2),
}
K_SUM, K_PROD = 100, 101


class KernelOp(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("nparams", ctypes.c_int32), ("params", ctypes.c_double * 3)]


class KProg(ctypes.Structure):
    _fields_ = [("nops", ctypes.c_int32), ("reserved", ctypes.c_int32), ("ops", KernelOp * MAX_OPS)]


def flatten_spec(spec):
    """nested tuple -> RPN list of (kind, [params])"""
    name = spec[0]
    if name in ("sum", "prod"):
        return flatten_spec(spec[1]) + flatten_spec(spec[2]) + [(K_SUM if name == "sum" else K_PROD, [])]
    kind, npar = LEAF_KINDS[name]
    params = [float(v) for v in spec[1:]]
    if len(params) != npar:
        raise ValueError(f"kernel {name} takes {npar} parameters, got {len(params)}")
    return [(kind, params)]


def fill_kprog(prog, spec):
    ops = flatten_spec(spec)
    if len(ops) > MAX_OPS:
        raise ValueError("kernel program too long")
    prog.nops = len(ops)
    prog.reserved = 0
    for i, (kind, params) in enumerate(ops):
        prog.ops[i].kind = kind
        prog.ops[i].nparams = len(params)
        for q in range(3):
            prog.ops[i].params[q] = params[q] if q < len(params) else 0.0
    return prog


def kprog(spec):
    if isinstance(spec, KProg):
        return spec
    return fill_kprog(KProg(), spec)


def build(force=False):
    """(Re)build libfriedrich_oracle.so with the committed Makefile."""
    src = os.path.join(_HERE, "friedrich_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


_lib = None
_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int64)
_i64 = ctypes.c_int64
_kp = ctypes.POINTER(KProg)


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        L.fro_kernel.restype = ctypes.c_double
        L.fro_kernel.argtypes = [_kp, _dp, _i64, _dp, _i64, _i64]
        L.fro_kernel_gradient.restype = ctypes.c_int
        L.fro_kernel_gradient.argtypes = [_kp, _dp, _i64, _dp, _i64, _i64, _dp]
        L.fro_fit_bandwidth_mean.restype = ctypes.c_double
        L.fro_fit_bandwidth_mean.argtypes = [_dp, _i64, _i64, _i64]
        L.fro_variance.restype = ctypes.c_double
        L.fro_variance.argtypes = [_dp, _i64]
        L.fro_mean.restype = ctypes.c_double
        L.fro_mean.argtypes = [_dp, _i64]
        L.fro_heuristic_fit.restype = None
        L.fro_heuristic_fit.argtypes = [_kp, _dp, _i64, _i64, _i64, _dp]
        L.fro_make_covariance_matrix.restype = None
        L.fro_make_covariance_matrix.argtypes = [_kp, _dp, _i64, _i64, _dp, _i64, _i64, _i64, _dp, _i64]
        L.fro_make_cholesky_cov_matrix.restype = ctypes.c_int
        L.fro_make_cholesky_cov_matrix.argtypes = [_kp, _dp, _i64, _i64, _i64, ctypes.c_double, ctypes.c_int,
                                                   ctypes.c_double, _dp, _i64, _ip, _ip]
        L.fro_cholesky.restype = ctypes.c_int
        L.fro_cholesky.argtypes = [_dp, _i64, _i64, ctypes.c_int, ctypes.c_double, _ip, _ip]
        L.fro_set_threads.restype = None
        L.fro_set_threads.argtypes = [ctypes.c_int]
        L.fro_get_threads.restype = ctypes.c_int
        L.fro_get_threads.argtypes = []
        L.fro_cholesky_cols_mt.restype = ctypes.c_int
        L.fro_cholesky_cols_mt.argtypes = [_dp, _i64, _i64, _i64, ctypes.c_int, ctypes.c_double, _ip, _ip]
        L.fro_make_cholesky_cov_matrix_cols_mt.restype = ctypes.c_int
        L.fro_make_cholesky_cov_matrix_cols_mt.argtypes = [_kp, _dp, _i64, _i64, _i64, ctypes.c_double, ctypes.c_int,
                                                           ctypes.c_double, _i64, _dp, _i64, _ip, _ip]
        L.fro_add_rows_cholesky_cov_matrix.restype = None
        L.fro_add_rows_cholesky_cov_matrix.argtypes = [_kp, _dp, _i64, _dp, _i64, _i64, _i64, _i64, ctypes.c_double]
        L.fro_make_gradient_covariance_matrices.restype = None
        L.fro_make_gradient_covariance_matrices.argtypes = [_kp, _dp, _i64, _i64, _i64, _dp]
        L.fro_solve_lower.restype = ctypes.c_int
        L.fro_solve_lower.argtypes = [_dp, _i64, _i64, _dp, _i64, _i64]
        for name in ("fro_ad_solve_lower", "fro_chol_solve"):
            getattr(L, name).restype = None
            getattr(L, name).argtypes = [_dp, _i64, _i64, _dp, _i64, _i64]
        L.fro_chol_inverse.restype = None
        L.fro_chol_inverse.argtypes = [_dp, _i64, _i64, _dp, _i64]
        L.fro_likelihood.restype = ctypes.c_double
        L.fro_likelihood.argtypes = [_kp, _dp, _i64, _i64, _dp, _i64, _i64, _dp, ctypes.c_double]
        L.fro_predict.restype = None
        L.fro_predict.argtypes = [_kp, _dp, _i64, _i64, _dp, _i64, _i64, _dp, _dp, _i64, _i64, _dp, _dp]
        L.fro_predict_variance.restype = ctypes.c_int
        L.fro_predict_variance.argtypes = [_kp, _dp, _i64, _i64, _dp, _i64, _i64, _dp, _i64, _i64, _dp]
        L.fro_predict_mean_variance.restype = None
        L.fro_predict_mean_variance.argtypes = [_kp, _dp, _i64, _i64, _dp, _i64, _i64, _dp, _dp, _i64, _i64, _dp,
                                                _dp, _dp]
        L.fro_predict_covariance.restype = ctypes.c_int
        L.fro_predict_covariance.argtypes = [_kp, _dp, _i64, _i64, _dp, _i64, _i64, _dp, _i64, _i64, _dp, _i64]
        L.fro_sample_at.restype = ctypes.c_int
        L.fro_sample_at.argtypes = [_kp, _dp, _i64, _i64, _dp, _i64, _i64, _dp, _dp, _i64, _i64, _dp, _dp, _dp, _dp]
        L.fro_mvn_sample.restype = None
        L.fro_mvn_sample.argtypes = [_dp, _dp, _i64, _dp, _dp]
        L.fro_gradient_marginal_likelihood.restype = None
        L.fro_gradient_marginal_likelihood.argtypes = [_kp, _dp, _i64, _i64, _dp, _i64, _i64, _dp, ctypes.c_double,
                                                       _dp]
        L.fro_scaled_gradient_marginal_likelihood.restype = None
        L.fro_scaled_gradient_marginal_likelihood.argtypes = [_kp, _dp, _i64, _i64, _dp, _i64, _i64, _dp, _dp, _dp]
        for name in ("fro_optimize_parameters", "fro_scaled_optimize_parameters"):
            getattr(L, name).restype = ctypes.c_int
            getattr(L, name).argtypes = [_kp, _dp, _dp, _i64, _i64, _dp, _i64, _i64, _dp, ctypes.c_int,
                                         ctypes.c_double, ctypes.c_int, ctypes.c_double]
        for name in ("fro_kprog_validate", "fro_kprog_nb_parameters", "fro_kprog_nb_gradients",
                     "fro_kprog_is_scalable"):
            getattr(L, name).restype = ctypes.c_int
            getattr(L, name).argtypes = [_kp]
        L.fro_kprog_get_parameters.restype = ctypes.c_int
        L.fro_kprog_get_parameters.argtypes = [_kp, _dp]
        L.fro_kprog_set_parameters.restype = ctypes.c_int
        L.fro_kprog_set_parameters.argtypes = [_kp, _dp, ctypes.c_int]
        L.fro_kprog_rescale.restype = ctypes.c_int
        L.fro_kprog_rescale.argtypes = [_kp, ctypes.c_double]
        _lib = L
    return _lib


# ------------------------------------------------------------------------------------------------
# array helpers


def fmat(a):
    """2-D float64 Fortran-ordered (column-major) copy/view"""
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 1:
        a = a.reshape(-1, 1)
    return np.asfortranarray(a)


def _ptr(a):
    return a.ctypes.data_as(_dp)


def _ld(a):
    """leading dimension of a column-major 2-D array (supports row-sliced views of F arrays)"""
    if a.ndim == 1:
        return a.shape[0]
    assert a.shape[0] <= 1 or a.strides[0] == 8, "need unit row stride (column-major)"
    return max(a.strides[1] // 8, a.shape[0], 1) if a.shape[1] > 1 else max(a.shape[0], 1)


def _vec(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))


# ------------------------------------------------------------------------------------------------
# primitives


def kernel(spec, x1, x2):
    p = kprog(spec)
    x1, x2 = _vec(x1), _vec(x2)
    return lib().fro_kernel(ctypes.byref(p), _ptr(x1), 1, _ptr(x2), 1, x1.shape[0])


def kernel_gradient(spec, x1, x2):
    p = kprog(spec)
    x1, x2 = _vec(x1), _vec(x2)
    out = np.zeros(3 * MAX_OPS)
    n = lib().fro_kernel_gradient(ctypes.byref(p), _ptr(x1), 1, _ptr(x2), 1, x1.shape[0], _ptr(out))
    return out[:n].copy()


def nb_parameters(spec):
    p = kprog(spec)
    return lib().fro_kprog_nb_parameters(ctypes.byref(p))


def get_parameters(spec):
    p = kprog(spec)
    out = np.zeros(3 * MAX_OPS)
    n = lib().fro_kprog_get_parameters(ctypes.byref(p), _ptr(out))
    return out[:n].copy()


def fit_bandwidth_mean(X):
    X = fmat(X)
    return lib().fro_fit_bandwidth_mean(_ptr(X), X.shape[0], _ld(X), X.shape[1])


def variance(y):
    y = _vec(y)
    return lib().fro_variance(_ptr(y), y.shape[0])


def mean(y):
    y = _vec(y)
    return lib().fro_mean(_ptr(y), y.shape[0])


def make_covariance_matrix(spec, A, B):
    p = kprog(spec)
    A, B = fmat(A), fmat(B)
    out = np.empty((A.shape[0], B.shape[0]), order="F")
    lib().fro_make_covariance_matrix(ctypes.byref(p), _ptr(A), A.shape[0], _ld(A), _ptr(B), B.shape[0], _ld(B),
                                     A.shape[1], _ptr(out), max(A.shape[0], 1))
    return out


def make_cholesky_cov_matrix(spec, X, noise, eps=None):
    """-> (status, L with NaN upper triangle, substituted column indices)"""
    p = kprog(spec)
    X = fmat(X)
    n = X.shape[0]
    out = np.empty((n, n), order="F")
    ns = ctypes.c_int64(0)
    idx = np.zeros(max(n, 1), dtype=np.int64)
    st = lib().fro_make_cholesky_cov_matrix(ctypes.byref(p), _ptr(X), n, _ld(X), X.shape[1], float(noise),
                                            0 if eps is None else 1, 0.0 if eps is None else float(eps), _ptr(out),
                                            max(n, 1), ctypes.byref(ns), idx.ctypes.data_as(_ip))
    return st, out, idx[:ns.value].copy()


def set_threads(n=0):
    """host threads for the independent loops (right-hand sides, Gram entries, axpy rows); n <= 0: every online core.
    Results are bit-identical for every thread count.  Returns the count in effect."""
    lib().fro_set_threads(int(n))
    return lib().fro_get_threads()


class threads:
    """with O.threads(): ... -- all cores inside the block, one thread (the reference's execution) outside"""

    def __init__(self, n=0):
        self.n = n

    def __enter__(self):
        self.prev = lib().fro_get_threads()
        return set_threads(self.n)

    def __exit__(self, *exc):
        lib().fro_set_threads(self.prev)


def make_cholesky_cov_matrix_cols(spec, X, noise, eps=None, ncols=None):
    """leading `ncols` columns of make_cholesky_cov_matrix's factor (n x ncols, NaN above the diagonal), computed with
    the blocked multi-thread schedule that is bit-identical to the reference order -> (status, Lcols, subst idx < ncols)"""
    p = kprog(spec)
    X = fmat(X)
    n = X.shape[0]
    ncols = n if ncols is None else min(int(ncols), n)
    out = np.empty((n, max(ncols, 1)), order="F")
    ns = ctypes.c_int64(0)
    idx = np.zeros(max(n, 1), dtype=np.int64)
    st = lib().fro_make_cholesky_cov_matrix_cols_mt(ctypes.byref(p), _ptr(X), n, _ld(X), X.shape[1], float(noise),
                                                    0 if eps is None else 1, 0.0 if eps is None else float(eps), ncols,
                                                    _ptr(out), max(n, 1), ctypes.byref(ns), idx.ctypes.data_as(_ip))
    return st, out[:, :ncols], idx[:ns.value].copy()


def cholesky_cols(A, sub=None, ncols=None):
    """fro_cholesky on the leading ncols columns of a copy of A (blocked, threaded, bit-identical)"""
    A = fmat(A).copy(order="F")
    n = A.shape[0]
    ncols = n if ncols is None else min(int(ncols), n)
    ns = ctypes.c_int64(0)
    idx = np.zeros(max(n, 1), dtype=np.int64)
    st = lib().fro_cholesky_cols_mt(_ptr(A), n, max(n, 1), ncols, 0 if sub is None else 1,
                                    0.0 if sub is None else float(sub), ctypes.byref(ns), idx.ctypes.data_as(_ip))
    return st, A[:, :ncols], idx[:ns.value].copy()


def cholesky(A, sub=None):
    """in-place left-looking Cholesky of the lower triangle of a copy of A -> (status, L, subst idx)"""
    A = fmat(A).copy(order="F")
    n = A.shape[0]
    ns = ctypes.c_int64(0)
    idx = np.zeros(max(n, 1), dtype=np.int64)
    st = lib().fro_cholesky(_ptr(A), n, max(n, 1), 0 if sub is None else 1, 0.0 if sub is None else float(sub),
                            ctypes.byref(ns), idx.ctypes.data_as(_ip))
    return st, A, idx[:ns.value].copy()


def add_rows_cholesky_cov_matrix(spec, L_old, X_all, nb_new, noise):
    p = kprog(spec)
    X_all = fmat(X_all)
    n_all = X_all.shape[0]
    n_old = n_all - nb_new
    L = np.full((n_all, n_all), np.nan, order="F")
    L[:n_old, :n_old] = L_old[:n_old, :n_old]
    lib().fro_add_rows_cholesky_cov_matrix(ctypes.byref(p), _ptr(L), n_all, _ptr(X_all), n_all, _ld(X_all),
                                           X_all.shape[1], nb_new, float(noise))
    return L


def make_gradient_covariance_matrices(spec, X):
    p = kprog(spec)
    X = fmat(X)
    n = X.shape[0]
    npar = lib().fro_kprog_nb_parameters(ctypes.byref(p))
    out = np.empty(npar * n * n)
    lib().fro_make_gradient_covariance_matrices(ctypes.byref(p), _ptr(X), n, _ld(X), X.shape[1], _ptr(out))
    return [out[q * n * n:(q + 1) * n * n].reshape((n, n), order="F") for q in range(npar)]


def solve_lower(L, B):
    L, B = fmat(L), fmat(B).copy(order="F")
    st = lib().fro_solve_lower(_ptr(L), L.shape[0], _ld(L), _ptr(B), B.shape[1], max(B.shape[0], 1))
    return st, B


def ad_solve_lower(L, B):
    L, B = fmat(L), fmat(B).copy(order="F")
    lib().fro_ad_solve_lower(_ptr(L), L.shape[0], _ld(L), _ptr(B), B.shape[1], max(B.shape[0], 1))
    return B


def chol_solve(L, B):
    L, B = fmat(L), fmat(B).copy(order="F")
    lib().fro_chol_solve(_ptr(L), L.shape[0], _ld(L), _ptr(B), B.shape[1], max(B.shape[0], 1))
    return B


def chol_inverse(L):
    L = fmat(L)
    n = L.shape[0]
    out = np.empty((n, n), order="F")
    lib().fro_chol_inverse(_ptr(L), n, _ld(L), _ptr(out), max(n, 1))
    return out


# ------------------------------------------------------------------------------------------------
# priors (src/parameters/prior.rs) -- host-side in the reference and here


class ZeroPrior:  # prior.rs:43-56
    def __init__(self, input_dimension=0):
        pass

    def prior(self, X):
        return np.zeros(np.asarray(X).shape[0])

    def fit(self, X, y):
        pass


class ConstantPrior:  # prior.rs:66-99
    def __init__(self, c=0.0):
        self.c = float(c)

    def prior(self, X):
        return np.full(np.asarray(X).shape[0], self.c)

    def fit(self, X, y):
        self.c = mean(y)  # :97


class LinearPrior:  # prior.rs:108-160
    def __init__(self, weights, intercept=0.0):
        self.weights = np.asarray(weights, dtype=np.float64).reshape(-1)
        self.intercept = float(intercept)

    @classmethod
    def default(cls, input_dimension):
        return cls(np.zeros(input_dimension), 0.0)

    def prior(self, X):
        return np.asarray(X, dtype=np.float64) @ self.weights + self.intercept  # :133-136

    def fit(self, X, y):
        # :139-159 : SVD least squares on [1 | X]
        # [nalgebra] SVD::solve(b, eps = 0): x = V diag(1 / sigma_i if sigma_i > eps else 0) U^T b
        A = np.hstack([np.ones((np.asarray(X).shape[0], 1)), np.asarray(X, dtype=np.float64)])
        U, sv, Vt = np.linalg.svd(A, full_matrices=False)
        inv = np.array([1.0 / v if v > 0.0 else 0.0 for v in sv])
        w = Vt.T @ (inv * (U.T @ np.asarray(y, dtype=np.float64)))
        self.intercept = float(w[0])
        self.weights = w[1:].copy()


# ------------------------------------------------------------------------------------------------
# GaussianProcess restated (src/gaussian_process/mod.rs, builder.rs)


class OracleGP:
    """Restatement of GaussianProcess<K, P>: owns X, residual y and the factor L (NaN upper triangle)."""

    def __init__(self, prior, kernel_spec, noise, cholesky_epsilon, X, y):  # mod.rs:142-167
        assert noise >= 0.0
        self.prior = prior
        self.prog = kprog(kernel_spec) if not isinstance(kernel_spec, KProg) else kernel_spec
        self.noise = float(noise)
        self.cholesky_epsilon = cholesky_epsilon
        self.X = fmat(X).copy(order="F")
        y = _vec(y)
        assert self.X.shape[0] == y.shape[0]
        self.y = y - prior.prior(self.X)  # :156
        st, self.L, self.subst = make_cholesky_cov_matrix(self.prog, self.X, self.noise, cholesky_epsilon)
        if st:
            raise FloatingPointError(f"Cholesky decomposition failed at column {st - 1}")

    # builder.rs:66-95 + :189-214 with fit_kernel().fit_prior() == GaussianProcess::default (mod.rs:96-102)
    @classmethod
    def default(cls, X, y, max_iter=100, convergence_fraction=0.05):
        X = fmat(X)
        y = _vec(y)
        prog = kprog(("squared_exp", 1.0, 1.0))
        noise = 0.1 * np.sqrt(variance(y))  # builder.rs:73
        lib().fro_heuristic_fit(ctypes.byref(prog), _ptr(X), X.shape[0], _ld(X), X.shape[1], _ptr(y))  # :195
        gp = cls(ConstantPrior(0.0), prog, noise, None, X, y)
        gp.fit_parameters(True, True, max_iter, convergence_fraction)
        return gp

    @property
    def n(self):
        return self.X.shape[0]

    def kernel_parameters(self):
        return get_parameters(self.prog)

    def add_samples(self, X_new, y_new):  # mod.rs:173-190
        X_new = fmat(X_new)
        y_new = _vec(y_new)
        assert X_new.shape[0] == y_new.shape[0]
        assert X_new.shape[1] == self.X.shape[1]
        y_new = y_new - self.prior.prior(X_new)
        self.X = np.asfortranarray(np.vstack([self.X, X_new]))
        self.y = np.concatenate([self.y, y_new])
        self.L = add_rows_cholesky_cov_matrix(self.prog, self.L, self.X, X_new.shape[0], self.noise)

    def likelihood(self):  # mod.rs:196-220
        return lib().fro_likelihood(ctypes.byref(self.prog), _ptr(self.L), self.n, _ld(self.L), _ptr(self.X),
                                    _ld(self.X), self.X.shape[1], _ptr(self.y), self.noise)

    def _q(self, Xq):
        Xq = fmat(Xq)
        assert Xq.shape[1] == self.X.shape[1]
        return Xq, _vec(self.prior.prior(Xq))

    def predict(self, Xq):  # mod.rs:226-244
        Xq, pq = self._q(Xq)
        out = np.empty(Xq.shape[0])
        lib().fro_predict(ctypes.byref(self.prog), _ptr(self.L), self.n, _ld(self.L), _ptr(self.X), _ld(self.X),
                          self.X.shape[1], _ptr(self.y), _ptr(Xq), Xq.shape[0], _ld(Xq), _ptr(pq), _ptr(out))
        return out

    def predict_variance(self, Xq):  # mod.rs:248-273
        Xq, _ = self._q(Xq)
        out = np.empty(Xq.shape[0])
        st = lib().fro_predict_variance(ctypes.byref(self.prog), _ptr(self.L), self.n, _ld(self.L), _ptr(self.X),
                                        _ld(self.X), self.X.shape[1], _ptr(Xq), Xq.shape[0], _ld(Xq), _ptr(out))
        if st:
            raise FloatingPointError("predict_covariance : solve failed")
        return out

    def predict_mean_variance(self, Xq):  # mod.rs:290-326
        Xq, pq = self._q(Xq)
        mean_, var = np.empty(Xq.shape[0]), np.empty(Xq.shape[0])
        lib().fro_predict_mean_variance(ctypes.byref(self.prog), _ptr(self.L), self.n, _ld(self.L), _ptr(self.X),
                                        _ld(self.X), self.X.shape[1], _ptr(self.y), _ptr(Xq), Xq.shape[0], _ld(Xq),
                                        _ptr(pq), _ptr(mean_), _ptr(var))
        return mean_, var

    def predict_covariance(self, Xq):  # mod.rs:329-350
        Xq, _ = self._q(Xq)
        m = Xq.shape[0]
        cov = np.empty((m, m), order="F")
        st = lib().fro_predict_covariance(ctypes.byref(self.prog), _ptr(self.L), self.n, _ld(self.L), _ptr(self.X),
                                          _ld(self.X), self.X.shape[1], _ptr(Xq), m, _ld(Xq), _ptr(cov), max(m, 1))
        if st:
            raise FloatingPointError("predict_covariance : solve failed")
        return cov

    def sample_at(self, Xq):  # mod.rs:371-392 -> (mean, cov, chol(cov).unpack())
        Xq, pq = self._q(Xq)
        m = Xq.shape[0]
        mean_ = np.empty(m)
        cov = np.empty((m, m), order="F")
        cov_l = np.empty((m, m), order="F")
        st = lib().fro_sample_at(ctypes.byref(self.prog), _ptr(self.L), self.n, _ld(self.L), _ptr(self.X),
                                 _ld(self.X), self.X.shape[1], _ptr(self.y), _ptr(Xq), m, _ld(Xq), _ptr(pq),
                                 _ptr(mean_), _ptr(cov), _ptr(cov_l))
        if st:
            raise FloatingPointError("MultivariateNormal: Cholesky decomposition failed!")
        return mean_, cov, cov_l

    @staticmethod
    def mvn_sample(mean_, cov_l, z):  # multivariate_normal.rs:68-73
        mean_, z = _vec(mean_), _vec(z)
        cov_l = fmat(cov_l)
        out = np.empty(mean_.shape[0])
        lib().fro_mvn_sample(_ptr(mean_), _ptr(cov_l), mean_.shape[0], _ptr(z), _ptr(out))
        return out

    def scaled_gradient(self):  # optimizer.rs:159-203
        npar = lib().fro_kprog_nb_parameters(ctypes.byref(self.prog))
        scale = ctypes.c_double(0.0)
        g = np.zeros(npar)
        lib().fro_scaled_gradient_marginal_likelihood(ctypes.byref(self.prog), _ptr(self.L), self.n, _ld(self.L),
                                                      _ptr(self.X), _ld(self.X), self.X.shape[1], _ptr(self.y),
                                                      ctypes.byref(scale), _ptr(g))
        return scale.value, g

    def gradient(self):  # optimizer.rs:24-60
        npar = lib().fro_kprog_nb_parameters(ctypes.byref(self.prog))
        g = np.zeros(npar + 1)
        lib().fro_gradient_marginal_likelihood(ctypes.byref(self.prog), _ptr(self.L), self.n, _ld(self.L),
                                               _ptr(self.X), _ld(self.X), self.X.shape[1], _ptr(self.y), self.noise,
                                               _ptr(g))
        return g

    def fit_parameters(self, fit_prior, fit_kernel, max_iter=100, convergence_fraction=0.05):  # mod.rs:406-445
        self.iterations = 0
        if fit_prior:
            y_full = self.y + self.prior.prior(self.X)  # :416-417
            self.prior.fit(self.X, y_full)
            self.y = y_full - self.prior.prior(self.X)
            if not fit_kernel:  # :423-430
                st, self.L, self.subst = make_cholesky_cov_matrix(self.prog, self.X, self.noise,
                                                                  self.cholesky_epsilon)
                if st:
                    raise FloatingPointError("Cholesky decomposition failed")
        if fit_kernel:  # :434-444
            scalable = lib().fro_kprog_is_scalable(ctypes.byref(self.prog))
            fn = lib().fro_scaled_optimize_parameters if scalable else lib().fro_optimize_parameters
            noise = ctypes.c_double(self.noise)
            eps = self.cholesky_epsilon
            it = fn(ctypes.byref(self.prog), ctypes.byref(noise), _ptr(self.L), self.n, _ld(self.L), _ptr(self.X),
                    _ld(self.X), self.X.shape[1], _ptr(self.y), 0 if eps is None else 1,
                    0.0 if eps is None else float(eps), int(max_iter), float(convergence_fraction))
            if it < 0:
                raise FloatingPointError(f"optimizer failed ({it})")
            self.noise = noise.value
            self.iterations = it
