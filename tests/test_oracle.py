"""CPU checks that pin the oracle (oracle/friedrich_oracle.c).

The reference's own tests hold no numeric assertion, so the oracle is "parity unpinned" with
respect to friedrich itself; these tests pin it against independent implementations instead: 50-digit mpmath for
the per-pair kernel formulas, scipy/LAPACK for the factorisation and solves, and algebraic identities
(add_samples == refit, K K^-1 == I) for the composite paths."""
import mpmath as mp
import numpy as np
import pytest
import scipy.linalg as sl

from conftest import ALL_KERNELS, PD_KERNELS, rand_inputs, rel_err
from oracle import oracle as O


def mp_kernel(spec, x, y):
    """independent 50-digit restatement of src/parameters/kernel.rs (Appendix B of SURVEY.md)"""
    mp.mp.dps = 50
    name = spec[0]
    if name == "sum":
        return mp_kernel(spec[1], x, y) + mp_kernel(spec[2], x, y)
    if name == "prod":
        return mp_kernel(spec[1], x, y) * mp_kernel(spec[2], x, y)
    x = [mp.mpf(float(v)) for v in x]
    y = [mp.mpf(float(v)) for v in y]
    s = sum((a - b) ** 2 for a, b in zip(x, y))
    u = sum(a * b for a, b in zip(x, y))
    r = mp.sqrt(s)
    P = [mp.mpf(float(v)) for v in spec[1:]]
    if name == "linear":
        return u + P[0]
    if name == "polynomial":
        return (P[0] * u + P[1]) ** P[2]
    if name in ("squared_exp", "gaussian"):
        return abs(P[1]) * mp.exp(-s / (2 * P[0] * P[0]))
    if name == "exponential":
        return abs(P[1]) * mp.exp(-r / (2 * P[0] * P[0]))
    if name == "matern1":
        xx = mp.sqrt(3) * r / abs(P[0])
        return abs(P[1]) * (1 + xx) * mp.exp(-xx)
    if name == "matern2":
        l = abs(P[0])
        xx = mp.sqrt(5) * r / l
        return abs(P[1]) * (1 + xx + 5 * r * r / (3 * l * l)) * mp.exp(-xx)
    if name == "hyper_tan":
        return mp.tanh(P[0] * u + P[1])
    if name == "multiquadric":
        return mp.sqrt(s * s + P[0] * P[0])  # hypot(||x-y||^2, c), as the reference writes it (kernel.rs:1049)
    if name == "rational_quadratic":
        return (1 + s / (2 * P[0] * P[1] * P[1])) ** (-P[0])
    raise ValueError(name)


@pytest.mark.parametrize("kernel", ALL_KERNELS, ids=lambda k: k[0] + str(len(k)))
def test_kernel_values_against_mpmath(kernel):
    rng = np.random.default_rng(0)
    for d in (1, 3, 16):
        for _ in range(5):
            x, y = rng.random(d), rng.random(d)
            want = float(mp_kernel(kernel, x, y))
            got = O.kernel(kernel, x, y)
            assert abs(got - want) <= 4e-15 * max(1.0, abs(want))


def test_kernel_gradient_layout_and_product_rule():
    x, y = np.array([0.3, 0.9]), np.array([0.7, 0.1])
    a, b = ("squared_exp", 0.8, 1.3), ("matern2", 0.7, -1.2)
    ga, gb = O.kernel_gradient(a, x, y), O.kernel_gradient(b, x, y)
    ka, kb = O.kernel(a, x, y), O.kernel(b, x, y)
    assert np.allclose(O.kernel_gradient(("sum", a, b), x, y), np.concatenate([ga, gb]), rtol=0, atol=0)
    assert np.allclose(O.kernel_gradient(("prod", a, b), x, y), np.concatenate([ga * kb, gb * ka]), rtol=1e-15)
    # SquaredExp gradient against finite differences (kernel.rs:563-576); grad_ampl carries the sign of ampl
    h = 1e-6
    fd_ls = (O.kernel(("squared_exp", 0.8 + h, 1.3), x, y) - O.kernel(("squared_exp", 0.8 - h, 1.3), x, y)) / (2 * h)
    assert abs(ga[0] - fd_ls) < 1e-8
    assert gb[1] < 0  # signum(ampl) * ...
    # Multiquadric declares 2 parameters but yields one gradient (kernel.rs:1039-1059)
    assert O.nb_parameters(("multiquadric", 0.5)) == 2
    assert len(O.kernel_gradient(("multiquadric", 0.5), x, y)) == 1


@pytest.mark.parametrize("kernel", PD_KERNELS, ids=lambda k: k[0] + str(len(k)))
def test_cholesky_against_lapack(kernel):
    X = rand_inputs(300, 4, 1)
    st, L, idx = O.make_cholesky_cov_matrix(kernel, X, 0.1)
    assert st == 0 and len(idx) == 0
    K = O.make_covariance_matrix(kernel, X, X) + 0.01 * np.eye(300)
    assert rel_err(np.tril(L), sl.cholesky(K, lower=True)) < 1e-12
    assert np.all(np.isnan(L[np.triu_indices(300, 1)]))  # algebra/mod.rs:67


def test_cholesky_substitute_rule():
    # nalgebra Cholesky::new_internal (Appendix A.1): a non-positive pivot takes sqrt(substitute) and the column
    # is scaled by it; without a usable substitute the factorisation fails at that column
    A = np.array([[4.0, 2.0], [2.0, 0.5]])  # second pivot = 0.5 - 1 = -0.5
    st, L, idx = O.cholesky(A)
    assert st == 2
    st, L, idx = O.cholesky(A, sub=0.25)
    assert st == 0 and idx.tolist() == [1]
    assert np.allclose(np.tril(L), [[2.0, 0.0], [1.0, 0.5]])
    st, _, _ = O.cholesky(A, sub=0.0)  # sqrt_denom(0) is None
    assert st == 2
    st, _, _ = O.cholesky(A, sub=-1.0)
    assert st == 2
    st, _, _ = O.cholesky(np.array([[0.0]]))  # exact zero pivot fails too
    assert st == 1
    st, L, idx = O.cholesky(np.array([[np.nan]]), sub=9.0)  # NaN pivot takes the substitute
    assert st == 0 and L[0, 0] == 3.0 and idx.tolist() == [0]


def test_solves_against_lapack():
    X = rand_inputs(200, 3, 2)
    k = ("matern2", 0.7, 1.2)
    st, L, _ = O.make_cholesky_cov_matrix(k, X, 0.1)
    Lt = np.tril(L)
    B = np.random.default_rng(0).standard_normal((200, 7))
    assert O.solve_lower(L, B)[0] == 0
    assert rel_err(O.solve_lower(L, B)[1], sl.solve_triangular(Lt, B, lower=True)) < 1e-12
    assert rel_err(O.ad_solve_lower(L, B), sl.solve_triangular(Lt, B, lower=True, trans="T")) < 1e-12
    assert rel_err(O.chol_solve(L, B), sl.cho_solve((Lt, True), B)) < 1e-11
    K = Lt @ Lt.T
    assert rel_err(O.chol_inverse(L) @ K, np.eye(200)) < 1e-10
    Z = Lt.copy()
    Z[5, 5] = 0.0
    assert O.solve_lower(Z, B)[0] == -1  # checked solve: zero on the diagonal (mod.rs:203,263,345)


@pytest.mark.parametrize("chunks", [[1], [3, 5], [40]])
def test_add_rows_equals_refit(chunks):
    k = ("squared_exp", 0.8, 1.3)
    Xall = rand_inputs(60 + sum(chunks), 3, 3)
    n = 60
    st, L, _ = O.make_cholesky_cov_matrix(k, Xall[:n], 0.1)
    for c in chunks:
        L = O.add_rows_cholesky_cov_matrix(k, L, Xall[:n + c], c, 0.1)
        n += c
    st, Lfull, _ = O.make_cholesky_cov_matrix(k, Xall, 0.1)
    assert rel_err(np.tril(L), np.tril(Lfull)) < 1e-13


def test_predict_family_against_dense_algebra():
    k = ("squared_exp", 0.7, 1.1)
    X, Xq = rand_inputs(150, 2, 4), rand_inputs(20, 2, 5)
    y = np.sin(X.sum(axis=1))
    prior = O.ConstantPrior(0.3)
    gp = O.OracleGP(prior, k, 0.1, None, X, y)
    K = O.make_covariance_matrix(k, X, X) + 0.01 * np.eye(150)
    Ks = O.make_covariance_matrix(k, X, Xq)
    Kss = O.make_covariance_matrix(k, Xq, Xq)
    Kinv = np.linalg.inv(K)
    mean = 0.3 + Ks.T @ Kinv @ (y - 0.3)
    cov = Kss - Ks.T @ Kinv @ Ks
    assert rel_err(gp.predict(Xq), mean) < 1e-10
    assert np.max(np.abs(gp.predict_variance(Xq) - np.diag(cov))) < 1e-10
    m2, v2 = gp.predict_mean_variance(Xq)
    assert rel_err(m2, mean) < 1e-10 and np.max(np.abs(v2 - np.diag(cov))) < 1e-10
    assert np.max(np.abs(gp.predict_covariance(Xq) - cov)) < 1e-10
    # likelihood as the reference writes it: log of the Gram DIAGONAL, not of L (mod.rs:208-213)
    want = -0.5 * ((y - 0.3) @ Kinv @ (y - 0.3) + np.sum(np.log(np.abs(np.diag(K)))) + 150 * np.log(2 * np.pi))
    assert abs(gp.likelihood() - want) < 1e-9 * abs(want)


def test_readme_example_default_fit():
    # src/main.rs:16-17 through GaussianProcess::default (heuristics + scaled ADAM); values are the oracle's own
    # (the reference prints, never asserts) and are frozen in tests/golden/readme.json
    gp = O.OracleGP.default([[0.8], [1.2], [3.8], [4.2]], [3.0, 4.0, -2.0, -2.0])
    assert gp.iterations == 10
    assert abs(gp.predict([[1.0]])[0] - 3.5490314) < 1e-6
    assert abs(gp.predict_variance([[1.0]])[0] - 0.0545010) < 1e-6
    assert abs(gp.prior.c - 0.75) < 1e-15


def test_heuristics():
    X = rand_inputs(80, 3, 6)
    dists = [np.linalg.norm(X[i] - X[j]) for i in range(80) for j in range(i + 1, 80)]
    assert abs(O.fit_bandwidth_mean(X) - np.mean(dists)) < 1e-13
    y = np.random.default_rng(1).standard_normal(50)
    assert abs(O.variance(y) - np.var(y)) < 1e-14  # population variance


# ---- host threads / blocked schedule: the same arithmetic, bit for bit ------------------------------------------------
@pytest.mark.parametrize("n,ncols,threads", [(1, 1, 4), (5, 5, 3), (300, 300, 4), (777, 500, 8), (1300, 1300, 5), (1300, 70, 8)])
def test_threaded_blocked_cholesky_is_bit_identical(n, ncols, threads):
    """fro_cholesky_cols_mt (right-looking panels of 64, rows dealt to threads) applies the updates of every element in
    ascending k with unfused multiply-adds, exactly like the left-looking restatement of nalgebra's Cholesky::new_internal:
    equal bits, equal substitution list, equal failure column."""
    rng = np.random.default_rng(n + ncols)
    Q = rng.standard_normal((n, n))
    A = Q @ Q.T + n * np.eye(n)
    if n > 100:
        A[n // 2, n // 2] = -1.0  # a negative pivot: failure without a substitute, one substitution with
    for sub in (None, 3.0):
        st, L1, idx1 = O.cholesky(A, sub)
        with O.threads(threads):
            st2, Lc, idx2 = O.cholesky_cols(A, sub, ncols)
        assert O.lib().fro_get_threads() == 1
        if st == 0 or st > ncols:
            assert st2 == 0
            assert np.array_equal(np.tril(Lc), np.tril(L1)[:, :ncols])
            assert idx2.tolist() == [i for i in idx1.tolist() if i < ncols]
        else:
            assert st2 == st


def test_threaded_gram_factor_and_solves_are_bit_identical():
    k = PD_KERNELS[1]
    X = rand_inputs(700, 4, 3)
    B = np.asfortranarray(np.random.default_rng(1).standard_normal((700, 37)))
    st, L1, _ = O.make_cholesky_cov_matrix(k, X, 0.1)
    K1 = O.make_covariance_matrix(k, X, X[:100])
    Z1, W1 = O.chol_solve(L1, B), O.solve_lower(L1, B)[1]
    with O.threads(6):
        st2, L2, _ = O.make_cholesky_cov_matrix_cols(k, X, 0.1)
        K2 = O.make_covariance_matrix(k, X, X[:100])
        Z2, W2 = O.chol_solve(L1, B), O.solve_lower(L1, B)[1]
        st3, L3, _ = O.make_cholesky_cov_matrix_cols(k, X, 0.1, None, 200)
    assert st == st2 == st3 == 0
    assert np.array_equal(np.tril(L1), np.tril(L2)) and np.array_equal(np.tril(L1)[:, :200], np.tril(L3))
    assert np.all(np.isnan(L2[np.triu_indices(700, 1)]))  # algebra/mod.rs:67
    assert np.array_equal(K1, K2) and np.array_equal(Z1, Z2) and np.array_equal(W1, W2)
