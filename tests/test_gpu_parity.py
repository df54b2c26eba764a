"""GPU parity tests: every call goes through the C ABI (libfriedrich_amd.so) and is compared with the CPU
oracle (oracle/) on the same seeded inputs.  Tolerance: BASELINE.json's north_star allows 1e-8 relative
error for f64 results; the tests hold the HIP path to TOL = 1e-9 (typically observed: 1e-13..1e-11), and the
substitution ("pivot") index lists to exact equality on margin fixtures."""
import numpy as np
import pytest

from conftest import ALL_KERNELS, PD_KERNELS, rand_inputs, rel_err
from oracle import oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-9        # north_star: <= 1e-8 relative
TOL_GRAM = 1e-13  # per-pair kernel values: only libm differences (exp/pow/tanh/hypot)


# ---- K5/K6 GEMM ------------------------------------------------------------------------------------
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
# (the last three shapes run interior 32-row tiles with K a multiple of 16 and at least four K-steps: the deep-prefetch path of
# gemm_f64_tile_m32, in all four storage combinations; the row tile 96..99 of the last one takes the predicated path)
@pytest.mark.parametrize("M,N,K", [(64, 64, 64), (200, 150, 77), (129, 1, 300), (1, 257, 5), (300, 260, 16), (17, 19, 0),
                                   (96, 256, 128), (160, 128, 64), (100, 128, 80)])
def test_gemm_matches_numpy(ctx, ta, tb, M, N, K):
    rng = np.random.default_rng(M * 1000 + N * 10 + K + ta * 2 + tb)
    A = np.asfortranarray(rng.standard_normal((K, M) if ta else (M, K)))
    B = np.asfortranarray(rng.standard_normal((N, K) if tb else (K, N)))
    C0 = np.asfortranarray(rng.standard_normal((M, N)))
    C = C0.copy(order="F")
    ctx.gemm(A, B, C, trans_a=bool(ta), trans_b=bool(tb), alpha=-0.75, beta=1.25)
    ref = -0.75 * ((A.T if ta else A) @ (B.T if tb else B)) + 1.25 * C0
    assert rel_err(C, ref) < 1e-13
    # beta == 0 must not read C (NaN-filled output buffer)
    C = np.full((M, N), np.nan, order="F")
    ctx.gemm(A, B, C, trans_a=bool(ta), trans_b=bool(tb), alpha=1.0, beta=0.0)
    assert rel_err(C, (A.T if ta else A) @ (B.T if tb else B)) < 1e-13


# ---- K1 Gram ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("kernel", ALL_KERNELS, ids=lambda k: k[0] + str(len(k)))
@pytest.mark.parametrize("n1,n2,d", [(150, 70, 5), (1, 1, 1), (129, 65, 16), (64, 200, 19)])
def test_gram_matches_oracle(ctx, kernel, n1, n2, d):
    A, B = rand_inputs(n1, d, 1), rand_inputs(n2, d, 2)
    got = ctx.gram(kernel, A, B)
    want = O.make_covariance_matrix(kernel, A, B)
    assert rel_err(got, want) < TOL_GRAM


def test_gram_exp_and_division_over_the_whole_argument_range(ctx):
    """The in-tree exp (Cody-Waite reduction + degree-13 polynomial + ldexp) and the three-instruction division by a
    launch-uniform divisor (kprog_device.hpp) against the oracle's libm over the whole range of arguments: exp(-t) for
    t = 0 ... 760 (gradual underflow and the flush to zero included) through the squared-exponential, exponential and
    Matern kernels on 1-D inputs.  An error of e ulps in the argument shows as e |t| 2^-53 relative in the value, so the
    bound is (4 + 2 t) ulps: both divisions are within 1 ulp of the correctly rounded quotient, both exps within 1 ulp."""
    t = np.concatenate([np.linspace(0.0, 760.0, 3001), np.array([1e-300, 1e-17, 0.5, 708.3, 709.78, 744.4, 745.13, 745.2, 746.0, 800.0, 1e4])])
    for spec, to_dist in ((("squared_exp", 0.7, 1.3), lambda tt: np.sqrt(2.0 * 0.7 * 0.7 * tt)),
                          (("exponential", 0.9, 0.7), lambda tt: 2.0 * 0.9 * 0.9 * tt),
                          (("matern1", 1.1, 0.9), lambda tt: tt * 1.1 / np.sqrt(3.0)),
                          (("matern2", 0.7, 1.2), lambda tt: tt * 0.7 / np.sqrt(5.0))):
        Y = np.asfortranarray(to_dist(t).reshape(-1, 1))
        X0 = np.zeros((1, 1), order="F")
        got = ctx.gram(spec, X0, Y)[0]
        want = O.make_covariance_matrix(spec, X0, Y)[0]
        ulp = np.spacing(np.maximum(np.abs(want), 5e-324))
        assert np.all(np.abs(got - want) <= (4.0 + 2.0 * t) * ulp), (spec, float(np.max(np.abs(got - want) / ulp)))
        assert np.all((want == 0.0) <= (got <= 5e-324 * 8))  # what the reference flushes to zero stays (sub)denormal-small


def test_gram_strided_inputs(ctx):
    # EMatrix::as_matrix(): leading dimension = capacity != nrows (extendable_matrix.rs:52-55)
    big = rand_inputs(300, 4, 3)
    view = big[:170]  # ld = 300
    k = ("matern2", 0.7, 1.2)
    import ctypes
    from friedrich_amd import _capi as C
    out = np.empty((170, 170), order="F")
    p = C.kprog(k)
    st = ctx.lib.fr_gram(ctx.h, ctypes.byref(p), big.ctypes.data, 170, 300, big.ctypes.data, 170, 300, 4,
                         out.ctypes.data, 170)
    assert st == 0
    assert rel_err(out, O.make_covariance_matrix(k, view, view)) < TOL_GRAM


def test_mean_pairwise_distance(ctx):
    for n, d in [(2, 1), (257, 3), (1000, 16)]:
        X = rand_inputs(n, d, n)
        assert abs(ctx.mean_pairwise_distance(X) / O.fit_bandwidth_mean(X) - 1.0) < 1e-12


# ---- K4/K5/K6 Cholesky -----------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 4, 63, 64, 65, 127, 128, 129, 200, 256, 300, 513, 1000])
@pytest.mark.parametrize("kernel", [PD_KERNELS[0], PD_KERNELS[1]], ids=["se", "matern2"])
def test_cholesky_matches_oracle(ctx, n, kernel):
    X = rand_inputs(n, 3, n)
    noise = 0.1
    st, L_o, idx_o = O.make_cholesky_cov_matrix(kernel, X, noise)
    assert st == 0 and len(idx_o) == 0
    chol = ctx.cholesky_from_inputs(kernel, X, noise)
    info = chol.info()
    assert info["n"] == n and info["n_subst"] == 0 and info["fail_col"] == -1
    L = chol.l()
    assert rel_err(L, np.tril(L_o)) < TOL
    assert np.all(np.triu(L, 1) == 0.0)
    Ln = chol.l(nan_upper=True)
    assert np.all(np.isnan(Ln[np.triu_indices(n, 1)]))  # serde image keeps NaN above the diagonal (algebra/mod.rs:67)
    chol.free()


@pytest.mark.parametrize("kernel", PD_KERNELS, ids=lambda k: k[0] + str(len(k)))
def test_cholesky_kernels_and_block_sizes(ctx, kernel):
    n = 700
    X = rand_inputs(n, 6, 11)
    st, L_o, _ = O.make_cholesky_cov_matrix(kernel, X, 0.05)
    assert st == 0
    for nb in (128, 256, 384, 512):
        ctx.set_option("nb", nb)
        chol = ctx.cholesky_from_inputs(kernel, X, 0.05)
        assert rel_err(chol.l(), np.tril(L_o)) < TOL
        chol.free()
    ctx.set_option("nb", 0)


def test_cholesky_large_outer_blocks(ctx):
    # nb = 768 / 1024 (the automatic choice at N >= 18432) with look-ahead active (n > 2 nb), against the oracle
    n = 2600
    kernel = PD_KERNELS[0]
    X = rand_inputs(n, 4, 17)
    st, L_o, _ = O.make_cholesky_cov_matrix(kernel, X, 0.05)
    assert st == 0
    for nb in (768, 1024):
        ctx.set_option("nb", nb)
        chol = ctx.cholesky_from_inputs(kernel, X, 0.05)
        assert rel_err(chol.l(), np.tril(L_o)) < TOL
        B = np.asfortranarray(np.random.default_rng(5).standard_normal((n, 3)))
        assert rel_err(chol.solve(B), O.chol_solve(L_o, B)) < 1e-8
        chol.free()
    ctx.set_option("nb", 0)


def test_gemm_split_k_and_batched_paths(ctx):
    # few result tiles + deep contraction: the product is cut along K (batched partial products + ordered reduction)
    rng = np.random.default_rng(12)
    for (M, N, K, ta, tb) in [(300, 200, 4096, False, False), (512, 512, 6144, False, True), (130, 700, 2048, True, False)]:
        A = rng.standard_normal((K, M) if ta else (M, K))
        B = rng.standard_normal((N, K) if tb else (K, N))
        C0 = rng.standard_normal((M, N))
        ref = 0.5 * (A.T if ta else A) @ (B.T if tb else B) - 2.0 * C0
        outs = []
        for sk in (1, 0):
            ctx.set_option("splitk", sk)
            C = np.asfortranarray(C0.copy())
            ctx.gemm(A, B, C=C, trans_a=ta, trans_b=tb, alpha=0.5, beta=-2.0)
            assert rel_err(C, ref) < 1e-12
            outs.append(C)
        ctx.set_option("splitk", 1)
        assert rel_err(outs[0], outs[1]) < 1e-13


def test_wide_solves_with_512_row_leaves(ctx):
    # >= 256 right-hand sides: the solves end in 512-row leaves (explicit 512-block inverses built on demand); n is not
    # a multiple of 512, so the last rows still take the 128-row path.  Also after add_rows (cache invalidated).
    n0, n, m = 1300, 1700, 300
    kernel = PD_KERNELS[1]
    X = rand_inputs(n, 5, 23)
    B = np.asfortranarray(np.random.default_rng(6).standard_normal((n, m)))
    chol = ctx.cholesky_from_inputs(kernel, X[:n0], 0.08, capacity_hint=n)
    _, L0, _ = O.make_cholesky_cov_matrix(kernel, X[:n0], 0.08)
    for leaf in (1, 0):
        ctx.set_option("leaf512", leaf)
        assert rel_err(chol.solve(B[:n0]), O.chol_solve(L0, B[:n0])) < 1e-8
        assert rel_err(chol.solve_lower(B[:n0]), O.solve_lower(L0, B[:n0])[1]) < 1e-9
    ctx.set_option("leaf512", 1)
    # a few right-hand sides take the memory-bound kernels (L read once per 16 columns); same recursion, same leaves
    for mm in (1, 2, 7, 16):
        bb = np.asfortranarray(B[:n0, :mm])
        assert rel_err(chol.solve(bb), O.chol_solve(L0, bb)) < 1e-8
        assert rel_err(chol.solve_lower(bb), O.solve_lower(L0, bb)[1]) < 1e-9
    chol.add_rows(kernel, X, n - n0, 0.08)
    _, L1, _ = O.make_cholesky_cov_matrix(kernel, X, 0.08)
    assert rel_err(chol.solve(B), O.chol_solve(L1, B)) < 1e-8
    assert rel_err(chol.solve_lower(B), O.solve_lower(L1, B)[1]) < 1e-9
    chol.free()


def test_readme_dataset(ctx):
    # the reference's only dataset (src/main.rs:16-17)
    X = np.array([[0.8], [1.2], [3.8], [4.2]])
    k = ("squared_exp", 1.0, 1.0)
    st, L_o, _ = O.make_cholesky_cov_matrix(k, X, 0.1)
    chol = ctx.cholesky_from_inputs(k, X, 0.1)
    assert rel_err(chol.l(), np.tril(L_o)) < 1e-14
    chol.free()


def test_substitution_indices_margin_fixture(ctx):
    # tanh "kernel" Gram matrices are robustly indefinite: every failing pivot is far from zero, so the
    # left-looking (reference) and blocked right-looking (HIP) factorisations must substitute the SAME columns
    n = 256
    X = rand_inputs(n, 2, 5) * 3.0
    k = ("hyper_tan", 1.0, 0.0)
    eps = 1e-6
    st, L_o, idx_o = O.make_cholesky_cov_matrix(k, X, 0.0, eps)
    assert st == 0 and len(idx_o) > 10
    chol = ctx.cholesky_from_inputs(k, X, 0.0, eps=eps)
    idx = chol.substitutions()
    assert idx.tolist() == idx_o.tolist()
    chol.free()


def test_substitution_from_matrix_negative_block(ctx):
    rng = np.random.default_rng(0)
    n = 200
    Q = rng.standard_normal((n, n))
    A = Q @ Q.T + n * np.eye(n)
    A[188:, 188:] -= 3.0 * n * np.eye(12)  # pivots 188.. strongly negative
    st, L_o, idx_o = O.cholesky(A, sub=200.0)
    assert st == 0 and len(idx_o) > 0
    chol = ctx.cholesky_from_matrix(A, eps=200.0)
    assert chol.substitutions().tolist() == idx_o.tolist()
    assert rel_err(chol.l(), np.tril(L_o)) < TOL
    chol.free()


def test_failure_status_and_column(ctx):
    from friedrich_amd.device import FriedrichError
    rng = np.random.default_rng(1)
    n = 300
    Q = rng.standard_normal((n, n))
    A = Q @ Q.T + n * np.eye(n)
    A[222, 222] = -5.0
    st_o, _, _ = O.cholesky(A)
    assert st_o == 1 + 222
    chol = ctx.cholesky_from_matrix(A, allow_failure=True)
    assert chol.info()["fail_col"] == 222
    chol.free()
    with pytest.raises(FriedrichError) as e:
        ctx.cholesky_from_matrix(A)
    assert e.value.status == 1
    # epsilon <= 0 cannot rescue the factorisation (algebra/mod.rs:85)
    chol = ctx.cholesky_from_matrix(A, eps=0.0, allow_failure=True)
    assert chol.info()["fail_col"] == 222
    chol.free()


# ---- solves ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,m", [(5, 1), (130, 3), (300, 200), (513, 129), (1000, 64)])
def test_solves_match_oracle(ctx, n, m):
    k = ("matern2", 0.7, 1.2)
    X = rand_inputs(n, 4, n + m)
    B = np.asfortranarray(np.random.default_rng(m).standard_normal((n, m)))
    st, L_o, _ = O.make_cholesky_cov_matrix(k, X, 0.1)
    chol = ctx.cholesky_from_inputs(k, X, 0.1)
    assert rel_err(chol.solve_lower(B), O.solve_lower(L_o, B)[1]) < TOL
    assert rel_err(chol.solve(B), O.chol_solve(L_o, B)) < TOL
    if n <= 513:
        assert rel_err(chol.inverse(), O.chol_inverse(L_o)) < TOL
    chol.free()


@pytest.mark.parametrize("n,m", [(4096, 192), (4096 + 700, 256), (6144 + 512, 700), (4100, 200)])
def test_big_leaf_solves_match_oracle(ctx, n, m):
    """A few hundred right-hand sides against >= 4096 rows: left-looking over 2048-row blocks with explicit 2048-block inverses
    (chol.hip: trsm_big; both leaf variants: split along K up to 640 columns, mirrored 32-row tiles above), the rows behind the
    last whole block through the 512-leaf recursion (700 / 4 rows: with and without a whole 512-block) -- forward
    (solve_lower_triangular, mod.rs:260-263) and forward + backward (Cholesky::solve, mod.rs:298, 379) against the oracle's
    substitutions, and against the library's own 512-leaf recursion."""
    k = ("squared_exp", 0.9, 1.3)
    X = rand_inputs(n, 6, n + m)
    B = np.asfortranarray(np.random.default_rng(m).standard_normal((n, m)))
    with O.threads():
        st, L_o, _ = O.make_cholesky_cov_matrix_cols(k, X, 0.3)
        L_o = np.tril(L_o)
        W_o = O.solve_lower(L_o, B)[1]
        Z_o = O.chol_solve(L_o, B)
    chol = ctx.cholesky_from_inputs(k, X, 0.3)
    W, Z = chol.solve_lower(B), chol.solve(B)
    assert rel_err(W, W_o) < TOL and rel_err(Z, Z_o) < TOL
    ctx.set_option("bigleaf_max", 0)
    try:
        W0, Z0 = chol.solve_lower(B), chol.solve(B)
    finally:
        ctx.set_option("bigleaf_max", -1)
    assert rel_err(W, W0) < 1e-11 and rel_err(Z, Z0) < 1e-11
    chol.free()


@pytest.mark.parametrize("m", [2, 16, 40])
def test_narrow_solves_switch_to_big_leaves_on_the_third_solve(ctx, m):
    """With fewer than 192 right-hand sides the 2048-row leaves are taken from the third narrow solve against a factor on
    (building their inverse blocks costs about two solves: chol.hip, use_big_leaves): the first solves run on K9 / the column
    groups, the later ones on the leaves -- every one of them against the oracle, forward and forward + backward, and the
    switch is undone by a refactorisation (the counter and the caches belong to a factor)."""
    n = 4100
    k = ("matern2", 0.9, 1.1)
    X = rand_inputs(n, 4, 4100 + m)
    B = np.asfortranarray(np.random.default_rng(m).standard_normal((n, m)))
    with O.threads():
        st, L_o, _ = O.make_cholesky_cov_matrix_cols(k, X, 0.2)
        L_o = np.tril(L_o)
        W_o, Z_o = O.solve_lower(L_o, B)[1], O.chol_solve(L_o, B)
    chol = ctx.cholesky_from_inputs(k, X, 0.2)
    for rep in range(2):
        outs = []
        for i in range(4):
            outs.append((chol.solve_lower(B), chol.solve(B)))
            assert rel_err(outs[-1][0], W_o) < TOL and rel_err(outs[-1][1], Z_o) < TOL, (rep, i)
        # two different paths produced these: equal to rounding, not bit for bit
        assert rel_err(outs[0][0], outs[3][0]) < 1e-11
        chol.refactor(k, 0.2)
    chol.free()


def test_big_leaf_inverses_follow_add_rows(ctx):
    """The cached 512- and 2048-block inverses are extended, not rebuilt, when rows are appended (the old blocks stay valid:
    algebra/mod.rs:108-124 only appends): a factor grown 4096 -> 6656 in 512-row chunks (its L21 solves of 512 right-hand sides
    take the big leaves themselves) solves like the directly factored one, block boundaries at 4096 (exact) and 6144 crossed."""
    k = ("squared_exp", 0.9, 1.3)
    n0, n1, m = 4096, 6656, 320
    X = rand_inputs(n1, 5, 17)
    B = np.asfortranarray(np.random.default_rng(3).standard_normal((n1, m)))
    grown = ctx.cholesky_from_inputs(k, X[:n0], 0.3, capacity_hint=n1)
    assert rel_err(grown.solve(B[:n0]), ctx.cholesky_from_inputs(k, X[:n0], 0.3).solve(B[:n0])) < 1e-11
    for hi in range(n0 + 512, n1 + 1, 512):
        grown.add_rows(k, np.asfortranarray(X[:hi]), 512, 0.3)
        if hi in (n0 + 512, n1):
            grown.solve_lower(B[:hi])  # (uses and extends the caches between appends)
    direct = ctx.cholesky_from_inputs(k, X, 0.3)
    assert rel_err(grown.l(), direct.l()) < 1e-10
    assert rel_err(grown.solve(B), direct.solve(B)) < 1e-9
    assert rel_err(grown.solve_lower(B), direct.solve_lower(B)) < 1e-9
    grown.free()
    direct.free()


def test_upload_download_roundtrip(ctx):
    k = ("squared_exp", 0.8, 1.3)
    n = 333
    X = rand_inputs(n, 3, 9)
    st, L_o, _ = O.make_cholesky_cov_matrix(k, X, 0.1)
    chol = ctx.cholesky_upload(L_o, X)  # NaN upper triangle, as serde would deliver it
    B = np.asfortranarray(np.random.default_rng(3).standard_normal((n, 7)))
    assert rel_err(chol.solve(B), O.chol_solve(L_o, B)) < TOL
    assert rel_err(chol.l(), np.tril(L_o)) == 0.0
    chol.free()


# ---- predict family --------------------------------------------------------------------------------
@pytest.mark.parametrize("kernel", [PD_KERNELS[0], PD_KERNELS[1], PD_KERNELS[4]], ids=["se", "matern2", "sum"])
@pytest.mark.parametrize("n,m,d", [(4, 1, 1), (200, 33, 2), (515, 130, 8)])
def test_predict_family_matches_oracle(ctx, kernel, n, m, d):
    X = rand_inputs(n, d, 100 + n)
    Xq = rand_inputs(m, d, 200 + m)
    y = np.sin(X.sum(axis=1)) + 0.3
    prior = O.ConstantPrior(0.25)
    gp = O.OracleGP(prior, kernel, 0.07, None, X, y)
    chol = ctx.cholesky_from_inputs(kernel, X, 0.07)
    yres = gp.y
    pq = prior.prior(Xq)
    assert rel_err(chol.predict_mean(kernel, yres, Xq, pq), gp.predict(Xq)) < TOL
    # the other association of the same product, K*^T (K^-1 y): same values up to rounding
    ctx.set_option("predict_assoc", 1)
    try:
        assert rel_err(chol.predict_mean(kernel, yres, Xq, pq), gp.predict(Xq)) < TOL
    finally:
        ctx.set_option("predict_assoc", 0)
    var = chol.predict_variance(kernel, Xq)
    assert np.max(np.abs(var - gp.predict_variance(Xq))) < TOL * np.max(np.abs(gp.predict_variance(Xq)) + 1.0)
    mean2, var2 = chol.predict_mean_variance(kernel, yres, Xq, pq)
    mo, vo = gp.predict_mean_variance(Xq)
    assert rel_err(mean2, mo) < TOL
    assert np.max(np.abs(var2 - vo)) < TOL * (np.max(np.abs(vo)) + 1.0)
    assert rel_err(chol.predict_covariance(kernel, Xq), gp.predict_covariance(Xq)) < 1e-8
    assert abs(chol.likelihood(kernel, yres, 0.07) / gp.likelihood() - 1.0) < TOL
    chol.free()


@pytest.mark.parametrize("n,m,d,ls", [(400, 16, 6, 0.6), (400, 12, 2, 0.3), (300, 150, 8, 0.9)])
def test_posterior_matches_oracle(ctx, n, m, d, ls):
    k = ("squared_exp", ls, 1.1)
    X, Xq = rand_inputs(n, d, 1), rand_inputs(m, d, 2)
    y = np.cos(X.sum(axis=1))
    gp = O.OracleGP(O.ZeroPrior(), k, 0.1, None, X, y)
    chol = ctx.cholesky_from_inputs(k, X, 0.1)
    mean, cov, cov_l = chol.posterior(k, y, Xq)
    mo, co, lo = gp.sample_at(Xq)
    cond = np.linalg.cond(co)
    assert rel_err(mean, mo) < TOL
    assert rel_err(cov, co) < TOL
    # the factor of the m x m posterior inherits cond(cov) as its error amplification
    assert rel_err(cov_l, lo) < max(TOL, 1e-13 * cond)
    # backward-stable factor of the lower triangle actually factored (cholesky() never reads the upper one)
    assert rel_err(np.tril(cov_l @ cov_l.T), np.tril(cov)) < 1e-13
    assert np.all(np.triu(cov_l, 1) == 0.0)
    z = np.linspace(-1, 1, m)
    assert rel_err(O.OracleGP.mvn_sample(mean, cov_l, z), O.OracleGP.mvn_sample(mo, lo, z)) < max(TOL, 1e-13 * cond)
    chol.free()


def test_posterior_failure_status(ctx):
    # a tanh "kernel" gives a robustly indefinite m x m posterior: MultivariateNormal::new panics in the
    # reference (multivariate_normal.rs:57); the ABI reports FR_NOT_POSITIVE_DEFINITE
    from friedrich_amd.device import FriedrichError
    k = ("hyper_tan", 1.0, 0.0)
    X = rand_inputs(50, 2, 3)
    Xq = np.asfortranarray(rand_inputs(8, 2, 4) * 3.0)
    y = np.zeros(50)
    chol = ctx.cholesky_from_inputs(k, X, 1.0)
    gp = O.OracleGP(O.ZeroPrior(), k, 1.0, None, X, y)
    with pytest.raises(FloatingPointError):
        gp.sample_at(Xq)
    with pytest.raises(FriedrichError) as e:
        chol.posterior(k, y, Xq)
    assert e.value.status == 1
    chol.free()


# ---- add_samples -----------------------------------------------------------------------------------
@pytest.mark.parametrize("n0,chunks", [(4, [4]), (100, [1, 27, 128]), (130, [200, 64]), (256, [256]),
                                       # >= 512 old rows: every append is the transposed forward solve -- one new row on the
                                       # single-column kernel (the Bayesian-optimisation loop), a few on K9, 17+ in column groups;
                                       # 1203 / 2049 old rows: a Schur product whose contraction has no divisor (ragged split-K slices)
                                       (640, [1, 1, 3, 16, 17, 130]), (1203, [1, 40]), (2049, [1, 2])])
def test_add_rows_matches_oracle_and_refit(ctx, n0, chunks):
    k = ("matern2", 0.8, 1.0)
    d = 3
    total = n0 + sum(chunks)
    Xall = rand_inputs(total, d, 77)
    noise = 0.1
    st, L_o, _ = O.make_cholesky_cov_matrix(k, Xall[:n0], noise)
    chol = ctx.cholesky_from_inputs(k, Xall[:n0], noise)  # capacity == n0: exercises EMatrix-style growth
    n = n0
    for c in chunks:
        L_o = O.add_rows_cholesky_cov_matrix(k, L_o, Xall[:n + c], c, noise)
        chol.add_rows(k, np.asfortranarray(Xall[:n + c]), c, noise)
        n += c
        assert chol.n == n
        assert rel_err(chol.l(), np.tril(L_o)) < TOL
    # equals a from-scratch factor of the grown set, and the solves still work (inverse blocks re-aligned)
    st, L_full, _ = O.make_cholesky_cov_matrix(k, Xall, noise)
    assert rel_err(chol.l(), np.tril(L_full)) < TOL
    B = np.asfortranarray(np.random.default_rng(5).standard_normal((total, 9)))
    assert rel_err(chol.solve(B), O.chol_solve(L_full, B)) < TOL
    chol.free()


def test_add_rows_has_no_epsilon_and_no_failure_check(ctx):
    # add_samples never applies cholesky_epsilon (mod.rs:185-189): a duplicated noiseless row yields a zero /
    # NaN pivot silently, exactly like Cholesky::insert_column's plain sqrt
    k = ("squared_exp", 1.0, 1.0)
    X = np.array([[0.0], [1.0], [2.0]])
    chol = ctx.cholesky_from_inputs(k, X, 0.0, eps=1e-6)
    Xall = np.asfortranarray(np.vstack([X, [[1.0]], [[3.0]]]))
    L_o = O.add_rows_cholesky_cov_matrix(k, O.make_cholesky_cov_matrix(k, X, 0.0, 1e-6)[1], Xall, 2, 0.0)
    chol.add_rows(k, Xall, 2, 0.0)
    L = chol.l()
    lo = np.tril(L_o)
    assert np.array_equal(np.isfinite(L), np.isfinite(lo))
    fin = np.isfinite(lo)
    assert rel_err(L[fin], lo[fin]) < 1e-6
    chol.free()


# ---- device-resident inputs (zero copy) --------------------------------------------------------------
def test_device_pointers(ctx):
    import torch
    k = ("squared_exp", 0.9, 1.0)
    n, m, d = 384, 50, 4
    X, Xq = rand_inputs(n, d, 21), rand_inputs(m, d, 22)
    y = np.sin(X.sum(axis=1))
    dev = torch.device("cuda:0")
    Xd = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev).t()      # column-major n x d on the GPU
    Xqd = torch.from_numpy(np.ascontiguousarray(Xq.T)).to(dev).t()
    yd = torch.from_numpy(y).to(dev)
    outd = torch.empty(m, dtype=torch.float64, device=dev)
    chol = ctx.cholesky_from_inputs(k, Xd, 0.1)
    chol.predict_mean(k, yd, Xqd, None, out=outd)
    ctx.synchronize()
    gp = O.OracleGP(O.ZeroPrior(), k, 0.1, None, X, y)
    assert rel_err(outd.cpu().numpy(), gp.predict(Xq)) < TOL
    chol.free()


# ---- size-independent properties at benchmark scale ---------------------------------------------------
def test_large_factor_properties(ctx):
    # N = 4096, d = 8 (BASELINE config 2): L L^T reproduces K, and solve() inverts K, without the O(n^3) oracle
    from friedrich_amd import synth
    n, d, m = 4096, 8, 256
    X, y, Xq = synth.make_problem(n, d, cfg=2, m=m)
    ls = ctx.mean_pairwise_distance(X)
    hp = synth.default_hyperparameters(X, y, ls)
    k = ("squared_exp", hp["ls"], hp["ampl"])
    chol = ctx.cholesky_from_inputs(k, X, hp["noise"])
    assert chol.info()["n_subst"] == 0
    L = chol.l()
    K = ctx.gram(k, X, X) + hp["noise"] ** 2 * np.eye(n)
    LLt = ctx.gemm(L, L, trans_b=True)
    assert rel_err(LLt, K) < 1e-12
    B = np.asfortranarray(np.random.default_rng(0).standard_normal((n, 8)))
    Z = chol.solve(B)
    assert rel_err(K @ Z, B) < 1e-9
    # predict_variance is non-negative and below the prior variance
    var = chol.predict_variance(k, Xq)
    assert np.all(var > -1e-10) and np.all(var <= hp["ampl"] + 1e-12)
    chol.free()


# ---- optimizer reductions ------------------------------------------------
@pytest.mark.parametrize("kernel", ALL_KERNELS, ids=lambda k: k[0] + str(len(k)))
def test_grad_terms_match_oracle(ctx, kernel):
    n, d = 300, 3
    X = rand_inputs(n, d, 31)
    y = np.sin(X.sum(axis=1))
    noise = 0.3
    # HyperTan / Multiquadric Gram matrices are indefinite on their own (the reference would substitute pivots, after
    # which K^-1 is not defined by the factor): their gradient bodies (kernel.rs:979-989, 1052-1059) are exercised inside
    # a positive-definite sum with a squared-exponential, so every leaf gradient is verified on the GPU
    gp = O.OracleGP(O.ZeroPrior(), kernel, noise, 1e-3, X, y)
    if len(gp.subst) > 0:
        kernel = ("sum", ("squared_exp", 0.6, 3.0), kernel)
        noise = 1.5
        gp = O.OracleGP(O.ZeroPrior(), kernel, noise, 1e-3, X, y)
    assert len(gp.subst) == 0
    chol = ctx.cholesky_from_inputs(kernel, X, noise, eps=1e-3)
    assert chol.info()["n_subst"] == 0
    npar = O.nb_parameters(kernel)
    g_o = gp.gradient()
    g, _ = chol.grad_terms(kernel, y, noise, scaled=False, nb_parameters=npar)
    assert g.shape == g_o.shape
    fin = np.isfinite(g_o)
    assert np.array_equal(np.isfinite(g), fin)
    assert np.max(np.abs(g[fin] - g_o[fin])) < 1e-8 * (np.max(np.abs(g_o[fin])) + 1.0)
    scale_o, gs_o = gp.scaled_gradient()
    gs, scale = chol.grad_terms(kernel, y, noise, scaled=True, nb_parameters=npar)
    assert abs(scale / scale_o - 1.0) < 1e-9
    fin = np.isfinite(gs_o)
    assert np.array_equal(np.isfinite(gs), fin)
    assert np.max(np.abs(gs[fin] - gs_o[fin])) < 1e-8 * (np.max(np.abs(gs_o[fin])) + 1.0)
    chol.free()


@pytest.mark.parametrize("ls", [-0.7, -2e-3], ids=["finite", "overflowing"])
def test_grad_terms_matern2_negative_length_scale(ctx, ls):
    """The Matern-5/2 gradient keeps the reference's SIGNED length scale in its exponent (kernel.rs:881-900): a negative ls
    evaluates exp(+x).  ls = -0.7: finite values through the positive-argument branch of the in-tree exp; ls = -2e-3:
    exp(+x) overflows -- the reference gets +inf there (the gradient is then inf / NaN), and so must the device, entry for
    entry (round 3's exp returned NaN for an infinite argument where libm returns +inf)."""
    n, d = 200, 2
    X = rand_inputs(n, d, 5)
    y = np.sin(X.sum(axis=1))
    kernel, noise = ("matern2", ls, 1.2), 0.3
    gp = O.OracleGP(O.ZeroPrior(), kernel, noise, None, X, y)
    chol = ctx.cholesky_from_inputs(kernel, X, noise)
    assert rel_err(chol.l(), np.tril(gp.L)) < 1e-9  # (the kernel itself takes |ls|: kernel.rs:873)
    g_o = gp.gradient()
    g, _ = chol.grad_terms(kernel, y, noise, scaled=False, nb_parameters=2)
    fin = np.isfinite(g_o)
    assert np.array_equal(np.isfinite(g), fin), (g, g_o)
    # (where exp(+x) overflows both gradients are non-finite; WHICH non-finite value -- inf - inf = NaN in the reference's
    # association, +-inf in the fused reduction's -- depends on the order of the sums and is not compared)
    if fin.any():
        assert np.max(np.abs(g[fin] - g_o[fin])) < 1e-8 * (np.max(np.abs(g_o[fin])) + 1.0)
    chol.free()


@pytest.mark.parametrize("n", [2700, 4096, 5200])  # (4096: the halving ends in whole 2048-row blocks, whose cached inverses are copied)
def test_grad_terms_triangular_inverse_path(ctx, n):
    """Above 2048 rows the gradient's K^-1 = W^T W comes from the recursive triangular inverse and products that skip the
    structural zeros (chol.hip: chol_tri_inverse; GemmArgs::tri, tiles claimed): against the oracle's gradient
    (optimizer.rs:24-60, 159-203) at a ragged size, and against the dense products (option tri_inverse = 0) beyond."""
    kernel = ("squared_exp", 0.9, 1.7)
    X = rand_inputs(n, 3, 77)
    y = np.sin(X.sum(axis=1))
    noise = 0.4
    chol = ctx.cholesky_from_inputs(kernel, X, noise)
    g, _ = chol.grad_terms(kernel, y, noise, scaled=False, nb_parameters=2)
    gs, scale = chol.grad_terms(kernel, y, noise, scaled=True, nb_parameters=2)
    ctx.set_option("tri_inverse", 0)
    try:
        g0, _ = chol.grad_terms(kernel, y, noise, scaled=False, nb_parameters=2)
        gs0, scale0 = chol.grad_terms(kernel, y, noise, scaled=True, nb_parameters=2)
    finally:
        ctx.set_option("tri_inverse", 1)
    assert np.max(np.abs(g - g0)) < 1e-10 * (np.max(np.abs(g0)) + 1.0) and np.max(np.abs(gs - gs0)) < 1e-10 * (np.max(np.abs(gs0)) + 1.0)
    assert abs(scale / scale0 - 1.0) < 1e-12
    # Cholesky::inverse goes the same way above 2048 rows
    Ki = chol.inverse()
    ctx.set_option("tri_inverse", 0)
    try:
        Ki0 = chol.inverse()
    finally:
        ctx.set_option("tri_inverse", 1)
    assert rel_err(Ki, Ki0) < 1e-11 and np.array_equal(Ki, Ki.T)
    if n <= 3000:
        with O.threads(0):
            gp = O.OracleGP(O.ZeroPrior(), kernel, noise, None, X, y)
            g_o = gp.gradient()
            scale_o, gs_o = gp.scaled_gradient()
        assert np.max(np.abs(g - g_o)) < 1e-8 * (np.max(np.abs(g_o)) + 1.0)
        assert abs(scale / scale_o - 1.0) < 1e-9 and np.max(np.abs(gs - gs_o)) < 1e-8 * (np.max(np.abs(gs_o)) + 1.0)
    chol.free()


def test_concurrent_predict_from_threads(ctx):
    # GaussianProcess is Send + Sync in the reference: several threads may call &self methods of one model at once.
    # The entry points of a context take turns (per-context lock); results must equal the sequential ones.
    import threading

    n, m, d = 900, 400, 4
    kernel = PD_KERNELS[0]
    X = rand_inputs(n, d, 41)
    y = np.cos(X.sum(axis=1))
    chol = ctx.cholesky_from_inputs(kernel, X, 0.1)
    queries = [rand_inputs(m, d, 500 + i) for i in range(4)]
    expect = [(chol.predict_mean(kernel, y, q, None), chol.predict_variance(kernel, q)) for q in queries]
    got = [None] * len(queries)
    errors = []

    def worker(i):
        try:
            for _ in range(5):
                got[i] = (chol.predict_mean(kernel, y, queries[i], None), chol.predict_variance(kernel, queries[i]))
        except BaseException as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(len(queries))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not errors, errors
    for (m0, v0), (m1, v1) in zip(expect, got):
        assert np.array_equal(m0, m1) and np.array_equal(v0, v1)
    chol.free()


# ---- cached alpha = K^-1 y (the reference's todo.md:10) ---------------------------------------
def test_cached_alpha_interleaved_with_add_samples_and_refit(ctx):
    """fr_chol_set_targets + predict with y = NULL: the cache is invalidated by add_samples (row count changes: targets have
    to be handed over again) and by a re-fit (alpha re-solved lazily), and every prediction equals the oracle's predict of
    the reference association.  Also the m = 1 persistent solves against the recursive path they replace."""
    from friedrich_amd.device import FriedrichError

    k = ("matern2", 0.8, 1.1)
    d, n0 = 4, 700
    chunks = [1, 130, 300]
    total = n0 + sum(chunks)
    Xall = rand_inputs(total, d, 91)
    yall = np.sin(Xall.sum(axis=1)) + 0.2
    Xq = rand_inputs(40, d, 92)
    prior = O.ConstantPrior(0.1)
    noise = 0.09
    gp = O.OracleGP(prior, k, noise, None, Xall[:n0], yall[:n0])
    chol = ctx.cholesky_from_inputs(k, Xall[:n0], noise)  # capacity == n0: growth reallocates the cache as well
    pq = prior.prior(Xq)
    with pytest.raises(FriedrichError):  # nothing cached yet
        chol.predict_mean(k, None, Xq, pq)
    chol.set_targets(gp.y)
    assert rel_err(chol.predict_mean(k, None, Xq, pq), gp.predict(Xq)) < TOL
    n = n0
    for c in chunks:
        gp.add_samples(Xall[n:n + c], yall[n:n + c])
        chol.add_rows(k, np.asfortranarray(Xall[:n + c]), c, noise)
        n += c
        with pytest.raises(FriedrichError):  # the cached targets cover the old rows only
            chol.predict_mean(k, None, Xq, pq)
        chol.set_targets(gp.y)
        assert rel_err(chol.predict_mean(k, None, Xq, pq), gp.predict(Xq)) < TOL
        assert rel_err(chol.predict_mean(k, None, Xq, pq), chol.predict_mean(k, gp.y, Xq, pq)) < TOL
    # re-fit with other hyper-parameters: same targets, alpha must follow the new factor
    k2 = ("matern2", 0.6, 0.9)
    gp2 = O.OracleGP(prior, k2, 0.12, None, Xall, yall)
    chol.refactor(k2, 0.12)
    assert rel_err(chol.predict_mean(k2, None, Xq, pq), gp2.predict(Xq)) < TOL
    chol.free()


@pytest.mark.parametrize("n", [1, 2, 127, 128, 129, 300, 1000, 2049])
def test_single_rhs_persistent_solves(ctx, n):
    """m = 1: one persistent launch per direction (trsv.hip) vs the oracle, vs the recursive path (option trsv = 0), for
    sizes around the 128-row block boundaries (partial last block, single block)."""
    k = PD_KERNELS[0]
    X = rand_inputs(n, 3, 1000 + n)
    b = np.random.default_rng(n).standard_normal((n, 1))
    st, L_o, _ = O.make_cholesky_cov_matrix(k, X, 0.1)
    chol = ctx.cholesky_from_inputs(k, X, 0.1)
    z1, w1 = chol.solve(b), chol.solve_lower(b)
    assert rel_err(z1, O.chol_solve(L_o, b)) < TOL
    assert rel_err(w1, O.solve_lower(L_o, b)[1]) < TOL
    ctx.set_option("trsv", 0)
    try:
        assert rel_err(chol.solve(b), z1) < 1e-11 and rel_err(chol.solve_lower(b), w1) < 1e-11
    finally:
        ctx.set_option("trsv", 1)
    # repeated calls are deterministic bit for bit (fixed summation order, hand-off granules re-initialised per call)
    assert np.array_equal(chol.solve(b), z1)
    chol.free()


def test_refinement_policy_and_forced_modes(ctx):
    """option "refine": well-conditioned fits never refine (no second factorisation, the fast kernels stay in use), the
    forced modes give the same factor to round-off, and an ill-conditioned handle refines its solves as well."""
    k = PD_KERNELS[0]
    X = rand_inputs(900, 3, 5)
    st, L_o, _ = O.make_cholesky_cov_matrix(k, X, 0.1)
    B = np.asfortranarray(np.random.default_rng(2).standard_normal((900, 5)))
    Ls = []
    for mode in (-1, 0, 1):
        ctx.set_option("refine", mode)
        chol = ctx.cholesky_from_inputs(k, X, 0.1)
        est, refined = chol.conditioning()
        assert refined == (mode == 1) and 1.0 <= est < 30.0
        assert rel_err(chol.l(), np.tril(L_o)) < TOL
        assert rel_err(chol.solve(B), O.chol_solve(L_o, B)) < TOL
        assert rel_err(chol.solve_lower(B[:, :1]), O.solve_lower(L_o, B[:, :1])[1]) < TOL
        Ls.append(chol.l())
        chol.free()
    ctx.set_option("refine", -1)
    assert rel_err(Ls[0], Ls[2]) < 1e-13 and np.array_equal(Ls[0], Ls[1])


def test_conditioning_tracked_through_add_rows_from_matrix_and_upload(ctx):
    """The conditioning estimate and the refinement decision follow EVERY way a factor comes into being or changes (round-2
    advisor: only fr_chol_from_inputs / refactor refreshed them).  Fixture: RBF, d = 1, length scale 0.05, sorted points --
    every 128 x 128 diagonal block is ill-conditioned (estimates ~1e3 at noise 1e-3).
      * add_rows: a well-conditioned factor (spread points, noise 0.3) takes 300 ill-conditioned rows: the appended blocks'
        estimates raise max_estimate above the threshold, the append is repeated with refinement, the handle refines from
        then on, and the grown factor matches the oracle's to the conditioning-limited accuracy of the refined path;
      * fr_chol_from_matrix on the explicit covariance matrix and fr_chol_upload_l of the oracle's factor: both report the
        estimate and switch refinement on; with option refine = 0 nothing refines and the estimate is still reported."""
    rng = np.random.default_rng(11)
    k = ("squared_exp", 0.05, 1.0)
    noise = 1e-3
    n0, n1 = 400, 300
    X = np.asfortranarray(np.concatenate([5.0 + 0.1 * np.arange(n0).reshape(-1, 1), np.sort(rng.random((n1, 1)), axis=0)]))  # a grid two length scales apart, then a dense cluster
    st, L_o, _ = O.make_cholesky_cov_matrix(k, X, noise)
    assert st == 0
    K = O.make_covariance_matrix(k, X, X) + noise * noise * np.eye(n0 + n1)
    ku = np.linalg.cond(K) * 2.2e-16
    chol = ctx.cholesky_from_inputs(k, X[:n0], noise, capacity_hint=n0 + n1)
    est0, ref0 = chol.conditioning()
    assert not ref0 and est0 < 30.0  # grid points two length scales apart: nearly diagonal
    chol.add_rows(k, X, n1, noise)
    est1, ref1 = chol.conditioning()
    assert ref1 and est1 > 100.0, (est1, ref1)
    assert rel_err(chol.l(), np.tril(L_o)) < max(1e-9, 0.1 * ku)
    B = np.asfortranarray(rng.standard_normal((n0 + n1, 3)))
    assert rel_err(chol.solve(B), O.chol_solve(L_o, B)) < max(1e-9, 2.0 * ku)
    chol.free()
    for mode in (-1, 0):
        ctx.set_option("refine", mode)
        try:
            c2 = ctx.cholesky_from_matrix(K)
            est2, ref2 = c2.conditioning()
            assert est2 > 100.0 and ref2 == (mode == -1)
            if mode == -1:
                assert rel_err(c2.l(), np.tril(L_o)) < max(1e-9, 0.1 * ku)
            c2.free()
            c3 = ctx.cholesky_upload(np.asfortranarray(np.tril(L_o)), X)
            est3, ref3 = c3.conditioning()
            assert est3 > 100.0 and ref3 == (mode == -1)
            assert rel_err(c3.solve(B), O.chol_solve(L_o, B)) < (max(1e-9, 2.0 * ku) if mode == -1 else 1e-4)
            c3.free()
        finally:
            ctx.set_option("refine", -1)


def test_roctx_ranges_do_not_disturb_a_run():
    """FRIEDRICH_AMD_ROCTX=1: every entry point opens a roctx range (marker library dlopen'ed on first use); a fit + predict in
    a fresh process behaves exactly as without it (the ranges only show in a tool: rocprofv3 --marker-trace)."""
    import os
    import subprocess
    import sys

    from conftest import ROOT

    code = ("import numpy as np, sys; sys.path.insert(0, %r); from friedrich_amd.device import Context; c = Context(); "
            "X = np.asfortranarray(np.random.default_rng(0).random((300, 3))); ch = c.cholesky_from_inputs(('squared_exp', 0.8, 1.3), X, 0.1); "
            "print(float(ch.predict_variance(('squared_exp', 0.8, 1.3), X[:2])[0])); ch.free(); c.close()") % ROOT
    outs = []
    for flag in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, FRIEDRICH_AMD_ROCTX=flag), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-1500:]
        outs.append(r.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1]


def test_degenerate_sizes(ctx):
    """One training row, one query, no query at all, a row-append of zero rows, a single right-hand side of length 1: the
    launches that would be empty are skipped, the answers are the oracle's (mod.rs:234-241, 260-263; algebra/mod.rs:97-126)."""
    k = PD_KERNELS[0]
    X = rand_inputs(3, 2, 11)
    y = np.array([0.3, -1.2, 0.8])
    for n in (1, 2, 3):
        st, L_o, _ = O.make_cholesky_cov_matrix(k, X[:n], 0.2)
        chol = ctx.cholesky_from_inputs(k, np.asfortranarray(X[:n]), 0.2, capacity_hint=4)
        assert rel_err(chol.l(), np.tril(L_o)) < TOL
        Xq = rand_inputs(2, 2, 12)
        gp = O.OracleGP(O.ConstantPrior(0.0), k, 0.2, None, X[:n], y[:n])
        assert rel_err(chol.predict_mean(k, y[:n], Xq), gp.predict(Xq)) < TOL
        assert rel_err(chol.predict_variance(k, Xq[:1]), gp.predict_variance(Xq[:1])) < TOL
        assert chol.predict_mean(k, y[:n], Xq[:0]).shape == (0,) and chol.predict_variance(k, Xq[:0]).shape == (0,)
        chol.add_rows(k, np.asfortranarray(X[:n]), 0, 0.2)  # nothing to add
        assert chol.info()["n"] == n and rel_err(chol.l(), np.tril(L_o)) < TOL
        b = np.asfortranarray(np.arange(1.0, n + 1.0).reshape(n, 1))
        assert rel_err(chol.solve(b), O.chol_solve(L_o, b)) < TOL
        if n < 3:
            chol.add_rows(k, np.asfortranarray(X[:n + 1]), 1, 0.2)
            st2, L2, _ = O.make_cholesky_cov_matrix(k, X[:n + 1], 0.2)
            assert rel_err(chol.l(), np.tril(L2)) < TOL
        chol.free()


@pytest.mark.parametrize("opt,val", [("lookahead", 0), ("nb", 1024)])
def test_schedule_options_keep_the_factor(ctx, opt, val):
    """Without look-ahead, and with 1024-column panels that switch to 512 columns for the tail (the large-N schedule forced
    onto a small matrix): the oracle's factor, the same conditioning estimate as the default path."""
    k = PD_KERNELS[0]
    n = 2700
    X = rand_inputs(n, 3, 4242)
    st, L_o, _ = O.make_cholesky_cov_matrix(k, X, 0.1)
    chol = ctx.cholesky_from_inputs(k, X, 0.1)
    L_d = chol.l()
    assert rel_err(L_d, np.tril(L_o)) < TOL
    est_d = chol.conditioning()
    ctx.set_option(opt, val)
    if opt == "nb":
        ctx.set_option("nb_switch_rows", 1700)  # 1024-column panels, then 512-column ones (the tail of a large fit)
    try:
        for rep in range(2):
            chol.refactor(k, 0.1)
            assert rel_err(chol.l(), L_d) < 1e-12
            est = chol.conditioning()
            assert est[1] == est_d[1] and abs(est[0] / est_d[0] - 1.0) < 1e-9
    finally:
        ctx.set_option(opt, 1 if opt == "lookahead" else 0)
        ctx.set_option("nb_switch_rows", 16384)
    chol.free()


@pytest.mark.parametrize("n", [1536, 2500, 3200])
def test_xcd_reservation(ctx, n):
    """The look-ahead pipeline with XCDs set aside for the panel chain (gemm_f64.hip: trailing-update tiles CLAIMED by the
    workgroups that do not sit on the panel stream's XCDs, panel launches on the b % 8 < R workgroups): with the
    diagonal-block kernel held fixed (option k4_flat = 0 / 1: round 5's flat variant and the staged one round differently),
    every reservation setting gives bit for bit the same factor -- same arithmetic, only placed differently -- and the
    automatic choice (flat where the kernel has its CU to itself) is a function of the input: two fits agree bit for bit."""
    k = PD_KERNELS[0]
    X = rand_inputs(n, 3, 77 + n)
    st, L_o, _ = O.make_cholesky_cov_matrix(k, X, 0.1)
    L_o = np.tril(L_o)
    chol = ctx.cholesky_from_inputs(k, X, 0.1)
    L_auto = chol.l()
    assert rel_err(L_auto, L_o) < TOL
    chol.refactor(k, 0.1)
    assert np.array_equal(chol.l(), L_auto)
    try:
        for flat in (0, 1):
            ctx.set_option("k4_flat", flat)
            ctx.set_option("xcd_reserve", -1)
            chol.refactor(k, 0.1)
            L_ref = chol.l()
            assert rel_err(L_ref, L_o) < TOL
            for r1 in (0, 1, 2, 3, 4):
                ctx.set_option("xcd_reserve", r1)
                for rep in range(2):  # (twice: the claim counters are recycled, the published XCD is known the second time)
                    chol.refactor(k, 0.1)
                    assert np.array_equal(chol.l(), L_ref), (flat, r1)
    finally:
        ctx.set_option("xcd_reserve", -1)
        ctx.set_option("k4_flat", -1)
    chol.free()


@pytest.mark.parametrize("n", [5300, 6700])
def test_cu_level_reservation_same_bits(ctx, n):
    """Above 4096 trailing rows the reservation is carried out by CUs (option cu_reserve, the default): the trailing update and the
    early look-ahead update run as RESIDENT workgroups that claim their tiles and vacate R CUs of every shader engine
    (syrk_lower_persist_f64_kernel / gemm_f64_persist_kernel), the panel stream's launches carry no idle workgroups.  Every
    tile is still computed whole by one workgroup: the factor is bit for bit the one of the XCD-level reservation, for every
    tier setting, twice in a row (the claim counters are recycled) -- and both agree with the oracle."""
    k = PD_KERNELS[0]
    X = rand_inputs(n, 3, 99 + n)
    st, L_o, _ = O.make_cholesky_cov_matrix(k, X, 0.1)
    chol = ctx.cholesky_from_inputs(k, X, 0.1)
    try:
        # panel_chain = 0: the launch chain of round 5 -- the property above, bit for bit.  panel_chain = 2 (round 6, the default): by CUs
        # the rows below a resident panel's diagonal block take the one-launch row solve (left-looking: one accumulation per sub-panel)
        # and look-ahead + trailing update are one launch; by XCDs they stay inside the resident launch (right-looking): the same
        # products summed in a different order -- equal to round-off, and each setting reproducible bit for bit
        for chain in (0, 2):
            ctx.set_option("panel_chain", chain)
            chol.refactor(k, 0.1)
            L_cu = chol.l()
            assert rel_err(L_cu, np.tril(L_o)) < TOL
            for opts in ({"cu_reserve": 0}, {"cu_reserve": 1, "reserve_rows2_cu": 8192}, {"cu_reserve": 1, "cu_reserve_min_rows": 1024},
                         {"cu_reserve": 1, "xcd_reserve": 1}, {"cu_reserve": 1, "xcd_reserve": 4}, {"cu_reserve": 1}):
                for o, v in {"cu_reserve": 1, "reserve_rows2_cu": 6144, "reserve_rows1_cu": 12288, "cu_reserve_min_rows": 4096, "xcd_reserve": -1, **opts}.items():
                    ctx.set_option(o, v)
                first = None
                for rep in range(2):
                    chol.refactor(k, 0.1)
                    L = chol.l()
                    if chain == 0:
                        assert np.array_equal(L, L_cu), opts
                    else:
                        assert rel_err(L, L_cu) < 1e-12, opts
                        assert first is None or np.array_equal(L, first), opts
                        first = L
    finally:
        for o, v in {"cu_reserve": 1, "reserve_rows2_cu": 6144, "cu_reserve_min_rows": 4096, "xcd_reserve": -1, "panel_chain": 2}.items():
            ctx.set_option(o, v)
    chol.free()


@pytest.mark.parametrize("n,m", [(2, 2), (127, 3), (128, 16), (129, 7), (300, 2), (1000, 16), (2049, 5), (700, 17), (1300, 100), (900, 300)])
def test_narrow_persistent_solves_2_to_16_columns(ctx, n, m):
    """2 .. 16 right-hand sides (and more, in column groups of 16): one persistent matrix-core launch per direction (trsm_narrow.hip), the backward sweep on
    the transposed copy kept in the strict upper triangle -- vs the oracle, vs the recursive path (option trsv = 0), after
    add_rows (copy rebuilt) and with the factor's download unaffected by the copy."""
    k = PD_KERNELS[1]
    X = rand_inputs(n + 40, 3, 2000 + n)
    B = np.asfortranarray(np.random.default_rng(n + m).standard_normal((n + 40, m)))
    st, L_o, _ = O.make_cholesky_cov_matrix(k, X[:n], 0.1)
    chol = ctx.cholesky_from_inputs(k, X[:n], 0.1)
    z, w_ = chol.solve(B[:n]), chol.solve_lower(B[:n])
    assert rel_err(z, O.chol_solve(L_o, B[:n])) < TOL
    assert rel_err(w_, O.solve_lower(L_o, B[:n])[1]) < TOL
    ctx.set_option("trsv", 0)
    try:
        assert rel_err(chol.solve(B[:n]), z) < 1e-11
    finally:
        ctx.set_option("trsv", 1)
    assert np.array_equal(chol.solve(B[:n]), z)  # deterministic
    L = chol.l()
    assert rel_err(L, np.tril(L_o)) < TOL and np.all(np.triu(L, 1) == 0.0)
    chol.add_rows(k, np.asfortranarray(X), 40, 0.1)
    st, L1, _ = O.make_cholesky_cov_matrix(k, X, 0.1)
    assert rel_err(chol.solve(B), O.chol_solve(L1, B)) < TOL
    chol.free()


def test_narrow_solves_follow_a_refactored_factor(ctx):
    """The single-group kernel keeps per-factor caches (chain products W_b L[b, b-1], the transposed copy): a refactor with
    other hyper-parameters, an upload and add_rows must each invalidate them -- 16 right-hand sides vs the oracle after every change."""
    n, m = 1000, 16
    X = rand_inputs(n + 24, 4, 77)
    B = np.asfortranarray(np.random.default_rng(78).standard_normal((n + 24, m)))
    k1, k2 = PD_KERNELS[1], PD_KERNELS[0]
    chol = ctx.cholesky_from_inputs(k1, X[:n], 0.2)
    st, L1, _ = O.make_cholesky_cov_matrix(k1, X[:n], 0.2)
    assert rel_err(chol.solve(B[:n]), O.chol_solve(L1, B[:n])) < TOL
    chol.refactor(k2, 0.35)
    st, L2, _ = O.make_cholesky_cov_matrix(k2, X[:n], 0.35)
    assert rel_err(chol.solve(B[:n]), O.chol_solve(L2, B[:n])) < TOL
    assert rel_err(chol.solve_lower(B[:n]), O.solve_lower(L2, B[:n])[1]) < TOL
    up = ctx.cholesky_upload(np.tril(L1), X[:n])
    assert rel_err(up.solve(B[:n]), O.chol_solve(L1, B[:n])) < TOL
    up.free()
    chol.add_rows(k2, np.asfortranarray(X), 24, 0.35)
    st, L3, _ = O.make_cholesky_cov_matrix(k2, X, 0.35)
    assert rel_err(chol.solve(B), O.chol_solve(L3, B)) < TOL
    chol.free()


# ---- `Input` staging (conversion/mod.rs:58-201) ---------------------------------------------
@pytest.mark.parametrize("n,d", [(1, 1), (1, 5), (333, 1), (1000, 7), (4100, 16)])
def test_inputs_to_device_layouts(ctx, n, d):
    """fr_inputs_to_device: Vec<Vec<f64>> (row pointers), row-major ndarray, column-major DMatrix -> the same device matrix;
    a Gram matrix / factor / predict built from the staged inputs equals the oracle's on the host inputs."""
    rng = np.random.default_rng(n * 31 + d)
    Xr = np.ascontiguousarray(rng.random((n, d)))  # row-major samples
    k = ("matern2", 0.9, 1.1)
    want = O.make_covariance_matrix(k, Xr, Xr[: min(n, 50)])
    staged = [ctx.inputs_to_device(Xr, "rowmajor"), ctx.inputs_to_device(np.asfortranarray(Xr), "colmajor"),
              ctx.inputs_to_device([Xr[i] for i in range(n)], "rowptrs")]
    head = ctx.inputs_to_device(Xr[: min(n, 50)], "rowmajor")
    for s_ in staged:
        assert (s_.rows, s_.cols) == (n, d) and s_.ld >= n and s_.ld % 64 == 0
        assert rel_err(ctx.gram(k, s_, head), want) < TOL_GRAM
    if n >= 300:
        gp = O.OracleGP(O.ZeroPrior(), k, 0.1, None, Xr, np.sin(Xr.sum(axis=1)))
        chol = ctx.cholesky_from_inputs(k, staged[0], 0.1)
        assert rel_err(chol.l(), np.tril(gp.L)) < TOL
        assert rel_err(chol.predict_mean(k, gp.y, head, None), gp.predict(Xr[: min(n, 50)])) < TOL
        chol.free()
    for s_ in staged + [head]:
        s_.free()


# ---- LinearPrior::fit (prior.rs:139-159) ------------------------------------------------------
@pytest.mark.parametrize("n,d", [(4, 1), (5, 3), (300, 2), (257, 16), (5000, 8), (40000, 16)])
def test_linear_prior_fit_matches_svd_least_squares(ctx, n, d):
    """device TSQR + small SVD vs the oracle's restatement of the reference (nalgebra's SVD solve with eps = 0), including an
    exactly rank-deficient design."""
    rng = np.random.default_rng(n + d)
    X = np.asfortranarray(rng.random((n, d)) * 3.0 - 1.0)
    y = X @ rng.standard_normal(d) + 0.7 + 0.05 * rng.standard_normal(n)
    prior = O.LinearPrior.default(d)
    prior.fit(X, y)
    w, b = ctx.linear_prior_fit(X, y)
    assert abs(b - prior.intercept) < 1e-9 * (1.0 + abs(prior.intercept))
    assert rel_err(w, prior.weights) < 1e-9
    if d >= 2 and n > d + 2:
        # a feature that is identically zero: its singular value is EXACTLY zero, the one case the reference's eps = 0
        # treats as rank deficient (the direction is dropped).  (A duplicated feature has a singular value of rounding size,
        # which nalgebra inverts: the reference's result is then rounding noise, nothing to compare with.)
        X2 = np.asfortranarray(np.hstack([X, np.zeros((n, 1))]))
        p2 = O.LinearPrior.default(d + 1)
        p2.fit(X2, y)
        w2, b2 = ctx.linear_prior_fit(X2, y)
        assert w2[-1] == 0.0 and abs(p2.weights[-1]) < 1e-12
        assert rel_err(w2, p2.weights) < 1e-9 and abs(b2 - p2.intercept) < 1e-9
