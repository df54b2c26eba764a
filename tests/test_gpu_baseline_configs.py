"""BASELINE.json's full-size configurations, checked through size-independent properties on the GPU (the O(n^3) oracle
finishes these sizes in hours, not seconds): L L^T reproduces K, solve() inverts K, add_samples in chunks lands on the
same factor as a from-scratch fit, the posterior is consistent with predict.  Everything stays resident in HBM."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _col_major(torch, a, dev):
    """host (n x m) array -> column-major device tensor"""
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64).T)).to(dev).t()


def _empty_cm(torch, n, m, dev):
    return torch.empty((m, n), dtype=torch.float64, device=dev).t()


@pytest.mark.parametrize("name,n,d,kernel_name,eps", [
    ("config2_matern52_eps", 16384, 16, "matern2", 1e-9),   # BASELINE configs[2]
    ("config3_rbf", 32768, 16, "squared_exp", None),          # BASELINE configs[3] (the bench workload)
    ("between_the_panel_tiers", 20480, 16, "squared_exp", None),  # 1024-column panels for 4096 rows, then 512 (pick_nb's threshold moved to 18432 in round 5)
    ("first_size_of_the_2048_tier", 23552, 16, "squared_exp", None),  # one 2048-column panel (23552 > 22528 rows), then 1024, then 512
    ("twice_config3", 65536, 16, "squared_exp", None),         # 32 GiB factor: 64-bit indexing, sized for 288 GB of HBM
])
def test_full_size_fit_properties(ctx, name, n, d, kernel_name, eps):
    torch = pytest.importorskip("torch")
    from friedrich_amd import synth

    dev = torch.device("cuda", 0)
    m = 512
    X, y, Xq = synth.make_problem(n, d, cfg=3, m=m)
    ls = ctx.mean_pairwise_distance(X)
    hp = synth.default_hyperparameters(X, y, ls)
    k = (kernel_name, hp["ls"], hp["ampl"])
    noise = hp["noise"]
    X_d = _col_major(torch, X, dev)
    torch.cuda.synchronize()
    chol = ctx.cholesky_from_inputs(k, X_d, noise, eps=eps, capacity_hint=n)
    info = chol.info()
    assert info["fail_col"] == -1 and info["n_subst"] == 0

    # K Z = V for Z = K^-1 V (two blocked triangular solves), K assembled once more, independently of the factor
    K = _empty_cm(torch, n, n, dev)
    ctx.gram(k, X_d, X_d, out=K)
    ctx.synchronize()  # the library launches on its own streams: order them against torch's by hand
    K.diagonal().add_(noise * noise)
    nprobe = 8
    V = _col_major(torch, np.random.default_rng(1).standard_normal((n, nprobe)), dev)
    Z = V.t().clone().t()  # column-major copy; device tensors are solved in place
    torch.cuda.synchronize()
    chol.solve(Z)
    KZ = _empty_cm(torch, n, nprobe, dev)
    ctx.gemm(K, Z, C=KZ)
    ctx.synchronize()
    err = float(torch.linalg.norm(KZ - V) / torch.linalg.norm(V))
    assert err < 1e-8, err  # north_star: <= 1e-8 relative error

    # the three solve paths agree at full size: 8 columns went through the matrix-core persistent kernel (K9: one product per
    # chain step, payload polling; more blocks than CUs at N = 65536); column 0 alone takes the single-column kernels (K8), 40
    # columns the column groups of the half-tile kernel, and with option trsv = 0 the recursion of GEMM launches
    Z1 = V[:, :1].t().clone().t()
    W40 = _col_major(torch, np.random.default_rng(3).standard_normal((n, 40)), dev)
    W40[:, :nprobe] = V
    Z40 = W40.t().clone().t()
    Zr = V.t().clone().t()
    torch.cuda.synchronize()
    chol.solve(Z1)
    chol.solve(Z40)
    ctx.set_option("trsv", 0)
    try:
        chol.solve(Zr)
    finally:
        ctx.set_option("trsv", 1)
    ctx.synchronize()
    zs = float(torch.max(torch.abs(Z)))
    assert float(torch.max(torch.abs(Z1 - Z[:, :1]))) / zs < 1e-10
    assert float(torch.max(torch.abs(Z40[:, :nprobe] - Z))) / zs < 1e-10
    assert float(torch.max(torch.abs(Zr - Z))) / zs < 1e-10
    assert ctx.counter("solve_retries") == 0

    # the two halves of solve() agree with each other: with U = L^-1 V,  v^T K^-1 v = u^T u  for every probe column
    U = V.t().clone().t()
    torch.cuda.synchronize()
    chol.solve_lower(U)
    ctx.synchronize()
    lhs = (V * Z).sum(dim=0)
    rhs = (U * U).sum(dim=0)
    assert float(torch.max(torch.abs(lhs - rhs) / torch.abs(rhs))) < 1e-9

    # predict_variance is non-negative and below the prior variance; predict stays finite
    var = chol.predict_variance(k, Xq)
    mean = chol.predict_mean(k, y - hp["prior"], Xq, np.full(m, hp["prior"]))
    assert np.all(np.isfinite(mean))
    assert np.all(var > -1e-9) and np.all(var <= hp["ampl"] + 1e-9)
    chol.free()


def test_config4_add_samples_in_chunks_and_sample_at(ctx):
    """BASELINE configs[4]: N = 8192, d = 8 RBF grown by add_samples in 512-row chunks (rank-k Cholesky update), then
    sample_at.  Property: the grown factor equals the from-scratch factor of the same rows (same pivot rule away from
    breakdown), and the posterior of sample_at is consistent with predict / predict_covariance."""
    from friedrich_amd import synth

    n, d, chunk, m = 8192, 8, 512, 64
    X, y, Xq = synth.make_problem(n, d, cfg=4, m=m)
    ls = ctx.mean_pairwise_distance(X)
    hp = synth.default_hyperparameters(X, y, ls)
    k = ("squared_exp", hp["ls"], hp["ampl"])
    noise = hp["noise"]
    grown = ctx.cholesky_from_inputs(k, X[:chunk], noise, capacity_hint=n)
    for hi in range(2 * chunk, n + 1, chunk):
        grown.add_rows(k, X[:hi], chunk, noise)
    assert grown.n == n
    direct = ctx.cholesky_from_inputs(k, X, noise, capacity_hint=n)
    Lg, Ld = grown.l(), direct.l()
    assert rel_err(Lg, Ld) < 1e-9
    # solves through the rebuilt inverse blocks agree as well
    B = np.asfortranarray(np.random.default_rng(2).standard_normal((n, 4)))
    assert rel_err(grown.solve(B), direct.solve(B)) < 1e-8
    # sample_at: posterior mean = predict, covariance = predict_covariance, cov_l cov_l^T = covariance
    prior_q = np.full(m, hp["prior"])
    mean, cov, cov_l = grown.posterior(k, y - hp["prior"], Xq, prior_q)
    assert rel_err(mean, grown.predict_mean(k, y - hp["prior"], Xq, prior_q)) < 1e-12
    assert rel_err(np.tril(cov), np.tril(grown.predict_covariance(k, Xq))) < 1e-9  # k** - (L^-1 K*)^T (L^-1 K*): cancellation
    assert rel_err(np.tril(cov_l @ cov_l.T), np.tril(cov)) < 1e-9
    grown.free()
    direct.free()
