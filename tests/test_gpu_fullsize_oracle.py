"""BASELINE.json's configurations compared DIRECTLY with the oracle at full size (round-1 verdict, "parity" item 2).

The oracle's factorisation runs here with its blocked multi-thread schedule (oracle/friedrich_oracle.c,
fro_cholesky_cols_mt: every element still sees the reference's k-ascending unfused multiply-adds, so it is bit-identical
to the single-thread restatement -- asserted in tests/test_oracle.py) and the solves with one host thread per
right-hand side.  That brings N = 4096 / 8192 down to seconds and the leading column blocks of N = 16384 / 32768 within
reach (column j of the left-looking factor depends only on columns < j).

Tolerance: north_star allows 1e-8 relative; the tests hold TOL = 1e-9 on the factor, 1e-8 where a solve or a cancellation
(k** - |L^-1 k*|^2) amplifies the round-off.
"""
import json
import os

import numpy as np
import pytest

from conftest import ROOT, rel_err
from oracle import oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-9


def _problem(ctx, n, d, cfg, m, kernel_name):
    from friedrich_amd import synth

    X, y, Xq = synth.make_problem(n, d, cfg=cfg, m=m)
    ls = ctx.mean_pairwise_distance(X)
    hp = synth.default_hyperparameters(X, y, ls)
    return X, y, Xq, hp, (kernel_name, hp["ls"], hp["ampl"])


def _leading_columns(chol, n, ncols):
    """first ncols columns of the device factor without an n x n host copy: download into HBM, slice there"""
    import torch

    dev = torch.device("cuda", 0)
    buf = torch.empty((n, n), dtype=torch.float64, device=dev).t()  # column-major n x n
    import ctypes
    chol.ctx.check(chol.lib.fr_chol_download_l(chol.h, ctypes.c_void_p(buf.data_ptr()), n, 0))
    chol.ctx.synchronize()
    out = buf[:, :ncols].cpu().numpy()
    del buf
    torch.cuda.empty_cache()
    return out


def test_config0_default_path_n512_matches_golden(ctx):
    """configs[0]: N = 512, d = 1, GaussianProcess::default (builder heuristics, constant-prior fit, scaled ADAM loop with
    one fr_grad_terms + one fr_chol_refactor per iteration) through the C ABI vs the oracle's committed golden run
    (mod.rs:96-102, builder.rs:189-214, optimizer.rs:211-283)."""
    from friedrich_amd import synth
    from host_mirror import DeviceGP, get_parameters

    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_v2.json")))["config0_default"]
    c = ref["config"]
    X, y, Xq = synth.make_problem(c["n"], c["d"], cfg=c["cfg"], m=c["m"])
    gp = DeviceGP.default(ctx, X, y, c["max_iter"], c["convergence_fraction"])
    assert gp.iterations == ref["iterations"]
    assert rel_err(get_parameters(gp.kernel), ref["params"]) < 1e-8
    assert abs(gp.noise / ref["noise"] - 1.0) < 1e-8
    assert abs(gp.prior_c - ref["prior"]) < 1e-13
    assert rel_err(gp.predict(Xq), ref["mean"]) < 1e-8
    var = gp.predict_variance(Xq)
    assert np.max(np.abs(var - np.array(ref["var"]))) < 1e-8 * np.max(np.abs(ref["var"]))
    assert abs(gp.likelihood() / ref["likelihood"] - 1.0) < 1e-8
    gp.close()


def test_config1_n4096_factor_and_predict_variance_vs_oracle(ctx):
    """configs[1]: N = 4096, d = 8, RBF: the whole factor and predict_variance (m = 1024) against the oracle."""
    n, d, m = 4096, 8, 1024
    X, y, Xq, hp, k = _problem(ctx, n, d, 1, m, "squared_exp")
    with O.threads():
        st, L_o, idx = O.make_cholesky_cov_matrix_cols(k, X, hp["noise"])
        assert st == 0 and len(idx) == 0
        L_o = np.asfortranarray(np.tril(L_o))
        kl = O.make_covariance_matrix(k, X, Xq)
        st, kl = O.solve_lower(L_o, kl)  # mod.rs:260-263
        assert st == 0
    var_o = np.array([O.kernel(k, Xq[i], Xq[i]) for i in range(m)]) - np.sum(kl * kl, axis=0)  # :266-270
    chol = ctx.cholesky_from_inputs(k, X, hp["noise"])
    assert chol.info()["n_subst"] == 0
    assert rel_err(chol.l(), L_o) < TOL
    var = chol.predict_variance(k, Xq)
    assert np.max(np.abs(var - var_o)) < 1e-8 * np.max(np.abs(var_o))
    # the m = 1 latency path (narrow solve kernels) gives the same numbers
    var1 = np.concatenate([chol.predict_variance(k, Xq[i:i + 1]) for i in range(4)])
    assert np.max(np.abs(var1 - var_o[:4])) < 1e-8 * np.max(np.abs(var_o))
    chol.free()


def test_config4_n8192_grown_factor_and_sample_at_vs_oracle(ctx):
    """configs[4]: N = 8192, d = 8 grown from 4096 by 8 add_samples of 512 rows, then sample_at (m = 256): the final
    factor against the oracle's factor of the 8192 rows, and mean / covariance / cholesky(cov) of the posterior."""
    n, d, chunk, m = 8192, 8, 512, 256
    X, y, Xq, hp, k = _problem(ctx, n, d, 4, m, "squared_exp")
    noise = hp["noise"]
    with O.threads():
        st, L_o, _ = O.make_cholesky_cov_matrix_cols(k, X, noise)
        assert st == 0
        L_o = np.asfortranarray(np.tril(L_o))
    grown = ctx.cholesky_from_inputs(k, X[:n // 2], noise, capacity_hint=n)
    for hi in range(n // 2 + chunk, n + 1, chunk):
        grown.add_rows(k, X[:hi], chunk, noise)
    assert grown.n == n
    assert rel_err(grown.l(), L_o) < TOL
    yres = y - hp["prior"]
    prior_q = np.full(m, hp["prior"])
    mean, cov, cov_l = grown.posterior(k, yres, Xq, prior_q)
    with O.threads():
        gp = O.OracleGP.__new__(O.OracleGP)  # the oracle model around the factor computed above (no second O(n^3))
        gp.prior, gp.prog, gp.noise, gp.cholesky_epsilon = O.ConstantPrior(hp["prior"]), O.kprog(k), noise, None
        gp.X, gp.y, gp.L, gp.subst = np.asfortranarray(X), yres, L_o, np.zeros(0, dtype=np.int64)
        mo, co, lo = gp.sample_at(Xq)
    assert rel_err(mean, mo) < 1e-8
    assert rel_err(np.tril(cov), np.tril(co)) < 1e-8
    cond = np.linalg.cond(co)
    assert rel_err(cov_l, lo) < max(1e-8, 1e-13 * cond)
    grown.free()


@pytest.mark.parametrize("name,n,d,kernel_name,with_eps,ncols", [
    ("config2_matern52_eps", 16384, 16, "matern2", True, 16384),
    ("config3_rbf", 32768, 16, "squared_exp", False, 8192),
])
def test_full_size_factor_vs_oracle(ctx, name, n, d, kernel_name, with_eps, ncols):
    """configs[2]: the WHOLE N = 16384 factor against the oracle (32 panels of 512 columns, both XCD-reservation tiers);
    configs[3]: the leading 8192 columns of the N = 32768 factor (eight look-ahead panels of 1024 columns, i.e. every column
    compared has seen up to seven trailing updates; columns of a left-looking factorisation depend only on the columns in
    front of them, so the oracle stops there; all 32768 columns: the opt-in test below).  The automatic
    1024 -> 512 panel switch of the last 16384 rows is covered against the oracle by the configs[2]-sized tail test below."""
    X, y, _, hp, k = _problem(ctx, n, d, 3, 0, kernel_name)
    noise = hp["noise"]
    eps = 1e-2 * noise * noise if with_eps else None  # builder.rs:151-style epsilon (SURVEY section 8d cfg 3)
    with O.threads():
        st, L_o, idx = O.make_cholesky_cov_matrix_cols(k, X, noise, eps, ncols)
    assert st == 0 and len(idx) == 0
    chol = ctx.cholesky_from_inputs(k, X, noise, eps=eps, capacity_hint=n)
    info = chol.info()
    assert info["fail_col"] == -1 and info["n_subst"] == 0
    L = _leading_columns(chol, n, ncols)
    assert rel_err(L, np.tril(L_o)) < TOL
    chol.free()


def test_config3_whole_n32768_factor_and_m4096_predict_vs_oracle(ctx):
    """configs[3], one GPU: ALL 32768 columns of the factor against the oracle (every panel of 1024, the switch to 512-column
    panels for the last 16384 rows, every XCD-reservation tier, the super-tiled trailing updates), then the predict family on
    that factor in the shapes bench.py times: **fr_predict_mean at N = 32768 x m = 4096** (left-looking 2048-row blocks, the
    mirrored-pair leaves) and **fr_predict_variance at m = 1024** (leaves cut along K), 16 sampled query columns of each
    compared with the oracle's solve / solve_lower on those columns (mod.rs:234-241, 260-270), plus 1 / 16 / 64 points (K8, K9,
    column groups).  In the default suite since round 5 (the oracle's part takes ~105 s on the GPU box's host threads); the
    result is written to gpurun_out/parity_full_32768.json and kept under profiles/."""
    import time

    n, d, m = 32768, 16, 4096
    X, y, Xq, hp, k = _problem(ctx, n, d, 3, m, "squared_exp")
    noise = hp["noise"]
    t0 = time.time()
    with O.threads():
        st, L_o, idx = O.make_cholesky_cov_matrix_cols(k, X, noise, None, n)
    t_oracle = time.time() - t0
    assert st == 0 and len(idx) == 0
    chol = ctx.cholesky_from_inputs(k, X, noise, capacity_hint=n)
    info = chol.info()
    assert info["fail_col"] == -1 and info["n_subst"] == 0
    worst = 0.0
    per_block = []
    for c0 in range(0, n, 4096):  # (column blocks: no second n x n host copy)
        Lc = _leading_columns(chol, n, c0 + 4096)[:, c0:]
        ref = L_o[:, c0:c0 + 4096]
        ref = np.where(np.arange(n)[:, None] >= (c0 + np.arange(4096))[None, :], ref, 0.0)
        e = rel_err(Lc, ref)
        per_block.append(float(e))
        worst = max(worst, e)
    yres = y - hp["prior"]
    # 64 leading points (the small-m solve paths) + 16 columns sampled over the m = 4096 / m = 1024 query sets
    rng = np.random.RandomState(5)
    sample_mean = np.sort(rng.choice(m, 16, replace=False))
    sample_var = np.sort(rng.choice(1024, 16, replace=False))
    rows = np.unique(np.concatenate([np.arange(64), sample_mean, sample_var]))
    with O.threads():
        gp = O.OracleGP.__new__(O.OracleGP)  # the oracle model around the factor computed above (no second O(n^3))
        gp.prior, gp.prog, gp.noise, gp.cholesky_epsilon = O.ConstantPrior(hp["prior"]), O.kprog(k), noise, None
        gp.X, gp.y, gp.L, gp.subst = np.asfortranarray(X), yres, np.asfortranarray(L_o), np.zeros(0, dtype=np.int64)
        mo_rows, vo_rows = gp.predict(Xq[rows]), gp.predict_variance(Xq[rows])
    mo = np.full(m, np.nan)
    vo = np.full(m, np.nan)
    mo[rows], vo[rows] = mo_rows, vo_rows
    pred = {}
    for mm in (1, 16, 64):
        mean = chol.predict_mean(k, yres, Xq[:mm], np.full(mm, hp["prior"]))
        var = chol.predict_variance(k, Xq[:mm])
        pred[str(mm)] = {"mean": float(rel_err(mean, mo[:mm])), "variance": float(rel_err(var, vo[:mm]))}
    # the bench's own predict shape: every one of the 4096 columns is solved, 16 of them are checked against the oracle
    mean = chol.predict_mean(k, yres, Xq, np.full(m, hp["prior"]))
    assert np.all(np.isfinite(mean))
    var = chol.predict_variance(k, Xq[:1024])
    assert np.all(np.isfinite(var))
    pred["4096_mean_16_sampled_columns"] = {"mean": float(rel_err(mean[sample_mean], mo[sample_mean]))}
    pred["1024_variance_16_sampled_columns"] = {"variance": float(rel_err(var[sample_var], vo[sample_var]))}
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_full_32768.json"), "w") as f:
        json.dump({"config": "configs[3] one GPU: N=32768 d=16 RBF, friedrich default hyper-parameters, whole factor vs the threaded oracle; predict_mean m=4096 and predict_variance m=1024 (bench shapes) on 16 sampled columns",
                   "relative_error_per_4096_column_block": per_block, "worst": float(worst), "tolerance": TOL,
                   "predict_relative_error_by_number_of_points": pred, "predict_tolerance": 1e-8,
                   "sampled_columns_mean": sample_mean.tolist(), "sampled_columns_variance": sample_var.tolist(),
                   "oracle_seconds": round(t_oracle, 1)}, f, indent=1)
    assert worst < TOL
    for mm, e in pred.items():
        for what, v in e.items():
            assert v < 1e-8, (mm, what, v)
    chol.free()


def test_large_n_schedule_on_n8192_vs_oracle(ctx):
    """The schedule of a fit at N >= 18432 -- 1024-column panels while the trailing update dominates, 512-column panels with
    XCDs set aside for the panel chain over the last rows -- forced onto a matrix the oracle factors in full (N = 8192:
    nb = 1024, switch to 512 columns below 4096 remaining rows): the same factor as the default schedule's, to TOL of the
    oracle's."""
    n, d = 8192, 8
    X, y, _, hp, k = _problem(ctx, n, d, 4, 0, "squared_exp")
    with O.threads():
        st, L_o, _ = O.make_cholesky_cov_matrix_cols(k, X, hp["noise"])
    assert st == 0
    L_o = np.tril(L_o)
    ctx.set_option("nb", 1024)
    ctx.set_option("nb_switch_rows", 4096)
    try:
        chol = ctx.cholesky_from_inputs(k, X, hp["noise"])
        assert rel_err(chol.l(), L_o) < TOL
        chol.free()
    finally:
        ctx.set_option("nb", 0)
        ctx.set_option("nb_switch_rows", 16384)


def _duplicated_rows_problem(n, d, ndup, cfg):
    from friedrich_amd import synth

    X, y, _ = synth.make_problem(n, d, cfg=cfg)
    X = np.array(X, order="F")
    # rows n - ndup .. n - 1 are exact copies of rows 0 .. ndup - 1: with noise = 0 the Gram matrix is singular there
    X[n - ndup:] = X[:ndup]
    return X, y


def test_config2_substitutions_fire_noise0_duplicated_rows(ctx):
    """configs[2] sub-case (ii): noise = 0 + 256 duplicated rows, cholesky_epsilon = 1e-2 * noise0^2, so
    the substitute fires.  On a size the oracle can do (N = 4096) the substituted column sets are compared: a duplicated
    row's pivot is an exact-arithmetic zero, so its computed sign is round-off and the left-looking (reference) and the
    blocked right-looking (HIP) orders may disagree inside that ambiguous band -- the symmetric difference is reported and
    bounded, every substituted column must lie in the duplicated block, and away from it the factors agree.  At N = 16384
    the count is reported with the same band."""
    from friedrich_amd import synth

    d, ndup = 16, 256
    report = {}
    for n, use_oracle in ((4096, True), (16384, False)):
        X, y = _duplicated_rows_problem(n, d, ndup, 3)
        ls = ctx.mean_pairwise_distance(X)
        hp = synth.default_hyperparameters(X, y, ls)
        k = ("matern2", hp["ls"], hp["ampl"])
        eps = 1e-2 * hp["noise"] ** 2
        chol = ctx.cholesky_from_inputs(k, X, 0.0, eps=eps, capacity_hint=n)
        idx = chol.substitutions()
        info = chol.info()
        assert info["fail_col"] == -1
        assert np.all(idx >= n - ndup), "a pivot outside the duplicated block was substituted"
        assert np.all(np.diff(idx) > 0)
        report[n] = {"hip": len(idx)}
        assert 0.85 * ndup <= len(idx) <= ndup  # measured: see profiles/r04/parity_band.json (the file this test writes)
        if use_oracle:
            with O.threads():
                st, L_o, idx_o = O.make_cholesky_cov_matrix_cols(k, X, 0.0, eps)
            assert st == 0 and np.all(idx_o >= n - ndup)
            sym = sorted(set(idx.tolist()) ^ set(idx_o.tolist()))
            report[n].update({"oracle": len(idx_o), "symmetric_difference": len(sym)})
            # The band, measured (profiles/r04/parity_band.json -- written by THIS test, so the numbers quoted in DESIGN.md
            # section 2 are the ones of the build that ran; the list itself is a function of the input, see
            # test_config2_substitution_list_is_deterministic) and held with a margin: both orders substitute at least 70 % of the duplicated block's
            # pivots, their counts differ by at most a quarter of the block, and at most 35 % of the block is decided
            # differently.  (The pivots in question are exact-arithmetic zeros: what each order computes there is pure
            # round-off of ~4000 accumulated products, so agreement beyond the band would be coincidence.)
            assert 0.70 * ndup <= len(idx_o) <= ndup and 0.70 * ndup <= len(idx) <= ndup
            assert abs(len(idx) - len(idx_o)) <= 0.25 * ndup
            assert len(sym) <= 0.35 * ndup
            # columns in front of the duplicated block are untouched by any substitution: plain parity there
            L = chol.l()
            assert rel_err(L[:, :n - ndup], np.tril(L_o)[:, :n - ndup]) < TOL
        chol.free()
    print("configs[2](ii) substitution counts:", report)
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "parity_band.json"), "w") as f:
        json.dump({"what": "configs[2](ii): noise = 0, 256 duplicated rows, cholesky_epsilon = 1e-2 noise0^2, Matern-5/2, d = 16: substituted "
                           "pivots of the HIP path (blocked right-looking) and of the oracle (left-looking, the reference's order)",
                   "counts": {str(k_): v for k_, v in report.items()}}, f, indent=1)


_SUBST_SCRIPT = """
import json, sys
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
from friedrich_amd import synth
from friedrich_amd.device import Context
from test_gpu_fullsize_oracle import _duplicated_rows_problem
ctx = Context()
n, d, ndup = {n}, 16, 256
X, y = _duplicated_rows_problem(n, d, ndup, 3)
ls = ctx.mean_pairwise_distance(X)
hp = synth.default_hyperparameters(X, y, ls)
k = ("matern2", hp["ls"], hp["ampl"])
chol = ctx.cholesky_from_inputs(k, X, 0.0, eps=1e-2 * hp["noise"] ** 2, capacity_hint=n)
L = chol.l()
print("RESULT " + json.dumps({{"idx": chol.substitutions().tolist(), "checksum": float(np.abs(L).sum()), "last": float(L[-1, -1])}}))
"""


def test_config2_substitution_list_is_deterministic(ctx):
    """north_star: "rank/pivot indices bit-exact".  Inside the ambiguous band of configs[2](ii) the list cannot be compared
    with the oracle's entry for entry (exact-zero pivots: the sign is the round-off of the summation order), but it has to be
    a FUNCTION OF THE INPUT: two fits in one process, a fit in a fresh process, and a fit in a process whose kernels are
    serialised (AMD_SERIALIZE_KERNEL=3: no two kernels overlap, tiles are claimed in another order, XCDs are reserved for
    nobody) return the identical list and the bit-identical factor -- no atomics in floating point, fixed summation order in
    the split-K reductions, every result tile computed by one workgroup whatever the placement."""
    import subprocess
    import sys

    from friedrich_amd import synth

    n, d, ndup = 4096, 16, 256
    X, y = _duplicated_rows_problem(n, d, ndup, 3)
    ls = ctx.mean_pairwise_distance(X)
    hp = synth.default_hyperparameters(X, y, ls)
    k = ("matern2", hp["ls"], hp["ampl"])
    eps = 1e-2 * hp["noise"] ** 2
    runs = []
    for _ in range(2):
        chol = ctx.cholesky_from_inputs(k, X, 0.0, eps=eps, capacity_hint=n)
        runs.append((chol.substitutions().tolist(), chol.l()))
        chol.refactor(k, 0.0, eps=eps)
        runs.append((chol.substitutions().tolist(), chol.l()))
        chol.free()
    for idx, L in runs[1:]:
        assert idx == runs[0][0]
        assert np.array_equal(L, runs[0][1])
    assert len(runs[0][0]) > 0.7 * ndup
    script = _SUBST_SCRIPT.format(root=ROOT, tests=os.path.join(ROOT, "tests"), n=n)
    for extra in ({}, {"AMD_SERIALIZE_KERNEL": "3", "AMD_SERIALIZE_COPY": "3"}):
        out = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=600, env=dict(os.environ, **extra))
        assert out.returncode == 0, out.stderr[-2000:]
        res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("RESULT ")][0][7:])
        assert res["idx"] == runs[0][0], (extra, len(res["idx"]), len(runs[0][0]))
        assert res["checksum"] == float(np.abs(runs[0][1]).sum()) and res["last"] == float(runs[0][1][-1, -1])


@pytest.mark.parametrize("noise", [1e-2, 1e-3, 1e-4, 1e-5, 1e-6])
@pytest.mark.parametrize("with_eps", [False, True])
def test_conditioning_sweep_rbf_d1(ctx, noise, with_eps):
    """Ill-conditioned regime (RBF, d = 1, short length scale, small noise: what cholesky_epsilon exists for;
    cond(K) = 1e6 .. 1e14, the 128 x 128 diagonal blocks nearly as bad).  The HIP path multiplies by explicit inverses of
    the diagonal blocks where the reference substitutes; alone that costs a factor cond(L_bb) of backward accuracy (measured
    on this sweep in round 2: L L^T - K grew to 4e-13 / 3e-12 at noise 1e-4 / 1e-5, and at 1e-6 a Schur complement went
    negative -- 896 substituted pivots where the oracle has none).  The library therefore estimates the conditioning of every
    diagonal block while factoring and, when one is ill-conditioned, factors again with one step of iterative refinement
    against the triangular block behind every such product (fr_chol_conditioning).  What is asserted:
      * the pivot decisions agree with the oracle at EVERY noise level (no substitution, no failure);
      * backward: L L^T = K and K x = b to round-off;
      * forward, against the oracle: two independent f64 factorisations of K (oracle vs LAPACK dpotrf, measured on this very
        sweep) differ by 0.01 cond u on L and 0.2 cond u on a solve; the HIP path is held to 0.1 cond u / 2 cond u, floored at
        1e-9 -- north_star's 1e-8 wherever cond(K) <= 4.5e7, the conditioning limit of the problem itself beyond."""
    n, d, m = 1024, 1, 64
    rng = np.random.default_rng(7)
    X = np.asfortranarray(np.sort(rng.random((n, d)), axis=0))
    Xq = np.asfortranarray(rng.random((m, d)))
    y = np.sin(6.0 * X[:, 0])
    k = ("squared_exp", 0.05, 1.0)
    eps = 1e-2 * noise * noise if with_eps else None
    st, L_o, idx_o = O.make_cholesky_cov_matrix(k, X, noise, eps)
    assert st == 0 and len(idx_o) == 0  # the oracle neither fails nor substitutes anywhere in this sweep
    K = O.make_covariance_matrix(k, X, X) + noise * noise * np.eye(n)
    chol = ctx.cholesky_from_inputs(k, X, noise, eps=eps, allow_failure=True)
    info = chol.info()
    est, refined = chol.conditioning()
    assert info["fail_col"] == -1 and info["n_subst"] == 0, (info, est, refined)
    assert refined  # every diagonal block of this fixture is ill-conditioned
    L = chol.l()
    L_o = np.tril(L_o)
    ku = np.linalg.cond(K) * 2.2e-16
    # backward: the factor reproduces K, the solve inverts it, to round-off
    assert rel_err(L @ L.T, K) < 1e-13
    B = np.asfortranarray(rng.standard_normal((n, 3)))
    Z = chol.solve(B)
    resid = np.max(np.abs(K @ Z - B)) / (np.max(np.abs(K)) * np.max(np.abs(Z)) * n)
    assert resid < 1e-13, resid
    # forward, against the oracle
    e_l = rel_err(L, L_o)
    e_s = rel_err(Z, O.chol_solve(L_o, B))
    gp = O.OracleGP(O.ZeroPrior(), k, noise, eps, X, y)
    e_p = rel_err(chol.predict_mean(k, y, Xq, None), gp.predict(Xq))
    var_o = gp.predict_variance(Xq)
    e_v = float(np.max(np.abs(chol.predict_variance(k, Xq) - var_o)) / np.max(np.abs(var_o)))
    print(f"conditioning sweep noise={noise:g} eps={eps}: block estimate {est:.1e} refined {refined}  cond*u={ku:.1e}  L {e_l:.1e}  solve {e_s:.1e}  "
          f"predict {e_p:.1e}  variance {e_v:.1e}  residual {resid:.1e}")
    assert e_l < max(1e-9, 0.1 * ku)
    assert e_s < max(1e-9, 2.0 * ku)
    assert e_p < max(1e-9, 2.0 * ku)
    assert e_v < max(1e-9, 2.0 * ku)
    chol.free()
