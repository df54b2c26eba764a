"""The resident panel chain (potf2.hip: panel_chain_kernel, option panel_chain) against the oracle and against the chain of launches it
replaces: ONE launch factors the kb x kb diagonal block of a panel (kb = 256 / 384 / 512) -- the flat diagonal-block body on workgroup 0,
LDS-resident 16-row slabs that hand over through memory -- and solves the rows below it.  Reference work replaced:
nalgebra's column Cholesky behind make_cholesky_cov_matrix (src/algebra/mod.rs:81-91) and Cholesky::insert_column (:124)."""
import numpy as np
import pytest

from conftest import PD_KERNELS, rand_inputs, rel_err
from oracle import oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-9


def _factor(ctx, mode, k, X, noise, eps=None, nb=0, allow_failure=False):
    ctx.set_option("panel_chain", mode)
    ctx.set_option("nb", nb)
    c0 = ctx.counter("panel_chain_launches")
    chol = ctx.cholesky_from_inputs(k, X, noise, eps=eps, allow_failure=allow_failure)
    out = (chol.l(), chol.info(), chol.substitutions() if eps is not None else None, ctx.counter("panel_chain_launches") - c0)
    chol.free()
    return out


@pytest.mark.parametrize("n", [256, 300, 384, 512, 513, 640, 1000, 1024, 1700, 2100, 4096, 5000])
def test_resident_chain_matches_oracle_and_launch_chain(ctx, n):
    """every shape class: diagonal block only (n = 256 .. 512), resident 16-row slabs below it (<= 512 rows below), 32- / 64-row
    workgroups (more), ragged last slabs (n not a multiple of 16), the look-ahead pipeline with the panel stream's XCDs / CUs set aside"""
    k = PD_KERNELS[1]
    X = rand_inputs(n, 5, n)
    st, L_o, _ = O.make_cholesky_cov_matrix(k, X, 0.1)
    assert st == 0
    try:
        L0, info0, _, used0 = _factor(ctx, 0, k, X, 0.1)
        L2, info2, _, used2 = _factor(ctx, 2, k, X, 0.1)
        # (a panel is taken when it is 256 / 384 / 512 columns wide: n = 300 is one panel of 300 columns and keeps the launch chain)
        assert used0 == 0 and (used2 >= 1) == (n >= 512 or n % 128 == 0), (used0, used2)
        assert rel_err(L2, np.tril(L_o)) < TOL and rel_err(L0, np.tril(L_o)) < TOL
        assert rel_err(L2, L0) < 1e-12  # (the same products in a different order of summation)
        assert info2 == info0 and info2["fail_col"] == -1 and info2["n_subst"] == 0
        assert ctx.counter("panel_chain_fallbacks") == 0
    finally:
        ctx.set_option("panel_chain", 2)
        ctx.set_option("nb", 0)


def test_resident_chain_pivot_rule_and_log_order(ctx):
    """cholesky_epsilon inside the resident launch: the substituted columns are logged in column order by workgroup 0, block after
    block -- the exact list of the oracle on the margin fixture (HyperTan Gram, indefinite) -- and a failure reports the first column"""
    n = 600
    X = rand_inputs(n, 2, 5) * 3.0
    k = ("hyper_tan", 1.0, 0.0)
    st, L_o, idx_o = O.make_cholesky_cov_matrix(k, X, 0.0, 1e-6)
    assert st == 0 and len(idx_o) > 100
    try:
        for nb in (256, 512):
            L2, info2, idx2, used = _factor(ctx, 2, k, X, 0.0, eps=1e-6, nb=nb)
            assert used >= 1
            assert idx2.tolist() == idx_o.tolist()  # (the factor itself overflows on this indefinite fixture: the list is what is pinned)
        st_f, _, _ = O.make_cholesky_cov_matrix(k, X, 0.0)
        assert st_f > 0
        _, info_f, _, used = _factor(ctx, 2, k, X, 0.0, allow_failure=True)
        assert used >= 1 and info_f["fail_col"] == st_f - 1
    finally:
        ctx.set_option("panel_chain", 2)
        ctx.set_option("nb", 0)


def test_resident_chain_is_deterministic_and_placement_independent(ctx):
    """two fits in one process and a fit with the XCD reservation off (the workgroups land elsewhere) give the bit-identical factor:
    every element is computed by one workgroup in a fixed order, the hand-offs carry no arithmetic"""
    n = 3072  # (whole 512-column panels: every diagonal block goes through the resident launch in both placements)
    k = PD_KERNELS[0]
    X = rand_inputs(n, 4, 11)
    try:
        La, *_ = _factor(ctx, 2, k, X, 0.1)
        Lb, *_ = _factor(ctx, 2, k, X, 0.1)
        ctx.set_option("xcd_reserve", 0)
        Lc, *_ = _factor(ctx, 2, k, X, 0.1)
        assert np.array_equal(La, Lb)
        assert np.array_equal(La, Lc)
    finally:
        ctx.set_option("xcd_reserve", -1)
        ctx.set_option("panel_chain", 2)


def test_timed_out_chain_falls_back_to_the_launch_chain():
    """FRIEDRICH_AMD_TEST_FORCE_SOLVE_TIMEOUT: every resident launch reports a timed-out hand-off; the factorisation is repeated on
    the chain of launches (counter panel_chain_fallbacks) and returns the right factor instead of FR_HIP_ERROR"""
    from test_gpu_solve_progress import _ctx_with_env

    n = 1200
    k = PD_KERNELS[1]
    X = rand_inputs(n, 5, 3)
    st, L_o, _ = O.make_cholesky_cov_matrix(k, X, 0.1)
    bad = _ctx_with_env(FRIEDRICH_AMD_TEST_FORCE_SOLVE_TIMEOUT=1)
    try:
        chol = bad.cholesky_from_inputs(k, X, 0.1)
        assert rel_err(chol.l(), np.tril(L_o)) < TOL
        assert bad.counter("panel_chain_fallbacks") >= 1
        f0 = bad.counter("panel_chain_fallbacks")
        chol.refactor(k, 0.1)
        assert rel_err(chol.l(), np.tril(L_o)) < TOL and bad.counter("panel_chain_fallbacks") > f0
        chol.free()
        A = np.asfortranarray(np.tril(L_o) @ np.tril(L_o).T)
        c2 = bad.cholesky_from_matrix(A)
        assert rel_err(c2.l(), np.tril(L_o)) < 1e-8
        c2.free()
    finally:
        bad.close()
