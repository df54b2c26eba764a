"""Forward progress of the persistent solves (trsv.hip, trsm_narrow.hip) and the fall-back behind them.

Round 2 dealt block r of a solve to workgroup r % G and relied on all G workgroups being resident; with fewer of them on
the chip (another context's kernels holding CUs, more blocks than CUs) the resident ones waited for blocks nobody had
started.  Blocks are now CLAIMED in order of arrival, so a block only waits for workgroups that already run; and an entry
point whose persistent solve still reports a timed-out hand-off repeats its work on the recursive GEMM path.  The
reference operations behind these solves: src/gaussian_process/mod.rs:235, 260-263 (predict / predict_variance of a few
points -- the Bayesian-optimisation loop of readme.md:7)."""
import os
import threading

import numpy as np
import pytest

from conftest import PD_KERNELS, rand_inputs, rel_err
from oracle import oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-9


def _ctx_with_env(**env):
    from friedrich_amd.device import Context

    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return Context()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("wgs", [1, 3])
@pytest.mark.parametrize("n,m", [(2049, 1), (2049, 16), (1300, 100), (3000, 7)])
def test_claimed_blocks_with_fewer_workgroups_than_blocks(ctx, wgs, n, m):
    """A grid of 1 or 3 workgroups walks through 11 ... 24 blocks per sweep (the round-2 dealing deadlocks here by
    construction: workgroup 0's second block waits for first blocks of workgroups that do not exist): same numbers as the
    full grid, bit for bit -- the arithmetic of a block does not depend on who computes it."""
    k = PD_KERNELS[1]
    X = rand_inputs(n, 3, 900 + n)
    B = np.asfortranarray(np.random.default_rng(n + m).standard_normal((n, m)))
    st, L_o, _ = O.make_cholesky_cov_matrix(k, X, 0.1)
    chol = ctx.cholesky_from_inputs(k, X, 0.1)
    z, w = chol.solve(B), chol.solve_lower(B)
    assert rel_err(z, O.chol_solve(L_o, B)) < TOL
    chol.free()
    c2 = _ctx_with_env(FRIEDRICH_AMD_TEST_MAX_WORKGROUPS=wgs)
    try:
        chol2 = c2.cholesky_from_inputs(k, X, 0.1)
        assert np.array_equal(chol2.solve(B), z) and np.array_equal(chol2.solve_lower(B), w)
        assert c2.counter("solve_retries") == 0
        chol2.free()
    finally:
        c2.close()


def test_persistent_solves_next_to_a_second_contexts_gemm(ctx):
    """N = 36864: 288 blocks for at most 256 workgroups, while a second context keeps the chip full of GEMM workgroups
    (their 2 x 72 KiB of LDS per CU leave no room for a solve workgroup until both have retired, so the solve's grid
    trickles in): one-column and 16-column predicts give the unloaded results, nothing times out, nothing is retried."""
    from friedrich_amd.device import Context

    n, d = 36864, 8
    k = PD_KERNELS[0]
    X = rand_inputs(n, d, 31)
    y = np.sin(X.sum(axis=1))
    Xq = rand_inputs(16, d, 32)
    chol = ctx.cholesky_from_inputs(k, X, 0.3)
    assert chol.info()["fail_col"] == -1
    mean1, var1 = chol.predict_mean(k, y, Xq[:1]), chol.predict_variance(k, Xq[:1])
    mean16, var16 = chol.predict_mean(k, y, Xq), chol.predict_variance(k, Xq)
    # property check of the unloaded results (the oracle cannot factor this size): K (K^-1 k*) = k*
    other = Context()
    stop = threading.Event()
    errs = []

    def load():
        try:
            A = np.asfortranarray(np.random.default_rng(5).standard_normal((8192, 8192)))
            xs = other.inputs_to_device(A, "colmajor")
            try:
                import torch

                C = torch.empty((8192, 8192), dtype=torch.float64, device="cuda").t()
                while not stop.is_set():
                    other.gemm(xs, xs, C)  # 8192^3: ~15 ms of a full chip per call
            finally:
                xs.free()
        except BaseException as e:  # noqa: BLE001
            errs.append(e)

    th = threading.Thread(target=load)
    th.start()
    try:
        for rep in range(6):
            assert np.array_equal(chol.predict_mean(k, y, Xq[:1]), mean1)
            assert np.array_equal(chol.predict_variance(k, Xq[:1]), var1)
            assert np.array_equal(chol.predict_mean(k, y, Xq), mean16)
            assert np.array_equal(chol.predict_variance(k, Xq), var16)
    finally:
        stop.set()
        th.join(120)
        other.close()
    assert not errs, errs
    assert ctx.counter("solve_retries") == 0
    chol.free()


def test_timed_out_solve_is_repeated_on_the_recursive_path():
    """Test hook FRIEDRICH_AMD_TEST_FORCE_SOLVE_TIMEOUT: every persistent launch of the context reports a timed-out
    hand-off.  Every entry point that solves then repeats its work with the persistent kernels off and returns the
    recursive path's numbers (1e-11 from the persistent ones: same blocks, different summation order) instead of
    FR_HIP_ERROR -- in-place device operands included (restored from their backup first)."""
    import torch

    from friedrich_amd.device import Context

    n, d, m = 1500, 4, 16
    k = PD_KERNELS[0]
    X = rand_inputs(n + 130, d, 77)
    y = np.sin(X.sum(axis=1))
    Xq = rand_inputs(m, d, 78)
    B = np.asfortranarray(np.random.default_rng(3).standard_normal((n, 5)))
    good = Context()
    bad = _ctx_with_env(FRIEDRICH_AMD_TEST_FORCE_SOLVE_TIMEOUT=1)
    try:
        cg, cb = good.cholesky_from_inputs(k, X[:n], 0.2, capacity_hint=n + 130), bad.cholesky_from_inputs(k, X[:n], 0.2, capacity_hint=n + 130)

        def both(fn):
            a, b = fn(cg), fn(cb)
            for u, v in zip(a if isinstance(a, tuple) else (a,), b if isinstance(b, tuple) else (b,)):
                assert rel_err(v, u) < 1e-10

        r0 = bad.counter("solve_retries")
        both(lambda c: c.predict_mean(k, y[:n], Xq[:1]))
        both(lambda c: c.predict_variance(k, Xq[:1]))
        both(lambda c: c.predict_mean(k, y[:n], Xq))
        both(lambda c: c.predict_mean_variance(k, y[:n], Xq))
        both(lambda c: c.posterior(k, y[:n], Xq))
        both(lambda c: np.array([c.likelihood(k, y[:n], 0.2)]))
        both(lambda c: c.solve(B))
        both(lambda c: c.solve_lower(B[:, :1]))
        both(lambda c: c.grad_terms(k, y[:n], 0.2, True, 2))
        assert bad.counter("solve_retries") >= r0 + 9 and good.counter("solve_retries") == 0

        def cached(c):
            c.set_targets(y[:n])
            return c.predict_mean(k, None, Xq)

        both(cached)
        # in place on a device operand
        outs = []
        for c in (cg, cb):
            Bd = torch.from_numpy(np.ascontiguousarray(B.T)).cuda().t()  # column-major n x 5 on the device
            c.solve(Bd)
            c.ctx.synchronize()  # device operand: the call does not wait for its stream (friedrich_amd.h, conventions)
            outs.append(Bd.cpu().numpy())
        assert rel_err(outs[1], outs[0]) < 1e-10
        # add_rows: the L21 solve (130 right-hand sides in column groups) is repeated, the append is restartable
        for c in (cg, cb):
            c.add_rows(k, np.asfortranarray(X), 130, 0.2)
        assert rel_err(cb.l(), cg.l()) < 1e-10
        _, L1, _ = O.make_cholesky_cov_matrix(k, X, 0.2)
        assert rel_err(cb.l(), np.tril(L1)) < TOL
        cg.free()
        cb.free()
    finally:
        good.close()
        bad.close()
