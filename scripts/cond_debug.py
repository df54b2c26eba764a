"""Developer probe: where does the factorisation of the ill-conditioned RBF d = 1 fixture leave the oracle?"""
import sys
import numpy as np
sys.path.insert(0, ".")
from friedrich_amd.device import Context
from oracle import oracle as O

ctx = Context()
rng = np.random.default_rng(7)
Xall = np.asfortranarray(np.sort(rng.random((1024, 1)), axis=0))
k = ("squared_exp", 0.05, 1.0)
for noise in (1e-2, 1e-4):
    for n in (32, 64, 128, 129, 160, 256, 512, 1024):
        X = np.asfortranarray(Xall[:n])
        st, L_o, _ = O.make_cholesky_cov_matrix(k, X, noise)
        chol = ctx.cholesky_from_inputs(k, X, noise, allow_failure=True)
        info = chol.info()
        L = chol.l()
        Lo = np.tril(L_o)
        bad = ~np.isfinite(L)
        err = np.abs(L - Lo) / np.max(np.abs(Lo))
        err[bad] = 1.0
        colerr = err.max(axis=0)
        first = int(np.argmax(colerr > 1e-6)) if np.any(colerr > 1e-6) else -1
        K = ctx.gram(k, X, X)
        Ko = O.make_covariance_matrix(k, X, X)
        print(f"noise {noise:g} n {n}: oracle st {st}, hip fail_col {info['fail_col']} nonfinite {int(bad.sum())} max err {colerr.max():.2e} "
              f"first bad col {first}  gram err {np.max(np.abs(K-Ko)):.1e} min L diag hip {np.nanmin(np.diag(L)):.3e} oracle {np.min(np.diag(Lo)):.3e}")
        if first >= 0 and n <= 256:
            rows = np.argsort(-err[:, first])[:4]
            print("    col", first, "worst rows", rows, "hip", L[rows, first], "oracle", Lo[rows, first])
        chol.free()
