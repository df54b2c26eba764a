"""Developer probe: the diagonal-block kernel alone (n <= 128) against numpy, every size, plus the substitution log."""
import sys

import numpy as np

sys.path.insert(0, ".")
from friedrich_amd.device import Context

ctx = Context()
rng = np.random.default_rng(1)
worst = 0.0
for n in list(range(1, 130, 1)) + [160, 200, 256, 300]:
    B = rng.standard_normal((n, n))
    A = B @ B.T / n + np.eye(n) * 0.5
    ch = ctx.cholesky_from_matrix(A)
    L = ch.l()
    Lref = np.linalg.cholesky(A)
    e1 = np.abs(np.tril(L) - Lref).max() / np.abs(Lref).max()
    Ainv = ch.inverse()
    e2 = np.abs(Ainv @ A - np.eye(n)).max()
    worst = max(worst, e1, e2)
    if e1 > 1e-12 or e2 > 1e-10 or not np.isfinite(e1 + e2):
        print(f"n={n}: factor err {e1:.2e} inverse err {e2:.2e}")
    ch.free()
print("worst", worst)
# indefinite matrix with substitution
n = 48
B = rng.standard_normal((n, n))
A = B + B.T
ch = ctx.cholesky_from_matrix(A, eps=1e-6, allow_failure=True)
print("info", ch.info())
