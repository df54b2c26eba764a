"""Developer probe: BASELINE configs[4] -- N = 8192, d = 8 grown by add_samples in 512-row chunks, then sample_at."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from friedrich_amd import synth
from friedrich_amd.device import Context

ctx = Context()
n, d, chunk, m = 8192, 8, 512, 256
X, y, Xq = synth.make_problem(n, d, cfg=4, m=m)
ls = ctx.mean_pairwise_distance(X)
hp = synth.default_hyperparameters(X, y, ls)
k = ("squared_exp", hp["ls"], hp["ampl"])
for rep in range(3):
    t0 = time.perf_counter()
    chol = ctx.cholesky_from_inputs(k, X[:chunk], hp["noise"], capacity_hint=n)
    t1 = time.perf_counter()
    per = []
    for hi in range(2 * chunk, n + 1, chunk):
        ta = time.perf_counter()
        chol.add_rows(k, X[:hi], chunk, hp["noise"])
        ctx.synchronize()
        per.append(time.perf_counter() - ta)
    t2 = time.perf_counter()
    mean, cov, cov_l = chol.posterior(k, y - hp["prior"], Xq, np.full(m, hp["prior"]))
    t3 = time.perf_counter()
    print(f"rep {rep}: first fit {1e3*(t1-t0):.1f} ms; 15 add_rows {1e3*(t2-t1):.1f} ms (first {1e3*per[0]:.2f}, last {1e3*per[-1]:.2f} ms); sample_at(m={m}) {1e3*(t3-t2):.1f} ms")
    chol.free()
t0 = time.perf_counter()
chol = ctx.cholesky_from_inputs(k, X, hp["noise"], capacity_hint=n)
print(f"from-scratch fit of all 8192 rows: {1e3*(time.perf_counter()-t0):.1f} ms")
chol.free()
# class profile of the last chunk (7680 -> 8192)
chol = ctx.cholesky_from_inputs(k, X[:n - chunk], hp["noise"], capacity_hint=n)
ctx.profile_reset()
ctx.profile_enable(True)
t0 = time.perf_counter()
chol.add_rows(k, X, chunk, hp["noise"])
t1 = time.perf_counter()
print(f"last chunk alone (profiled): {1e3*(t1-t0):.2f} ms")
for name, p in ctx.profile().items():
    if p["launches"]:
        print(f"    {name:11s} {p['ms']:8.3f} ms  {p['launches']:5d} launches")
ctx.profile_enable(False)
