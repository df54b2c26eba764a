"""Developer probe: one BASELINE configuration end to end (for rocprofv3 passes):  config_run.py <1|2|3>
  1: N=4096  d=8  RBF,               fit + predict_variance
  2: N=16384 d=16 Matern-5/2 + cholesky_epsilon, fit + predict + predict_variance
  3: N=32768 d=16 RBF,               fit + predict
  4: N=8192  d=8  RBF,               4096 rows + 8 x add_samples(512) + sample_at (m = 256)"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from friedrich_amd import synth
from friedrich_amd.device import Context

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n, d, kname, eps = {1: (4096, 8, "squared_exp", None), 2: (16384, 16, "matern2", -1.0), 3: (32768, 16, "squared_exp", None),
                     4: (8192, 8, "squared_exp", None)}[cfg]
m = 1024
ctx = Context()
for o in [a[2:] for a in sys.argv[2:] if a.startswith("--")]:  # --name=value context options
    ctx.set_option(o.split("=")[0], int(o.split("=")[1]))
X, y, Xq = synth.make_problem(n, d, cfg=cfg, m=m)
ls = ctx.mean_pairwise_distance(X)
hp = synth.default_hyperparameters(X, y, ls)
k = (kname, hp["ls"], hp["ampl"])
if eps is not None and eps < 0:
    eps = 1e-2 * hp["noise"] ** 2  # cholesky_epsilon of SURVEY.md section 8d, cfg 3
if cfg == 4:
    for rep in range(2):
        g = ctx.cholesky_from_inputs(k, X[:4096], hp["noise"], capacity_hint=n)
        ctx.synchronize()
        t0 = time.perf_counter()
        for hi in range(4096 + 512, n + 1, 512):
            g.add_rows(k, X[:hi], 512, hp["noise"])
        t1 = time.perf_counter()
        mean, cov, cov_l = g.posterior(k, y - hp["prior"], Xq[:256], np.full(256, hp["prior"]))
        t2 = time.perf_counter()
        g.free()
    print(f"config 4: N=4096 -> {n} in 512-row chunks: add_samples {1e3*(t1-t0):.1f} ms, sample_at(m=256) {1e3*(t2-t1):.1f} ms")
    sys.exit(0)
chol = ctx.cholesky_from_inputs(k, X, hp["noise"], eps=eps, capacity_hint=n)
for rep in range(2):
    t0 = time.perf_counter()
    chol.refactor(k, hp["noise"], eps=eps)
    t1 = time.perf_counter()
    mean = chol.predict_mean(k, y - hp["prior"], Xq, np.full(m, hp["prior"]))
    t2 = time.perf_counter()
    var = chol.predict_variance(k, Xq)
    t3 = time.perf_counter()
print(f"config {cfg}: N={n} d={d} {kname} eps={eps}: fit {1e3*(t1-t0):.1f} ms ({n**3/3/(t1-t0)/1e12:.1f} TF/s)  predict(m={m}) {1e3*(t2-t1):.1f} ms  "
      f"variance {1e3*(t3-t2):.1f} ms  substitutions {chol.info()['n_subst']}")
