"""Developer probe: predict / predict_variance with few query rows: memory-bound narrow kernels vs the GEMM path."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from friedrich_amd import synth
from friedrich_amd.device import Context

ctx = Context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
X, y, _ = synth.make_problem(n, 16, cfg=4, m=8)
ls = ctx.mean_pairwise_distance(X)
hp = synth.default_hyperparameters(X, y, ls)
k = ("squared_exp", hp["ls"], hp["ampl"])
chol = ctx.cholesky_from_inputs(k, X, hp["noise"], capacity_hint=n)
for m in (2, 8, 16, 32, 64, 128, 256):
    _, _, Xq = synth.make_problem(n, 16, cfg=4, m=m)
    for sk in (0, 1):
        ctx.set_option("narrow_max", 4096 if sk else 1)
        tp, tv = [], []
        for rep in range(3):
            t0 = time.perf_counter()
            mean = chol.predict_mean(k, y - hp["prior"], Xq, np.full(m, hp["prior"]))
            t1 = time.perf_counter()
            var = chol.predict_variance(k, Xq)
            t2 = time.perf_counter()
            tp.append(t1 - t0)
            tv.append(t2 - t1)
        print(f"n={n} m={m} narrow={sk}: predict {1e3*min(tp):.2f} ms  variance {1e3*min(tv):.2f} ms", flush=True)
