"""Developer probe: fit time against the padding of the factor's leading dimension (option ld_pad), same process."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from friedrich_amd import synth
from friedrich_amd.device import Context

ctx = Context()
ctx.set_option("nb", 512)
for n in [int(a) for a in sys.argv[1].split(",")]:
    X, y, Xq = synth.make_problem(n, 16, cfg=4, m=1024)
    ls = ctx.mean_pairwise_distance(X)
    hp = synth.default_hyperparameters(X, y, ls)
    k = ("squared_exp", hp["ls"], hp["ampl"])
    for pad in (0, 64, 192, 0, 192):
        ctx.set_option("ld_pad", pad)
        chol = ctx.cholesky_from_inputs(k, X, hp["noise"], capacity_hint=n)
        ts = []
        for rep in range(3):
            t0 = time.perf_counter()
            chol.refactor(k, hp["noise"])
            ts.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        chol.predict_variance(k, Xq)
        tv = time.perf_counter() - t0
        print(f"n={n} ld_pad={pad}: fit min {1e3*min(ts):.1f} ms  variance(m=1024) {1e3*tv:.1f} ms", flush=True)
        chol.free()
