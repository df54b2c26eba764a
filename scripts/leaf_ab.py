"""Developer probe: predict / predict_variance with 128-row and 512-row leaves of the triangular solves, same process."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from friedrich_amd import synth
from friedrich_amd.device import Context

ctx = Context()
for n in [int(a) for a in sys.argv[1].split(",")]:
    for m in (1024, 4096):
        X, y, Xq = synth.make_problem(n, 16, cfg=4, m=m)
        ls = ctx.mean_pairwise_distance(X)
        hp = synth.default_hyperparameters(X, y, ls)
        k = ("squared_exp", hp["ls"], hp["ampl"])
        chol = ctx.cholesky_from_inputs(k, X, hp["noise"], capacity_hint=n)
        ref = None
        for rnd in range(2):
            for leaf in (0, 1):
                ctx.set_option("leaf512", leaf)
                tp, tv = [], []
                for rep in range(2):
                    chol.refactor(k, hp["noise"])  # as in the bench: the cached block inverses are rebuilt every step
                    t0 = time.perf_counter()
                    mean = chol.predict_mean(k, y - hp["prior"], Xq, np.full(m, hp["prior"]))
                    t1 = time.perf_counter()
                    var = chol.predict_variance(k, Xq)
                    t2 = time.perf_counter()
                    tp.append(t1 - t0)
                    tv.append(t2 - t1)
                if ref is None:
                    ref = (mean.copy(), var.copy())
                dm = float(np.max(np.abs(mean - ref[0])) / np.max(np.abs(ref[0])))
                dv = float(np.max(np.abs(var - ref[1])) / np.max(np.abs(ref[1])))
                print(f"n={n} m={m} round {rnd} leaf512={leaf}: predict {1e3*min(tp):.1f} ms  variance {1e3*min(tv):.1f} ms  (rel diff {dm:.1e} {dv:.1e})", flush=True)
        chol.free()
