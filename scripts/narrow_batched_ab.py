"""Developer probe: predict_variance (one forward solve) by number of query points, persistent column-group solve vs the
recursive GEMM path."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from friedrich_amd import synth
from friedrich_amd.device import Context
ctx = Context()
for n in (4096, 8192, 16384, 32768):
    d = 16
    X, y, Xq = synth.make_problem(n, d, cfg=4, m=1024)
    ls = ctx.mean_pairwise_distance(X)
    hp = synth.default_hyperparameters(X, y, ls)
    k = ("squared_exp", hp["ls"], hp["ampl"])
    chol = ctx.cholesky_from_inputs(k, X, hp["noise"], capacity_hint=n)
    for m in (32, 64, 96, 128, 192, 256, 384, 512):
        q = np.asfortranarray(Xq[:m])
        res = []
        for thr in (0, 1024):
            ctx.set_option("narrow_batched_max", thr)
            chol.predict_variance(k, q); ctx.synchronize()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter(); chol.predict_variance(k, q); ctx.synchronize(); best = min(best, time.perf_counter() - t0)
            res.append(1e3 * best)
        print(f"n={n} m={m}: variance (forward solve) GEMM path {res[0]:.2f} ms, column groups {res[1]:.2f} ms", flush=True)
    chol.free()
