"""Developer probe: does the leading dimension of the factor matter (ld = 32768 doubles = a 256 KiB stride between columns)?
Fit + predict(m) at N with capacity_hint = N, N + 128, N + 256, N + 384 in one process, interleaved.   ld_pad_ab.py [N] [m]"""
import statistics
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from friedrich_amd import synth
from friedrich_amd.device import Context

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
m = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ctx = Context()
d = 8 if n <= 8192 else 16
X, y, Xq = synth.make_problem(n, d, cfg=3, m=m)
ls = ctx.mean_pairwise_distance(X)
hp = synth.default_hyperparameters(X, y, ls)
k = ("squared_exp", hp["ls"], hp["ampl"])
pads = [0, 128, 256, 384]
fits = {p: [] for p in pads}
preds = {p: [] for p in pads}
for rnd in range(2):
    for p in pads:
        chol = ctx.cholesky_from_inputs(k, X, hp["noise"], capacity_hint=n + p)
        for rep in range(3):
            t0 = time.perf_counter()
            chol.refactor(k, hp["noise"])
            fits[p].append(1e3 * (time.perf_counter() - t0))
        for rep in range(2):
            t0 = time.perf_counter()
            chol.predict_mean(k, y - hp["prior"], Xq, np.full(m, hp["prior"]))
            preds[p].append(1e3 * (time.perf_counter() - t0))
        chol.free()
for p in pads:
    print(f"n={n} capacity n+{p}: fit min {min(fits[p]):.2f} median {statistics.median(fits[p]):.2f} ms   predict(m={m}) min {min(preds[p]):.2f} median {statistics.median(preds[p]):.2f} ms", flush=True)
