"""Developer probe: sample_at(256) on the factor of configs[4] (4096 rows + 8 x add_samples(512)), 20 calls, min / median.
   posterior_time.py [path of an alternative libfriedrich_amd.so]"""
import statistics
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from friedrich_amd import _capi

if len(sys.argv) > 1:
    _capi.LIB_PATH = sys.argv[1]
from friedrich_amd import synth
from friedrich_amd.device import Context

n = 8192
ctx = Context()
X, y, Xq = synth.make_problem(n, 8, cfg=4, m=1024)
ls = ctx.mean_pairwise_distance(X)
hp = synth.default_hyperparameters(X, y, ls)
k = ("squared_exp", hp["ls"], hp["ampl"])
for handle in range(3):  # (the first handle warms the context's workspace pool; the later ones are what configs[4] times)
    g = ctx.cholesky_from_inputs(k, X[:4096], hp["noise"], capacity_hint=n)
    ta = []
    for hi in range(4096 + 512, n + 1, 512):
        t0 = time.perf_counter()
        g.add_rows(k, X[:hi], 512, hp["noise"])
        ta.append(1e3 * (time.perf_counter() - t0))
    ts = []
    for rep in range(20):
        t0 = time.perf_counter()
        g.posterior(k, y - hp["prior"], Xq[:256], np.full(256, hp["prior"]))
        ts.append(1e3 * (time.perf_counter() - t0))
    print(f"{_capi.LIB_PATH} handle {handle}: appends {sum(ta):.2f} ms ({' '.join('%.2f' % t for t in ta)}); sample_at(256) all calls "
          f"{' '.join('%.2f' % t for t in ts)}  min {min(ts):.3f} median {statistics.median(ts):.3f} ms")
    g.free()
