"""Developer probe: add_samples of a FEW rows (1, 8, 64, 127) onto a large factor -- the Bayesian-optimisation loop of readme.md:7
(one new sample per iteration) -- and the predict that follows it.   add_few_probe.py [n,n,...]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from friedrich_amd import synth
from friedrich_amd.device import Context
ctx = Context()
for n in [int(a) for a in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["8192", "32768"])]:
    d = 8
    X, y, Xq = synth.make_problem(n + 1024, d, cfg=4, m=16)
    ls = ctx.mean_pairwise_distance(X[:n]); hp = synth.default_hyperparameters(X[:n], y[:n], ls)
    k = ("squared_exp", hp["ls"], hp["ampl"])
    chol = ctx.cholesky_from_inputs(k, X[:n], hp["noise"], capacity_hint=n + 1024)
    cur = n
    for add in (1, 1, 1, 8, 64, 127):
        t0 = time.perf_counter()
        chol.add_rows(k, X[:cur + add], add, hp["noise"]); ctx.synchronize()
        t1 = time.perf_counter()
        chol.predict_variance(k, Xq); ctx.synchronize()
        t2 = time.perf_counter()
        chol.predict_variance(k, Xq); ctx.synchronize()
        t3 = time.perf_counter()
        cur += add
        print(f"n={cur - add} + {add}: add_rows {1e3*(t1-t0):.2f} ms, first predict_variance(16) after it {1e3*(t2-t1):.2f} ms, second {1e3*(t3-t2):.2f} ms", flush=True)
    chol.free()
