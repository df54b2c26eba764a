import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from friedrich_amd import synth
from friedrich_amd.device import Context
ctx = Context()
for n, reps in ((32768, 300), (2049, 2000), (36864, 150)):
    d = 8
    X, y, Xq = synth.make_problem(n, d, cfg=4, m=16)
    k = ("squared_exp", 1.1, 0.3)
    chol = ctx.cholesky_from_inputs(k, X, 0.3, capacity_hint=n)
    ref16 = chol.predict_variance(k, Xq); ref1 = chol.predict_mean(k, y, Xq[:1]); ref16m = chol.predict_mean(k, y, Xq)
    t0 = time.time()
    bad = 0
    for i in range(reps):
        if not np.array_equal(chol.predict_variance(k, Xq), ref16): bad += 1
        if not np.array_equal(chol.predict_mean(k, y, Xq[:1]), ref1): bad += 1
        if not np.array_equal(chol.predict_mean(k, y, Xq), ref16m): bad += 1
    print(f"n={n}: {reps} repetitions of three predicts, mismatches {bad}, retries {ctx.counter('solve_retries')}, {time.time()-t0:.1f} s", flush=True)
    chol.free()
