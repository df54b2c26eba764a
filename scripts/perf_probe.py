"""Developer probe: fit / predict timings and the per-kernel-class profile at a few sizes (not the judged bench)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from friedrich_amd import synth
from friedrich_amd.device import Context

sizes = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [4096, 8192, 16384]
d = int(sys.argv[2]) if len(sys.argv) > 2 else 16
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 256
m = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
ctx = Context()
ctx.set_option("nb", nb)
for n in sizes:
    X, y, Xq = synth.make_problem(n, d, cfg=4, m=m)
    ls = ctx.mean_pairwise_distance(X)
    hp = synth.default_hyperparameters(X, y, ls)
    k = ("squared_exp", hp["ls"], hp["ampl"])
    chol = ctx.cholesky_from_inputs(k, X, hp["noise"], capacity_hint=n)  # warm-up (allocations)
    for rep in range(2):
        ctx.profile_reset()
        ctx.profile_enable(rep == 1)
        t0 = time.perf_counter()
        chol.refactor(k, hp["noise"])
        t1 = time.perf_counter()
        mean = chol.predict_mean(k, y - hp["prior"], Xq, np.full(m, hp["prior"]))
        t2 = time.perf_counter()
        var = chol.predict_variance(k, Xq)
        t3 = time.perf_counter()
        print(f"n={n} d={d} nb={nb} prof={rep}: fit {1e3*(t1-t0):.1f} ms ({n**3/3/(t1-t0)/1e12:.2f} TF/s chol-equivalent)  "
              f"predict(m={m}) {1e3*(t2-t1):.1f} ms  variance {1e3*(t3-t2):.1f} ms", flush=True)
    prof = ctx.profile()
    for name, p in prof.items():
        if p["launches"]:
            print(f"    {name:11s} {p['ms']:9.2f} ms  {p['launches']:6d} launches  {p['flops']/max(p['ms'],1e-9)/1e9:9.2f} TF/s  "
                  f"{p['bytes']/max(p['ms'],1e-9)/1e9:8.2f} TB/s(alg)")
    ctx.profile_enable(False)
    chol.free()
