"""Developer probe: fit time with XCDs set aside for the panel stream.
    xcd_reserve_ab.py <n,n,...> <R,R,...>      R = option xcd_reserve (-1: automatic tiers, 0: off, 1..4: for the whole fit)"""
import sys
import time

sys.path.insert(0, ".")
import numpy as np
from friedrich_amd import synth
from friedrich_amd.device import Context

ctx = Context()
variants = [int(v) for v in sys.argv[2].split(",")]
for n in [int(a) for a in sys.argv[1].split(",")]:
    X, y, Xq = synth.make_problem(n, 16, cfg=4, m=64)
    ls = ctx.mean_pairwise_distance(X)
    hp = synth.default_hyperparameters(X, y, ls)
    k = ("squared_exp", hp["ls"], hp["ampl"])
    chol = ctx.cholesky_from_inputs(k, X, hp["noise"], capacity_hint=n)
    ref = chol.l() if n <= 8192 else None
    for rnd in range(2):
        for r in variants:
            ctx.set_option("xcd_reserve", r)
            ts = []
            for rep in range(4):
                t0 = time.perf_counter()
                chol.refactor(k, hp["noise"])
                ts.append(time.perf_counter() - t0)
            extra = ""
            if ref is not None:
                extra = f"  max |dL| {float(np.max(np.abs(chol.l() - ref))):.1e}"
            if rnd == 1:
                print(f"n={n} xcd_reserve={r}: fit min {1e3*min(ts):.2f} ms{extra}", flush=True)
    ctx.set_option("xcd_reserve", -1)
    chol.free()
