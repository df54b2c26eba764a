"""Developer probe: fit time against a 0/1 context option, same process:  opt_ab.py <option> <n,n,...>"""
import sys
import time

sys.path.insert(0, ".")
from friedrich_amd import synth
from friedrich_amd.device import Context

ctx = Context()
opt = sys.argv[1]
for n in [int(a) for a in sys.argv[2].split(",")]:
    X, y, Xq = synth.make_problem(n, 16, cfg=4, m=64)
    ls = ctx.mean_pairwise_distance(X)
    hp = synth.default_hyperparameters(X, y, ls)
    k = ("squared_exp", hp["ls"], hp["ampl"])
    chol = ctx.cholesky_from_inputs(k, X, hp["noise"], capacity_hint=n)
    ref = chol.l() if n <= 8192 else None
    for rnd in range(2):
        for v in (0, 1):
            ctx.set_option(opt, v)
            ts = []
            for rep in range(4):
                t0 = time.perf_counter()
                chol.refactor(k, hp["noise"])
                ts.append(time.perf_counter() - t0)
            extra = ""
            if ref is not None:
                import numpy as np
                extra = f"  max |dL| {float(np.max(np.abs(chol.l() - ref))):.1e}"
            print(f"n={n} round {rnd} {opt}={v}: fit min {1e3*min(ts):.2f} ms{extra}", flush=True)
    chol.free()
