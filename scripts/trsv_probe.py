"""Developer probe: the single right-hand-side persistent solves by direction (device-resident b: no staging in the timing)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from friedrich_amd import synth
from friedrich_amd.device import Context

ctx = Context()
dev = torch.device("cuda", 0)
for n in [int(a) for a in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["8192", "16384", "32768"])]:
    d = 16
    X, y, _ = synth.make_problem(n, d, cfg=4)
    ls = ctx.mean_pairwise_distance(X)
    hp = synth.default_hyperparameters(X, y, ls)
    k = ("squared_exp", hp["ls"], hp["ampl"])
    chol = ctx.cholesky_from_inputs(k, X, hp["noise"], capacity_hint=n)
    b0 = torch.from_numpy(y).to(dev)
    b = b0.clone()

    def run(fn, reps=5):
        best = 1e9
        for _ in range(reps):
            b.copy_(b0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            ctx.synchronize()
            best = min(best, time.perf_counter() - t0)
        return 1e3 * best

    for trsv in (1, 0):
        ctx.set_option("trsv", trsv)
        f = run(lambda: chol.solve_lower(b))
        fb = run(lambda: chol.solve(b))
        gb = 4.0 * n * n / 1e9
        print(f"n={n} trsv={trsv}: forward {f:.3f} ms ({gb / f:.0f} GB/s)  forward+backward {fb:.3f} ms  => backward {fb - f:.3f} ms ({gb / max(fb - f, 1e-9):.0f} GB/s)",
              flush=True)
    ctx.set_option("trsv", 1)
    chol.free()

