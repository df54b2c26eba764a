"""Developer probe: gradient terms (fr_grad_terms) by kernel program -- single leaves have their own reduction kernels, composite
programs (Sum / Prod) run the generic stack machine: how much of an optimizer iteration is the reduction?"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from friedrich_amd import synth
from friedrich_amd.device import Context
ctx = Context()
for n in [int(a) for a in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["8192", "16384"])]:
    X, y, _ = synth.make_problem(n, 16, cfg=4)
    ls = ctx.mean_pairwise_distance(X); hp = synth.default_hyperparameters(X, y, ls)
    for name, k, npar in (("squared_exp", ("squared_exp", hp["ls"], hp["ampl"]), 2), ("matern2", ("matern2", hp["ls"], hp["ampl"]), 2),
                          ("rational_quadratic", ("rational_quadratic", 1.5, hp["ls"]), 2),
                          ("sum(se, matern1)", ("sum", ("squared_exp", hp["ls"], hp["ampl"]), ("matern1", hp["ls"], 0.3)), 4),
                          ("prod(matern2, se)", ("prod", ("matern2", hp["ls"], 1.0), ("squared_exp", 2 * hp["ls"], hp["ampl"])), 4)):
        chol = ctx.cholesky_from_inputs(k, X, max(hp["noise"], 0.05))
        ctx.profile_reset(); ctx.profile_enable(True)
        chol.grad_terms(k, y, hp["noise"], True, npar); ctx.synchronize()
        t0 = time.perf_counter(); chol.grad_terms(k, y, hp["noise"], True, npar); ctx.synchronize(); dt = time.perf_counter() - t0
        t1 = time.perf_counter(); chol.refactor(k, max(hp["noise"], 0.05)); ctx.synchronize(); dr = time.perf_counter() - t1
        p = ctx.profile(); ctx.profile_enable(False)
        print(f"n={n} {name:20s}: grad_terms {1e3*dt:8.2f} ms (reduce class {p['reduce']['ms']/2:7.2f} ms per call), refactor {1e3*dr:7.2f} ms (gram class {p['gram']['ms']:6.2f} ms)", flush=True)
        chol.free()
