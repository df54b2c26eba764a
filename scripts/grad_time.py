"""Developer probe: one optimizer iteration of fit_parameters (refactor + gradient terms, optimizer.rs:24-60 / 159-203) by N,
and the kernels behind it (run under rocprofv3 --kernel-trace --stats for the breakdown)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from friedrich_amd import synth
from friedrich_amd.device import Context
ctx = Context()
for n in [int(a) for a in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["2048", "8192", "16384"])]:
    X, y, Xq = synth.make_problem(n, 8, cfg=4, m=16)
    ls = ctx.mean_pairwise_distance(X); hp = synth.default_hyperparameters(X, y, ls)
    k = ("squared_exp", hp["ls"], hp["ampl"])
    chol = ctx.cholesky_from_inputs(k, X, hp["noise"], capacity_hint=n)
    chol.grad_terms(k, y, hp["noise"], True, 2); ctx.synchronize()
    tf = tg = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); chol.refactor(k, hp["noise"]); ctx.synchronize(); tf = min(tf, time.perf_counter() - t0)
        t0 = time.perf_counter(); chol.grad_terms(k, y, hp["noise"], True, 2); ctx.synchronize(); tg = min(tg, time.perf_counter() - t0)
    fl = 2.0 * n ** 3 / 3.0  # inverse from the factor: n^3 / 3 (triangular inverse) + n^3 / 3 (W^T W, lower half)
    print(f"n={n}: refactor {1e3*tf:.2f} ms, gradient terms {1e3*tg:.2f} ms ({fl/tg/1e12:.1f} TF/s counting 2 n^3 / 3)", flush=True)
    chol.free()
