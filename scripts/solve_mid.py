"""Developer probe (for rocprofv3 --kernel-trace): ONE forward solve of m right-hand sides against a factor of n rows, after two
warm-up solves.   solve_mid.py n m [option=value ...]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from friedrich_amd import synth
from friedrich_amd.device import Context
n, m = int(sys.argv[1]), int(sys.argv[2])
ctx = Context(); dev = torch.device("cuda", 0)
for o in sys.argv[3:]:
    ctx.set_option(o.split("=")[0], int(o.split("=")[1]))
X, y, _ = synth.make_problem(n, 8, cfg=4)
ls = ctx.mean_pairwise_distance(X); hp = synth.default_hyperparameters(X, y, ls)
k = ("squared_exp", hp["ls"], hp["ampl"])
chol = ctx.cholesky_from_inputs(k, X, hp["noise"], capacity_hint=n)
B0 = torch.randn((m, n), dtype=torch.float64, device=dev).t()
B = B0.clone()
for rep in range(3):
    B.copy_(B0); torch.cuda.synchronize()
    t0 = time.perf_counter(); chol.solve_lower(B); ctx.synchronize(); dt = time.perf_counter() - t0
print(f"n={n} m={m}: forward solve {1e3 * dt:.3f} ms ({n * n * m / dt / 1e12:.1f} TF/s)")
