import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from friedrich_amd import synth
from friedrich_amd.device import Context
ctx = Context(); dev = torch.device("cuda", 0)
n, m = 32768, 4096
X, y, _ = synth.make_problem(n, 16, cfg=4)
ls = ctx.mean_pairwise_distance(X); hp = synth.default_hyperparameters(X, y, ls)
k = ("squared_exp", hp["ls"], hp["ampl"])
chol = ctx.cholesky_from_inputs(k, X, hp["noise"], capacity_hint=n)
B0 = torch.randn((m, n), dtype=torch.float64, device=dev).t()
B = B0.clone()
def run(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        B.copy_(B0); torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); ctx.synchronize(); best = min(best, time.perf_counter() - t0)
    return 1e3 * best
f = run(lambda: chol.solve_lower(B)); fb = run(lambda: chol.solve(B))
fl = n * n * m
print(f"forward {f:.1f} ms ({fl / f / 1e9:.1f} TF/s), forward + backward {fb:.1f} ms => backward {fb - f:.1f} ms ({fl / (fb - f) / 1e9:.1f} TF/s)")
