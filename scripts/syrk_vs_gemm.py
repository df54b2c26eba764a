"""Developer probe, one process: the full-mode GEMM alone vs the lower-mode SYRK launches of a factorisation
(look-ahead off: alone on the chip; look-ahead on: next to the panel stream)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from friedrich_amd import synth
from friedrich_amd.device import Context

ctx = Context()
dev = torch.device("cuda", 0)
M, K = 24576, 1024
A = torch.randn((K, M), dtype=torch.float64, device=dev).t()
C = torch.zeros((M, M), dtype=torch.float64, device=dev).t()
ts = []
for rep in range(6):
    t0 = time.perf_counter()
    ctx.gemm(A, A, C=C, trans_b=True, alpha=-1e-9, beta=1.0)
    ctx.synchronize()
    ts.append(time.perf_counter() - t0)
print(f"full-mode GEMM {M}^2 x {K}: {2.0*M*M*K/min(ts)/1e12:.2f} TF/s")
del A, C
n = 32768
X, y, _ = synth.make_problem(n, 16, cfg=4, m=8)
k = ("squared_exp", 1.6, 0.2)
ctx.set_option("nb", 1024)
chol = ctx.cholesky_from_inputs(k, X, 0.05, capacity_hint=n)
for la in (0, 1, 0, 1):
    ctx.set_option("lookahead", la)
    chol.refactor(k, 0.05)
    ctx.profile_reset()
    ctx.profile_enable(True)
    t0 = time.perf_counter()
    chol.refactor(k, 0.05)
    dt = time.perf_counter() - t0
    p = ctx.profile()["syrk"]
    ctx.profile_enable(False)
    print(f"lookahead={la}: fit {1e3*dt:.1f} ms; SYRK launches {p['launches']}, {p['flops']/p['ms']/1e9:.2f} TF/s")
