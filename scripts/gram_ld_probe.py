"""Developer probe: Gram assembly time against the leading dimension of the output (power-of-two strides)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from friedrich_amd import synth
from friedrich_amd.device import Context

ctx = Context()
n, d = 16384, 16
X, y, _ = synth.make_problem(n, d, cfg=4)
dev = torch.device("cuda", 0)
X_d = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev).t()
k = ("squared_exp", 1.6, 0.2)
for ld in (n, n + 64, n + 128, n + 192, n + 1024):
    buf = torch.empty((n, ld), dtype=torch.float64, device=dev)  # row i of buf = column i of the matrix
    out = buf.t()[:n, :]  # n x n view, stride(0) = 1, stride(1) = ld
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        ctx.gram(k, X_d, X_d, out=out)
        ctx.synchronize()
        dt = time.perf_counter() - t0
    print(f"n={n} ld={ld}: {1e3*dt:.2f} ms  ({8.0*n*n/dt/1e9:.0f} GB/s written)")
