"""Developer probe: FP64 GEMM rate over time (first launches vs sustained) to separate kernel quality from clock behaviour."""
import sys
import time

import torch

sys.path.insert(0, ".")
from friedrich_amd.device import Context

ctx = Context()
dev = torch.device("cuda", 0)
M, K = (int(sys.argv[1]) if len(sys.argv) > 1 else 16384), (int(sys.argv[2]) if len(sys.argv) > 2 else 1024)
A = torch.randn((K, M), dtype=torch.float64, device=dev).t()
C = torch.zeros((M, M), dtype=torch.float64, device=dev).t()
fl = 2.0 * M * M * K
t_begin = time.perf_counter()
for burst in range(4):
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 0.5:
        for _ in range(4):
            ctx.gemm(A, A, C=C, trans_b=True, alpha=-1e-9, beta=1.0)
        ctx.synchronize()
        n += 4
    dt = time.perf_counter() - t0
    print(f"t={time.perf_counter()-t_begin:5.1f}s  {n*fl/dt/1e12:.2f} TF/s", flush=True)
