"""Developer probe: FP64 GEMM rate against the leading dimensions of C and of the operand panel (power-of-two strides)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from friedrich_amd.device import Context

ctx = Context()
dev = torch.device("cuda", 0)
n, K = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 512


def cm(rows, cols, ld):
    buf = torch.zeros((cols, ld), dtype=torch.float64, device=dev)
    return buf.t()[:rows, :]


for pad_c, pad_a in ((0, 0), (64, 0), (192, 0), (0, 192), (192, 192), (1024, 1024)):
    C = cm(n, n, n + pad_c)
    A = cm(n, K, n + pad_a)
    A.normal_()
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        ctx.gemm(A, A, C=C, trans_b=True, alpha=-1.0, beta=1.0)
        ctx.synchronize()
        dt = time.perf_counter() - t0
    print(f"M=N={n} K={K} ldc=n+{pad_c} lda=n+{pad_a}: {1e3*dt:.3f} ms  {2.0*n*n*K/dt/1e12:.2f} TF/s")
