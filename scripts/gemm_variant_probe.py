"""Developer probe: the four operand-layout variants of the FP64 GEMM on one deep-K shape (C -= op(A) op(B))."""
import sys
import time

import torch

sys.path.insert(0, ".")
from friedrich_amd.device import Context

ctx = Context()
dev = torch.device("cuda", 0)
M, N, K = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (16384, 4096, 16384)))


def cm(rows, cols):
    return torch.randn((cols, rows), dtype=torch.float64, device=dev).t()  # rows x cols, column-major


C = cm(M, N)
for ta in (False, True):
    for tb in (False, True):
        A = cm(K, M) if ta else cm(M, K)
        B = cm(N, K) if tb else cm(K, N)
        torch.cuda.synchronize()
        ts = []
        for rep in range(4):
            t0 = time.perf_counter()
            ctx.gemm(A, B, C=C, trans_a=ta, trans_b=tb, alpha=-1.0, beta=1.0)
            ctx.synchronize()
            ts.append(time.perf_counter() - t0)
        # a_kmajor = trans_a (element (m,k) at A[k + m ld]);  b_kmajor = not trans_b
        print(f"M={M} N={N} K={K} a_kmajor={int(ta)} b_kmajor={int(not tb)}: {2.0*M*N*K/min(ts)/1e12:.2f} TF/s")
