"""Developer probe: full-mode GEMM on exactly the operand layout of a trailing update (panel and C inside one
ld = 32768 allocation) vs separate allocations -- isolates layout from the lower-mode tile set."""
import sys
import time

import torch

sys.path.insert(0, ".")
from friedrich_amd.device import Context

ctx = Context()
dev = torch.device("cuda", 0)
n, nb = 32768, 1024
W = torch.zeros((n, n), dtype=torch.float64, device=dev).t()  # column-major n x n, ld = n
W[:, :nb].normal_()


def rate(A, C, M, label):
    ts = []
    for rep in range(5):
        t0 = time.perf_counter()
        ctx.gemm(A, A, C=C, trans_b=True, alpha=-1e-9, beta=1.0)
        ctx.synchronize()
        ts.append(time.perf_counter() - t0)
    print(f"{label}: {2.0*M*M*nb/min(ts)/1e12:.2f} TF/s")


M = n - 2 * nb
rate(W[2 * nb:, :nb], W[2 * nb:, 2 * nb:], M, f"full-mode, panel and C inside the factor buffer (M={M})")
A2 = torch.randn((nb, M), dtype=torch.float64, device=dev).t()
rate(A2, W[2 * nb:, 2 * nb:], M, "full-mode, separate panel, C inside the factor buffer")
M2 = 16384
C2 = torch.zeros((M2, M2), dtype=torch.float64, device=dev).t()
rate(W[2 * nb:2 * nb + M2, :nb], C2, M2, f"full-mode, panel inside the factor buffer, separate C (M={M2})")

# the lower-triangular tile set on the same operands, with the tile orders of option gemm_tile
ctx.set_option("gemm_lower_probe", 1)
for mode, name in ((0, "row by row (default)"), (5, "column by column"), (3, "8 x 8 super-tiles")):
    ctx.set_option("gemm_tile", mode)
    ts = []
    for rep in range(5):
        t0 = time.perf_counter()
        ctx.gemm(W[2 * nb:, :nb], W[2 * nb:, :nb], C=W[2 * nb:, 2 * nb:], trans_b=True, alpha=-1e-9, beta=1.0)
        ctx.synchronize()
        ts.append(time.perf_counter() - t0)
    print(f"lower-mode, {name}: {1.0*M*(M+1)*nb/min(ts)/1e12:.2f} TF/s")
