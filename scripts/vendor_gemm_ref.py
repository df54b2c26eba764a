"""Developer probe: the vendor BLAS (through torch: rocBLAS / hipBLASLt) on the shapes of this library's dominant GEMMs, f64.
A reference point for DESIGN.md, not used by the library."""
import time

import torch

dev = torch.device("cuda", 0)


def bench(fn, flops, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return flops / min(ts) / 1e12, min(ts) * 1e3


for (M, N, K) in [(16384, 16384, 512), (16384, 16384, 1024), (32768, 32768, 1024), (16384, 4096, 16384)]:
    A = torch.randn((M, K), dtype=torch.float64, device=dev)
    B = torch.randn((N, K), dtype=torch.float64, device=dev)
    C = torch.zeros((M, N), dtype=torch.float64, device=dev)
    tf, ms = bench(lambda: torch.addmm(C, A, B.t(), beta=1.0, alpha=-1.0, out=C), 2.0 * M * N * K)
    print(f"vendor f64 C -= A B^T  M={M} N={N} K={K}: {tf:.1f} TF/s ({ms:.2f} ms)")
    del A, B, C
