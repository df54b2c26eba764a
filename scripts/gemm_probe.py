"""Developer probe: time the SYRK-shaped NT GEMM (C -= A B^T) on device-resident operands."""
import sys
sys.path.insert(0, ".")
import torch
from friedrich_amd.device import Context
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
K = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dbg = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ctx = Context()
ctx.set_option("gemm_tile", dbg)
dev = torch.device("cuda:0")
A = torch.randn(K, M, dtype=torch.float64, device=dev).t()      # M x K column-major
B = torch.randn(K, M, dtype=torch.float64, device=dev).t()
C = torch.randn(M, M, dtype=torch.float64, device=dev).t()
for rep in range(3):
    ctx.profile_reset(); ctx.profile_enable(True)
    for _ in range(5):
        ctx.gemm(A, B, C, trans_b=True, alpha=-1.0, beta=1.0)
    p = ctx.profile()["gemm_solve"]
    ctx.profile_enable(False)
print(f"M=N={M} K={K} dbg={dbg}: {p['flops']/p['ms']/1e9:.2f} TF/s  ({p['ms']/p['launches']:.3f} ms/launch)")
