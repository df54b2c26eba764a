"""Developer probe: the per-step terms of the sharded factorisation's critical-path model, measured on ONE
GPU through the C ABI with the shapes a rank sees at N = 32768, nb = 512, W = 8 ranks:
    D      the 512 x 512 diagonal block: factor_panel restricted to its own rows (fr_chol_from_matrix of a 512 x 512 matrix,
           device-resident; profile classes potf2 + gemm_panel give the kernel time, the wall clock the launch-bound time)
    R1     the next panel's diagonal row tile: 512 rows solved against D (4 sub-panels: product + inverse product)
    u1     512 x 512 x 512 lower update of the next diagonal block
    LA1    512 x 512 x 512 update of the R1 rows of the next panel
    LA2    (n - k) x 512 x 512 update of the rest of the next panel's block column
    slice  a rank's slice of the bulk rows: (n - k) / 8 rows solved against D
    syrk   a rank's share of the trailing update: 1 / 8 of the (n - k)^2 x 512 lower product
    copy   pack / unpack of a slice (device copy bandwidth)
The transfer terms of the model (fan-out of 2.5 + 2 MB, scatter and all-gather of the slices) cannot be measured on one GPU;
they are priced from the link rate in DESIGN.md."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from friedrich_amd.device import Context

ctx = Context()
dev = torch.device("cuda", 0)
n, nb, W = 32768, 512, 8


def cm(rows, cols):  # column-major device matrix
    return torch.randn((cols, rows), dtype=torch.float64, device=dev).t()


def best(fn, reps=20):
    fn(); ctx.synchronize()
    b = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ctx.synchronize(); b = min(b, time.perf_counter() - t0)
    return 1e6 * b


# D: factor a 512 x 512 SPD matrix resident on the device
Q = torch.randn((nb, nb), dtype=torch.float64, device=dev)
S = (Q @ Q.t() + nb * torch.eye(nb, dtype=torch.float64, device=dev)).t().contiguous().t()
def factor_d():
    c = ctx.cholesky_from_matrix(S)
    c.free()
print(f"D (512 x 512 diagonal block, incl. handle allocation and info read-back): {best(factor_d):.0f} us")
ctx.profile_reset(); ctx.profile_enable(True)
for _ in range(10):
    factor_d()
p = ctx.profile(); ctx.profile_enable(False)
print(f"   kernel time per factorisation: potf2 {1e3 * p['potf2']['ms'] / 10:.0f} us in {p['potf2']['launches'] // 10} launches, "
      f"panel products {1e3 * p['gemm_panel']['ms'] / 10:.0f} us in {p['gemm_panel']['launches'] // 10} launches")

A512, B512, C512 = cm(nb, nb), cm(nb, nb), cm(nb, nb)
print(f"u1 / LA1 (512 x 512 x 512 product): {best(lambda: ctx.gemm(A512, B512, C512, trans_b=True, alpha=-1.0, beta=1.0)):.0f} us")
# R1 / slice solves: 4 sub-panels, S_s <- (S_s - S_<s L^T) W_s^T
import ctypes
ctx.lib.fr_panel_rows_solve.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                        ctypes.c_int64, ctypes.c_void_p]
def solve_rows(rows):
    """the library's own one-launch kernel (round 5: through the developer hook fr_panel_rows_solve; rounds 3 and 4 priced this
    term with seven separate products).  Timings of random operands: the kernel's time does not depend on the values"""
    Sx, L, Wi = cm(rows, nb), cm(nb, nb), cm(128, 4 * 128)
    def run():
        st = ctx.lib.fr_panel_rows_solve(ctx.h, Sx.data_ptr(), rows, rows, L.data_ptr(), nb, nb, Wi.data_ptr())
        assert st == 0
    return best(run)
print(f"R1 (512 rows against D): {solve_rows(512):.0f} us")
for k in (0, 8192, 16384, 24576):
    rest = n - k - nb
    sl = -(-(rest // W) // 128) * 128
    P = cm(rest, nb)
    Cn = cm(rest, nb)
    la2 = best(lambda: ctx.gemm(P, P[:nb], Cn, trans_b=True, alpha=-1.0, beta=1.0), 5)
    src, dst = cm(sl, nb), cm(sl, nb)
    def do_copy():
        dst.copy_(src)
        torch.cuda.synchronize()
    cp = best(do_copy, 5)
    share = rest * rest * nb / W / 6.2e13 * 1e6
    print(f"k = {k}: rest {rest} rows | LA2 {la2:.0f} us | slice of {sl} rows: solve {solve_rows(sl):.0f} us, copy {cp:.0f} us "
          f"({2 * 8 * sl * nb / cp / 1e3:.0f} GB/s) | a rank's share of the trailing update at 62 TF/s: {share:.0f} us")
