"""Developer probe: the one-launch panel-row solve (gemm_f64.hip: rows_solve16_kernel / rows_solve_kernel) alone on the chip --
the sharded schedule's R1 and slice solves.   FRIEDRICH_AMD_ROWS_SOLVE16=0/1 rows_solve_probe.py [rows ...]"""
import ctypes
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from friedrich_amd import _capi as C
from friedrich_amd.device import Context

ctx = Context()
lib = ctx.lib
lib.fr_panel_rows_solve.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                    ctypes.c_void_p]
dev = torch.device("cuda:0")
kb = 512
rng = np.random.default_rng(0)
A = rng.standard_normal((kb, kb))
Lh = np.linalg.cholesky(A @ A.T + kb * np.eye(kb))
W = np.stack([np.linalg.inv(Lh[128 * s:128 * s + 128, 128 * s:128 * s + 128]) for s in range(4)])  # [s][n][k] -> stored column-major per block
Ld = torch.from_numpy(np.asfortranarray(Lh).T.copy()).to(dev)            # column-major L: element (i, j) at i + j * kb
Wd = torch.from_numpy(np.stack([np.asfortranarray(w).T.copy() for w in W])).to(dev)
for rows in [int(a) for a in sys.argv[1:]] or [512, 1024, 4096]:
    S0 = rng.standard_normal((rows, kb))
    want = np.linalg.solve(Lh, S0.T).T  # S L^-T
    Sd = torch.from_numpy(np.asfortranarray(S0).T.copy()).to(dev)
    def run():
        st = lib.fr_panel_rows_solve(ctx.h, Sd.data_ptr(), rows, rows, Ld.data_ptr(), kb, kb, Wd.data_ptr())
        assert st == 0
    run()
    ctx.synchronize()
    got = Sd.cpu().numpy().T
    err = np.max(np.abs(got - want)) / np.max(np.abs(want))
    best = 1e9
    for _ in range(20):
        Sd.copy_(torch.from_numpy(np.asfortranarray(S0).T.copy()))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run()
        ctx.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"rows={rows}: {1e6 * best:.1f} us (launch + synchronise), rel. error vs numpy {err:.1e}")
