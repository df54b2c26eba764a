"""Developer probe: A/B of the GEMM tile orders inside one process (option gemm_tile: 0 default = super-tiles for
full-mode only, 1 plain, 2 both modes, 3 lower-mode only)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from friedrich_amd import synth
from friedrich_amd.device import Context

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 512
ctx = Context()
MODES = [int(a) for a in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 1, 2, 3]
ctx.set_option("nb", nb)
X, y, Xq = synth.make_problem(n, 16, cfg=4, m=1024)
ls = ctx.mean_pairwise_distance(X)
hp = synth.default_hyperparameters(X, y, ls)
k = ("squared_exp", hp["ls"], hp["ampl"])
chol = ctx.cholesky_from_inputs(k, X, hp["noise"], capacity_hint=n)
for rnd in range(2):
    for mode in MODES:
        ctx.set_option("gemm_tile", mode)
        ts = []
        for rep in range(3):
            t0 = time.perf_counter()
            chol.refactor(k, hp["noise"])
            ts.append(time.perf_counter() - t0)
        ctx.profile_reset()
        ctx.profile_enable(True)
        chol.refactor(k, hp["noise"])
        pr = ctx.profile()
        ctx.profile_enable(False)
        print(f"round {rnd} order {mode}: fit min {1e3*min(ts):.1f} ms  med {1e3*sorted(ts)[1]:.1f} ms  "
              f"syrk {pr['syrk']['ms']:.1f} panel-gemm {pr['gemm_panel']['ms']:.1f} potf2 {pr['potf2']['ms']:.1f}", flush=True)
