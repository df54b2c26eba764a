"""Developer probe (round 6): the resident panel chain (potf2.hip: panel_chain_kernel, option panel_chain) against the chain of
launches it replaces, in one process: factor agreement, time per fit, launches taken.
    python scripts/panel_chain_probe.py [sizes] [--mode=1|2]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from friedrich_amd import synth
from friedrich_amd.device import Context

args = [a for a in sys.argv[1:] if not a.startswith("--")]
opts = dict(a[2:].split("=") for a in sys.argv[1:] if a.startswith("--"))
sizes = [int(a) for a in args[0].split(",")] if args else [256, 384, 512, 640, 1000, 1024, 2048, 4096]
pc = int(opts.get("mode", 1))
ctx = Context()
for n in sizes:
    d = 8
    X, y, _ = synth.make_problem(n, d, cfg=4)
    ls = ctx.mean_pairwise_distance(X)
    hp = synth.default_hyperparameters(X, y, ls)
    k = ("squared_exp", hp["ls"], hp["ampl"])
    res = {}
    for mode in (0, pc):
        ctx.set_option("panel_chain", mode)
        c0 = ctx.counter("panel_chain_launches")
        chol = ctx.cholesky_from_inputs(k, X, hp["noise"], capacity_hint=n)
        ctx.synchronize()
        best = 1e9
        for _ in range(7):
            t0 = time.perf_counter()
            chol.refactor(k, hp["noise"])
            ctx.synchronize()
            best = min(best, time.perf_counter() - t0)
        L = chol.l()
        res[mode] = (best, L, ctx.counter("panel_chain_launches") - c0, chol.conditioning()[0])
        chol.free()
    L0, L1 = res[0][1], res[pc][1]
    err = np.max(np.abs(L0 - L1)) / np.max(np.abs(L0))
    print(f"n={n}: launch chain {1e3 * res[0][0]:.3f} ms, resident chain {1e3 * res[pc][0]:.3f} ms ({res[pc][2]} chain launches in 8 fits), "
          f"max rel diff of L {err:.2e}, estimates {res[0][3]:.3g} / {res[pc][3]:.3g}, fallbacks {ctx.counter('panel_chain_fallbacks')}", flush=True)
ctx.close()
