"""Developer probe: fit time with XCDs set aside for the panel stream.
    xcd_reserve_ab.py <n,n,...> <crit:R1:rest1:R2:rest2,...>
crit = option panel_crit (10 + crit: option panel_rl as well), R1 XCDs while the trailing matrix has <= rest1 rows (0: always), R2 XCDs below rest2 rows."""
import sys
import time

sys.path.insert(0, ".")
import numpy as np
from friedrich_amd import synth
from friedrich_amd.device import Context

ctx = Context()
variants = [tuple(int(x) for x in v.split(":")) for v in sys.argv[2].split(",")]
for n in [int(a) for a in sys.argv[1].split(",")]:
    X, y, Xq = synth.make_problem(n, 16, cfg=4, m=64)
    ls = ctx.mean_pairwise_distance(X)
    hp = synth.default_hyperparameters(X, y, ls)
    k = ("squared_exp", hp["ls"], hp["ampl"])
    chol = ctx.cholesky_from_inputs(k, X, hp["noise"], capacity_hint=n)
    ref = chol.l() if n <= 8192 else None
    for rnd in range(2):
        for (crit, r, rest, r2, rest2) in variants:
            ctx.set_option("panel_crit", crit % 10)
            ctx.set_option("panel_rl", crit // 10)
            ctx.set_option("xcd_reserve", r)
            ctx.set_option("xcd_reserve_rest", rest)
            ctx.set_option("xcd_reserve2", r2)
            ctx.set_option("xcd_reserve_rest2", rest2)
            ts = []
            for rep in range(4):
                t0 = time.perf_counter()
                chol.refactor(k, hp["noise"])
                ts.append(time.perf_counter() - t0)
            extra = ""
            if ref is not None:
                extra = f"  max |dL| {float(np.max(np.abs(chol.l() - ref))):.1e}"
            if rnd == 1:
                print(f"n={n} crit={crit} R1={r} rest1<={rest} R2={r2} rest2<={rest2}: fit min {1e3*min(ts):.2f} ms{extra}", flush=True)
    for o in ("xcd_reserve", "xcd_reserve2", "panel_crit", "panel_rl"):
        ctx.set_option(o, 0)
    chol.free()
