"""Developer probe: where the diagonal-block server's time goes (option panel_debug)."""
import sys
import time
sys.path.insert(0, ".")
from friedrich_amd import synth
from friedrich_amd.device import Context

ctx = Context()
ctx.set_option("panel_debug", 1)
for n in [int(a) for a in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["2048", "8192", "16384", "32768"])]:
    d = 8 if n <= 8192 else 16
    X, y, _ = synth.make_problem(n, d, cfg=4)
    ls = ctx.mean_pairwise_distance(X)
    hp = synth.default_hyperparameters(X, y, ls)
    k = ("squared_exp", hp["ls"], hp["ampl"])
    chol = ctx.cholesky_from_inputs(k, X, hp["noise"], capacity_hint=n)
    for _ in range(2):
        t0 = time.perf_counter()
        chol.refactor(k, hp["noise"])
        t1 = time.perf_counter()
    print(f"n={n}: fit {1e3 * (t1 - t0):.2f} ms", flush=True)
    ctx.set_option("panel_debug", 2)
    chol.free()
