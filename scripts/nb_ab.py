"""Developer probe: fit time against the outer block size nb, same process (box-to-box variation is ~3 %)."""
import sys
import time

sys.path.insert(0, ".")
from friedrich_amd import synth
from friedrich_amd.device import Context

ctx = Context()
NBS = [int(a) for a in sys.argv[2].split(",")] if len(sys.argv) > 2 else [256, 384, 512, 640, 768, 1024]
for n in [int(a) for a in sys.argv[1].split(",")]:
    X, y, Xq = synth.make_problem(n, 16, cfg=4, m=1024)
    ls = ctx.mean_pairwise_distance(X)
    hp = synth.default_hyperparameters(X, y, ls)
    k = ("squared_exp", hp["ls"], hp["ampl"])
    chol = ctx.cholesky_from_inputs(k, X, hp["noise"], capacity_hint=n)
    for rnd in range(2):
        for nb in NBS:
            ctx.set_option("nb", nb)
            ts = []
            for rep in range(3):
                t0 = time.perf_counter()
                chol.refactor(k, hp["noise"])
                ts.append(time.perf_counter() - t0)
            print(f"n={n} round {rnd} nb={nb}: fit min {1e3*min(ts):.1f} ms", flush=True)
    chol.free()
