"""Developer probe (rocprofv3 target): a few fits at one size.  fit_only.py N [reps] [--opt name=value ...]"""
import sys
import time
sys.path.insert(0, ".")
from friedrich_amd import synth
from friedrich_amd.device import Context

args = [a for a in sys.argv[1:] if not a.startswith("--")]
opts = [a[2:] for a in sys.argv[1:] if a.startswith("--")]
n = int(args[0]) if args else 32768
reps = int(args[1]) if len(args) > 1 else 3
ctx = Context()
for o in opts:
    name, val = o.split("=")
    ctx.set_option(name, int(val))
d = 8 if n <= 8192 else 16
X, y, _ = synth.make_problem(n, d, cfg=4)
ls = ctx.mean_pairwise_distance(X)
hp = synth.default_hyperparameters(X, y, ls)
k = ("matern2" if n == 16384 else "squared_exp", hp["ls"], hp["ampl"])
chol = ctx.cholesky_from_inputs(k, X, hp["noise"], capacity_hint=n)
for _ in range(reps):
    t0 = time.perf_counter()
    chol.refactor(k, hp["noise"])
    print(f"n={n} {opts}: fit {1e3 * (time.perf_counter() - t0):.2f} ms", flush=True)
