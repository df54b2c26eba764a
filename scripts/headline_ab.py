"""Round-5 A/Bs of the headline fit (N = 32768, d = 16, RBF) in ONE process on ONE box (box-to-box spread is 3-4 %):
   headline_ab.py [n]     -> one line per variant: fit min / median over the repetitions of two interleaved rounds, and the
                             trailing update's in-situ rate (HIP events inside the library, as bench.py's roofline)
Variants: outer block size 1024 / 1536 / 2048 (K of the trailing update); one XCD set aside for the panel stream while the
panels are 1024 wide (option xcd_reserve_big_rows); the flat diagonal-block kernel on the look-ahead stream everywhere
(option k4_flat = 1: today only where the kernel has its CU to itself) and nowhere (0)."""
import statistics
import sys
import time

sys.path.insert(0, ".")
from friedrich_amd import synth
from friedrich_amd.device import Context

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
ctx = Context()
X, y, Xq = synth.make_problem(n, 16, cfg=4, m=64)
ls = ctx.mean_pairwise_distance(X)
hp = synth.default_hyperparameters(X, y, ls)
k = ("squared_exp", hp["ls"], hp["ampl"])
chol = ctx.cholesky_from_inputs(k, X, hp["noise"], capacity_hint=n)
DEFAULTS = {"nb": 0, "xcd_reserve_big_rows": 0, "k4_flat": -1, "nb_switch_rows": 16384, "nb_big_rows": 22528}
VARIANTS = [
    ("default (panels 2048 wide above 22528 rows, 1024 above 16384, then 512)", {}),
    ("nb 1024 -> 512 below 16384 rows (nb_big_rows = 0: the default before)", {"nb_big_rows": 0}),
    ("nb = 1536", {"nb": 1536}),
    ("nb = 2048", {"nb": 2048}),
    ("nb = 2048, 512 below 8192 rows", {"nb": 2048, "nb_switch_rows": 8192}),
    ("one XCD set aside below 24576 rows at nb = 1024", {"xcd_reserve_big_rows": 24576}),
    ("one XCD set aside for every 1024-panel", {"xcd_reserve_big_rows": 1 << 30}),
    ("k4_flat = 1 (flat diagonal-block kernel next to the trailing update too)", {"k4_flat": 1}),
    ("k4_flat = 0 (staged kernel everywhere)", {"k4_flat": 0}),
]
times = {name: [] for name, _ in VARIANTS}
rates = {name: [] for name, _ in VARIANTS}
for rnd in range(2):
    for name, opts in VARIANTS:
        for o, v in {**DEFAULTS, **opts}.items():
            ctx.set_option(o, v)
        chol.refactor(k, hp["noise"])
        ctx.profile_reset()
        ctx.profile_enable(True, classes=["syrk"])
        for rep in range(3):
            t0 = time.perf_counter()
            chol.refactor(k, hp["noise"])
            times[name].append(1e3 * (time.perf_counter() - t0))
        p = ctx.profile()["syrk"]
        ctx.profile_enable(False)
        rates[name].append(p["flops"] / max(p["ms"], 1e-9) / 1e9)
for name, _ in VARIANTS:
    t = times[name]
    print(f"n={n}  {name}: fit min {min(t):.1f} ms  median {statistics.median(t):.1f} ms   trailing update {statistics.mean(rates[name]):.1f} TF/s "
          f"(frac {statistics.mean(rates[name]) / 78.6:.3f})", flush=True)
chol.free()
