"""Developer probe: fit time for sets of context options, same process, two interleaved rounds.
   optset_ab.py <n,n,...> "<name=v,name=v>;<name=v>;..."      (an empty set = the defaults; every option named anywhere is reset
                                                              to its value in RESET before each variant)"""
import statistics
import sys
import time

sys.path.insert(0, ".")
from friedrich_amd import synth
from friedrich_amd.device import Context

RESET = {"nb_big_rows": 22528, "cu_reserve": 1, "reserve_rows2_cu": 6144, "reserve_rows1_cu": 12288, "cu_reserve_min_rows": 4096, "reserve_rows1": 16384, "reserve_rows2": 8192, "reserve_rows4": 4096, "k4_flat": -1, "xcd_reserve": -1, "nb": 0, "xcd_reserve_big_rows": 0, "nb_switch_rows": 16384,
         "lookahead": 1, "splitk": 1}
ctx = Context()
variants = []
for spec in sys.argv[2].split(";"):
    opts = {}
    for kv in [s for s in spec.split(",") if s]:
        k, v = kv.split("=")
        opts[k] = int(v)
    variants.append((spec or "defaults", opts))
names = set(k for _, o in variants for k in o)
for n in [int(a) for a in sys.argv[1].split(",")]:
    d = 8 if n <= 8192 else 16
    X, y, Xq = synth.make_problem(n, d, cfg=4, m=64)
    ls = ctx.mean_pairwise_distance(X)
    hp = synth.default_hyperparameters(X, y, ls)
    k = ("matern2" if n == 16384 else "squared_exp", hp["ls"], hp["ampl"])
    chol = ctx.cholesky_from_inputs(k, X, hp["noise"], capacity_hint=n)
    ref = None
    times = {name: [] for name, _ in variants}
    same = {}
    for rnd in range(2):
        for name, opts in variants:
            for o in names:
                ctx.set_option(o, opts.get(o, RESET[o]))
            chol.refactor(k, hp["noise"])
            for rep in range(5):
                t0 = time.perf_counter()
                chol.refactor(k, hp["noise"])
                times[name].append(1e3 * (time.perf_counter() - t0))
            if n <= 8192:
                import numpy as np
                L = chol.l()
                if ref is None:
                    ref = L
                same[name] = bool(np.array_equal(L, ref))
    for name, _ in variants:
        t = times[name]
        print(f"n={n}  [{name}]: fit min {min(t):.3f} ms  median {statistics.median(t):.3f} ms" + (f"  bit-identical to the first variant: {same[name]}" if name in same else ""), flush=True)
    chol.free()
