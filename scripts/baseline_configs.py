"""Developer probe: the timings the round-2 targets are stated in, in one warm process (not the judged bench).
    python scripts/baseline_configs.py [sizes] [--opt name=value ...]
prints one JSON line per size: fit (Gram + Cholesky) ms / TF/s, m = 1 and m = 16 predict latency, likelihood,
predict_variance m = 1024, and for N = 8192 the configs[4] add_samples sequence."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from friedrich_amd import synth
from friedrich_amd.device import Context

args = [a for a in sys.argv[1:] if not a.startswith("--")]
opts = [a[2:] for a in sys.argv[1:] if a.startswith("--")]
sizes = [int(a) for a in args[0].split(",")] if args else [4096, 8192, 16384, 32768]
ctx = Context()
for o in opts:
    name, val = o.split("=")
    ctx.set_option(name, int(val))


def timed(fn, reps=3):
    fn()
    ctx.synchronize()
    best = 1e30
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ctx.synchronize()
        best = min(best, time.perf_counter() - t0)
    return 1e3 * best


for n in sizes:
    d = 8 if n <= 8192 else 16
    kname = "matern2" if n == 16384 else "squared_exp"
    X, y, Xq = synth.make_problem(n, d, cfg=4, m=1024)
    ls = ctx.mean_pairwise_distance(X)
    hp = synth.default_hyperparameters(X, y, ls)
    k = (kname, hp["ls"], hp["ampl"])
    eps = 1e-2 * hp["noise"] ** 2 if n == 16384 else None
    yres = y - hp["prior"]
    chol = ctx.cholesky_from_inputs(k, X, hp["noise"], eps=eps, capacity_hint=n)
    rec = {"n": n, "d": d, "kernel": kname, "opts": opts}
    rec["fit_ms"] = timed(lambda: chol.refactor(k, hp["noise"], eps=eps))
    rec["fit_tflops"] = n ** 3 / 3.0 / (rec["fit_ms"] * 1e-3) / 1e12
    rec["cond_estimate"], rec["refined"] = chol.conditioning()
    for m in (1, 16, 1024):
        q, pq = np.asfortranarray(Xq[:m]), np.full(m, hp["prior"])
        rec[f"predict_m{m}_ms"] = timed(lambda: chol.predict_mean(k, yres, q, pq))
        rec[f"variance_m{m}_ms"] = timed(lambda: chol.predict_variance(k, q))
    rec["likelihood_ms"] = timed(lambda: chol.likelihood(k, yres, hp["noise"]))
    chol.free()
    if n == 8192:  # configs[4]: 4096 -> 8192 in 512-row chunks
        t_best = 1e30
        Xf = np.asfortranarray(X)  # the caller's matrix as the reference holds it (EMatrix: column-major with a capacity): its leading rows go by pointer + ld
        for _ in range(2):
            g = ctx.cholesky_from_inputs(k, Xf[:4096], hp["noise"], capacity_hint=n)
            ctx.synchronize()
            t0 = time.perf_counter()
            for hi in range(4096 + 512, n + 1, 512):
                g.add_rows(k, Xf[:hi], 512, hp["noise"])
            ctx.synchronize()
            t_best = min(t_best, time.perf_counter() - t0)
            g.free()
        rec["add_samples_8x512_ms"] = 1e3 * t_best
    print(json.dumps(rec), flush=True)
ctx.close()
