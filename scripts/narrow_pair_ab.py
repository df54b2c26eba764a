"""Developer probe: narrow persistent solves with 16 vs 32 right-hand sides per column group (option narrow_pair_min)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from friedrich_amd import synth
from friedrich_amd.device import Context
ctx = Context()
ctx.set_option("narrow_batched_max", 1024)
for n in [int(a) for a in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["6144", "32768"])]:
    X, y, Xq = synth.make_problem(n + 512, 16, cfg=4, m=1024)
    ls = ctx.mean_pairwise_distance(X)
    hp = synth.default_hyperparameters(X, y, ls)
    k = ("squared_exp", hp["ls"], hp["ampl"])
    chol = ctx.cholesky_from_inputs(k, X[:n], hp["noise"], capacity_hint=n + 512)
    for m in (32, 64, 128, 256, 512):
        q = np.asfortranarray(Xq[:m])
        line, ref = f"n={n} m={m}:", None
        for pm in (0, 32):
            ctx.set_option("narrow_pair_min", pm)
            v = chol.predict_variance(k, q); p = chol.predict_mean(k, y[:n], q); ctx.synchronize()
            ref = ref or (v, p)
            err = max(float(np.max(np.abs(v - ref[0]))), float(np.max(np.abs(p - ref[1]))))
            tv = tp = 1e9
            for _ in range(3):
                t0 = time.perf_counter(); chol.predict_variance(k, q); ctx.synchronize(); tv = min(tv, time.perf_counter() - t0)
                t0 = time.perf_counter(); chol.predict_mean(k, y[:n], q); ctx.synchronize(); tp = min(tp, time.perf_counter() - t0)
            line += f"  [{'32' if pm else '16'} per group] variance {1e3*tv:.2f} predict {1e3*tp:.2f} ms (|d| {err:.0e})"
        print(line, flush=True)
    chol.free()
    if n <= 8192:
        for pm in (0, 32):
            ctx.set_option("narrow_pair_min", pm)
            best = 1e9
            for _ in range(3):
                chol = ctx.cholesky_from_inputs(k, X[:n], hp["noise"], capacity_hint=n + 512)
                ctx.synchronize()
                t0 = time.perf_counter(); chol.add_rows(k, np.asfortranarray(X[:n + 512]), 512, hp["noise"]); ctx.synchronize()
                best = min(best, time.perf_counter() - t0)
                chol.free()
            print(f"n={n}: add_samples(512) [{'32' if pm else '16'} per group] {1e3*best:.2f} ms", flush=True)
ctx.set_option("narrow_pair_min", -1)
