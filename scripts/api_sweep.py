"""Developer probe: every dense entry point of the C ABI once at a few sizes, with the flop count that bounds it -- to spot
the calls that run far from the matrix-core rate (the gradient terms did before their triangular products)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from friedrich_amd import synth
from friedrich_amd.device import Context
ctx = Context()
def best(fn, reps=2):
    fn(); ctx.synchronize()
    b = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ctx.synchronize(); b = min(b, time.perf_counter() - t0)
    return b
for n in [int(a) for a in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["8192", "32768"])]:
    X, y, Xq = synth.make_problem(n, 16, cfg=4, m=2048)
    ls = ctx.mean_pairwise_distance(X); hp = synth.default_hyperparameters(X, y, ls)
    k = ("squared_exp", hp["ls"], hp["ampl"])
    chol = ctx.cholesky_from_inputs(k, X, hp["noise"], capacity_hint=n + 2048)
    rows = []
    for m in (256, 2048):
        q = np.asfortranarray(Xq[:m]); pr = np.zeros(m)
        rows.append((f"posterior (sample_at) m={m}", best(lambda: chol.posterior(k, y, q, pr)), n * n * m + n * m * m + m ** 3 / 3))
        rows.append((f"predict_covariance m={m}", best(lambda: chol.predict_covariance(k, q)) if hasattr(chol, "predict_covariance") else float("nan"), n * n * m + n * m * m))
        rows.append((f"predict_mean_variance m={m}", best(lambda: chol.predict_mean_variance(k, y, q, pr)), 2 * n * n * m))
    B = np.asfortranarray(np.random.default_rng(0).standard_normal((n, 64)))
    rows.append(("solve 64 columns (host in / out)", best(lambda: chol.solve(B)), 2 * n * n * 64))
    if n <= 16384:
        rows.append(("inverse (host out)", best(lambda: chol.inverse(), 1), 2 * n ** 3 / 3))
        rows.append(("download_l", best(lambda: chol.l(), 1), 0))
    Xa = np.asfortranarray(np.vstack([X, X[:2048] + 0.37]))
    def add():
        c2 = ctx.cholesky_from_inputs(k, X, hp["noise"], capacity_hint=n + 2048); ctx.synchronize()
        t0 = time.perf_counter(); c2.add_rows(k, Xa, 2048, hp["noise"]); ctx.synchronize(); dt = time.perf_counter() - t0
        c2.free(); return dt
    add(); rows.append(("add_samples(2048)", min(add(), add()), n * n * 2048 + n * 2048 ** 2 + 2048 ** 3 / 3))
    for name, t, fl in rows:
        print(f"n={n}: {name:38s} {1e3*t:9.2f} ms" + (f"  {fl/t/1e12:6.1f} TF/s" if fl else ""), flush=True)
    chol.free()
