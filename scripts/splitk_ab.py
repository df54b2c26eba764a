"""Developer probe: the split-K rule of launch_gemm (few result tiles, deep contraction) on the wide triangular solves:
predict / predict_variance by number of queries against option tuples  tiles:mink:target[:slice]."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from friedrich_amd import synth
from friedrich_amd.device import Context
ctx = Context()
rules = [tuple(int(x) for x in r.split(":")) for r in (sys.argv[2] if len(sys.argv) > 2 else "192:2048:384,256:512:512,256:512:512:128,256:256:512:128,256:512:1024:128").split(",")]
for n in [int(a) for a in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["16384", "32768"])]:
    X, y, Xq = synth.make_problem(n, 16, cfg=4, m=4096)
    ls = ctx.mean_pairwise_distance(X); hp = synth.default_hyperparameters(X, y, ls)
    k = ("squared_exp", hp["ls"], hp["ampl"])
    chol = ctx.cholesky_from_inputs(k, X, hp["noise"], capacity_hint=n)
    for m in (512, 1024, 2048, 4096):
        q = np.asfortranarray(Xq[:m])
        line, ref = f"n={n} m={m}:", None
        for rule in rules:
            tl, mk, tg = rule[:3]
            sl = rule[3] if len(rule) > 3 else 256
            ctx.set_option("splitk_tiles", tl); ctx.set_option("splitk_mink", mk); ctx.set_option("splitk_target", tg); ctx.set_option("splitk_slice", sl)
            v = chol.predict_variance(k, q); ctx.synchronize()
            ref = v if ref is None else ref
            tv = 1e9
            for _ in range(3):
                t0 = time.perf_counter(); chol.predict_variance(k, q); ctx.synchronize(); tv = min(tv, time.perf_counter() - t0)
            line += f"  [{tl}:{mk}:{tg}:{sl}] {1e3*tv:.2f} ms ({np.max(np.abs(v-ref))/np.max(np.abs(ref)):.0e})"
        print(line, flush=True)
    chol.free()
