"""Developer probe: mid-size right-hand-side counts (128 .. 1024) -- the recursive GEMM path against the persistent solve in
column groups of 16 right-hand sides and the 2048-row leaves (round 4 also measured 32- and 64-column groups with this script: history at 6fd4683).  predict_variance = cross-Gram + ONE forward solve +
epilogue; solve() = forward + backward on a device operand.   narrow_wide_ab.py [n,n,...]"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from friedrich_amd import synth
from friedrich_amd.device import Context
ctx = Context()
dev = torch.device("cuda", 0)
MODES = [("gemm", dict(narrow_batched_max=0, bigleaf_max=0, bigleaf_min=-1)),
         ("big", dict(narrow_batched_max=0, bigleaf_max=-1, bigleaf_min=1)),
         ("g16", dict(narrow_batched_max=4096, bigleaf_max=0, bigleaf_min=-1)),
         ("auto", dict(narrow_batched_max=-1, bigleaf_max=-1, bigleaf_min=-1))]
for n in [int(a) for a in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["4096", "8192", "16384", "32768"])]:
    X, y, Xq = synth.make_problem(n, 8, cfg=4, m=2048)
    ls = ctx.mean_pairwise_distance(X)
    hp = synth.default_hyperparameters(X, y, ls)
    k = ("squared_exp", hp["ls"], hp["ampl"])
    chol = ctx.cholesky_from_inputs(k, X, hp["noise"])
    for m in (32, 64, 128, 256, 512, 1024, 2048):
        q = torch.from_numpy(np.ascontiguousarray(Xq[:m].T)).to(dev).t()
        var = torch.empty((m,), dtype=torch.float64, device=dev)
        B0 = torch.randn((m, n), dtype=torch.float64, device=dev).t()
        line, ref = f"n={n} m={m}:", None
        for name, opts in MODES:
            for o, v in opts.items():
                ctx.set_option(o, v)
            chol.predict_variance(k, q, out=var); ctx.synchronize()
            B = B0.clone(); torch.cuda.synchronize(); chol.solve(B); ctx.synchronize()
            got = (var.cpu().numpy().copy(), B.cpu().numpy().copy())
            ref = ref or got
            err = max(float(np.max(np.abs(got[0] - ref[0])) / np.max(np.abs(ref[0]))), float(np.max(np.abs(got[1] - ref[1])) / np.max(np.abs(ref[1]))))
            tv = ts = 1e9
            for _ in range(3):
                t0 = time.perf_counter(); chol.predict_variance(k, q, out=var); ctx.synchronize(); tv = min(tv, time.perf_counter() - t0)
                B = B0.clone(); torch.cuda.synchronize()
                t0 = time.perf_counter(); chol.solve(B); ctx.synchronize(); ts = min(ts, time.perf_counter() - t0)
            line += f"  [{name}] var {1e3*tv:.2f} solve {1e3*ts:.2f} ms ({n*n*m/tv/1e12:.0f} TF/s, d {err:.0e})"
        print(line, flush=True)
    chol.free()
