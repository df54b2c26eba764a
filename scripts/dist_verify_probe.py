"""Developer probe (round 5): what bench.py --local-ranks W --verify does, step by step -- W thread-ranks on ONE GPU: sharded fit,
optionally the sharded gradient terms, then every rank's share of the predictions against a single-rank fit made by the rank
itself on a context of its own (concurrently with its peers: the chip is crowded).   dist_verify_probe.py N W [grad=1] [ref_cu=1]"""
import sys
import threading

import numpy as np

sys.path.insert(0, ".")
from friedrich_amd import synth
from friedrich_amd.device import Context

n, W = int(sys.argv[1]), int(sys.argv[2])
do_grad = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ref_cu = int(sys.argv[4]) if len(sys.argv) > 4 else 1
m = 4096
X, y, Xq = synth.make_problem(n, 16, cfg=4, m=m)
c0 = Context()
ls = c0.mean_pairwise_distance(X)
hp = synth.default_hyperparameters(X, y, ls)
c0.close()
k = ("squared_exp", hp["ls"], hp["ampl"])
out = [None] * W
bar = threading.Barrier(W)


def worker(r):
    ctx = Context()
    ctx.comm_init_local(777, r, W)
    ctx.set_option("dist_schedule", 2)
    chol = ctx.cholesky_from_inputs(k, X, hp["noise"])
    lo, hi = (m * r) // W, (m * (r + 1)) // W
    yc = y - hp["prior"]
    a = chol.predict_mean(k, yc, Xq[lo:hi])
    g = None
    if do_grad:
        g, _ = chol.grad_terms(k, yc, hp["noise"], True, 2)
    b = chol.predict_mean(k, yc, Xq[lo:hi])
    bar.wait()
    rc = Context()
    rc.set_option("cu_reserve", ref_cu)
    ref = rc.cholesky_from_inputs(k, X, hp["noise"])
    w = ref.predict_mean(k, yc, Xq[lo:hi])
    dl = None
    if n <= 16384:
        dl = float(np.max(np.abs(ref.l() - chol.l())))
    ref.free()
    rc.close()
    out[r] = (float(np.max(np.abs(a - w)) / np.max(np.abs(w))), float(np.max(np.abs(b - w)) / np.max(np.abs(w))), g, dl)
    chol.free()
    ctx.close()


th = [threading.Thread(target=worker, args=(r,)) for r in range(W)]
[t.start() for t in th]
[t.join() for t in th]
for r, o in enumerate(out):
    print(f"rank {r}: predict before grad_terms vs own single-rank fit {o[0]:.2e}, after {o[1]:.2e}, grad {o[2]}, max |dL| {o[3]}")
