"""Developer probe (round 5): T threads, a context each, all fitting the SAME problem at the same time on ONE GPU and predicting --
every result must equal, bit for bit, the one a lone context produced before (kernels are deterministic whatever the placement;
what differs under contention is who gets which CU when).   contention_stress.py N T reps [name=value ...]"""
import sys
import threading

import numpy as np

sys.path.insert(0, ".")
from friedrich_amd import synth
from friedrich_amd.device import Context

n, T, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
opts = [a.split("=") for a in sys.argv[4:]]
m = 512
X, y, Xq = synth.make_problem(n, 16, cfg=4, m=m)
c0 = Context()
for o, v in opts:
    c0.set_option(o, int(v))
ls = c0.mean_pairwise_distance(X)
hp = synth.default_hyperparameters(X, y, ls)
k = ("squared_exp", hp["ls"], hp["ampl"])
yc = y - hp["prior"]
ch = c0.cholesky_from_inputs(k, X, hp["noise"])
want = ch.predict_mean(k, yc, Xq)
Lsum = None
if n <= 16384:
    L0 = ch.l()
ch.free()
c0.close()
bad = []
bar = threading.Barrier(T)


def worker(r):
    ctx = Context()
    for o, v in opts:
        ctx.set_option(o, int(v))
    for rep in range(reps):
        bar.wait()
        c = ctx.cholesky_from_inputs(k, X, hp["noise"])
        got = c.predict_mean(k, yc, Xq)
        if not np.array_equal(got, want):
            err = float(np.max(np.abs(got - want)) / np.max(np.abs(want)))
            where = ""
            if n <= 16384:
                L = c.l()
                d = np.argwhere(L != L0)
                where = f" factor differs in {len(d)} entries, first at {d[0].tolist() if len(d) else None}, last at {d[-1].tolist() if len(d) else None}"
            bad.append(f"thread {r} rep {rep}: predict deviates {err:.2e};{where}")
        c.free()
    ctx.close()


th = [threading.Thread(target=worker, args=(r,)) for r in range(T)]
[t.start() for t in th]
[t.join() for t in th]
print(f"n={n} T={T} reps={reps} {opts}: {len(bad)} bad of {T * reps}")
for b in bad[:12]:
    print("  ", b)
