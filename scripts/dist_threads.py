"""Developer probe: a sharded factorisation with W thread-ranks sharing ONE GPU (in-process transport), for kernel traces of
the sharded schedules' launches (rocprofv3 --kernel-trace --stats -- python scripts/dist_threads.py 8192 2 2) and as a smoke
run of the schedules at sizes the unit tests do not reach.   dist_threads.py N W schedule [reps]"""
import sys
import threading
import time

import numpy as np

sys.path.insert(0, ".")
from friedrich_amd import synth
from friedrich_amd.device import Context

n, W, sched = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
X, y, _ = synth.make_problem(n, 8, cfg=4)
c0 = Context()
ls = c0.mean_pairwise_distance(X)
hp = synth.default_hyperparameters(X, y, ls)
k = ("squared_exp", hp["ls"], hp["ampl"])
ref = c0.cholesky_from_inputs(k, X, hp["noise"])
Lref = ref.l()
ref.free()
c0.close()
out = [None] * W


def worker(r):
    ctx = Context()
    ctx.comm_init_local(4242, r, W)
    ctx.set_option("dist_schedule", sched)
    chol = ctx.cholesky_from_inputs(k, X, hp["noise"])
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        chol.refactor(k, hp["noise"])
        ts.append(time.perf_counter() - t0)
    out[r] = (min(ts), float(np.max(np.abs(chol.l() - Lref)) / np.max(np.abs(Lref))))
    chol.free()
    ctx.close()


th = [threading.Thread(target=worker, args=(r,)) for r in range(W)]
[t.start() for t in th]
[t.join() for t in th]
print(f"n={n} W={W} schedule={sched}: per-rank fit min {[round(1e3 * o[0], 2) for o in out]} ms (ranks share one GPU: not a scaling number), "
      f"max rel. deviation from the single-rank factor {max(o[1] for o in out):.1e}")
