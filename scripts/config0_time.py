"""Developer probe: BASELINE configs[0] -- GaussianProcess::default at N = 512, d = 1 through the C ABI (tests/host_mirror.py
stands in for friedrich's host-side ADAM loop): total time and time per optimizer iteration."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from friedrich_amd import synth
from friedrich_amd.device import Context
from host_mirror import DeviceGP
ctx = Context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
X, y, Xq = synth.make_problem(n, 1, cfg=1, m=64)
for rep in range(3):
    t0 = time.perf_counter()
    gp = DeviceGP.default(ctx, X, y)
    ctx.synchronize()
    t1 = time.perf_counter()
    mean = gp.predict(Xq)
    t2 = time.perf_counter()
    print(f"n={n}: default() {1e3*(t1-t0):.1f} ms, {gp.iterations} iterations -> {1e3*(t1-t0)/max(gp.iterations,1):.2f} ms per iteration; predict(64) {1e3*(t2-t1):.2f} ms", flush=True)
