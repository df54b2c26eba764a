"""Developer probe: per-class profile of one factorisation with look-ahead on/off."""
import sys
sys.path.insert(0, ".")
from friedrich_amd import synth
from friedrich_amd.device import Context
n = int(sys.argv[1]); nb = int(sys.argv[2]); la = int(sys.argv[3])
ctx = Context(); ctx.set_option("nb", nb); ctx.set_option("lookahead", la)
X, y, _ = synth.make_problem(n, 16, cfg=4)
k = ("squared_exp", 1.6, 0.2)
chol = ctx.cholesky_from_inputs(k, X, 0.05)
import time
ctx.profile_reset(); ctx.profile_enable(True)
t0 = time.perf_counter(); chol.refactor(k, 0.05); t1 = time.perf_counter()
print(f"n={n} nb={nb} lookahead={la}: fit {1e3*(t1-t0):.1f} ms")
for name, p in ctx.profile().items():
    if p["launches"]:
        print(f"    {name:11s} {p['ms']:9.2f} ms  {p['launches']:6d} launches  avg {1e3*p['ms']/p['launches']:8.1f} us")
