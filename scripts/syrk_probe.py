"""Developer probe: one Cholesky at a given size (for rocprofv3 kernel-trace / PMC passes)."""
import sys
sys.path.insert(0, ".")
from friedrich_amd import synth
from friedrich_amd.device import Context
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 256
ctx = Context()
ctx.set_option("nb", nb)
X, y, _ = synth.make_problem(n, 16, cfg=4)
k = ("squared_exp", 1.6, 0.2)
chol = ctx.cholesky_from_inputs(k, X, 0.05)
chol.refactor(k, 0.05)
print("done", chol.info())
