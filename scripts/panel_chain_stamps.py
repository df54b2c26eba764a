"""Developer probe (round 6): where a resident panel-chain launch spends its time (FRIEDRICH_AMD_CHAIN_TS=1: 100 MHz stamps written by
workgroup 0 and by the first slab of every tile row; the XCDs' clocks are not synchronised, so only differences inside one workgroup
are quoted).    FRIEDRICH_AMD_CHAIN_TS=1 python scripts/panel_chain_stamps.py [n]"""
import os
import sys

os.environ["FRIEDRICH_AMD_CHAIN_TS"] = "1"
sys.path.insert(0, ".")
from friedrich_amd import synth
from friedrich_amd.device import Context

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = Context()
X, y, _ = synth.make_problem(n, 8, cfg=4)
ls = ctx.mean_pairwise_distance(X)
hp = synth.default_hyperparameters(X, y, ls)
k = ("squared_exp", hp["ls"], hp["ampl"])
chol = ctx.cholesky_from_inputs(k, X, hp["noise"], capacity_hint=n)
for rep in range(3):
    chol.refactor(k, hp["noise"])
    ctx.synchronize()
    ts = [ctx.counter(f"chain_ts:{i}") for i in range(128)]
    us = lambda a, b: (ts[b] - ts[a]) / 100.0
    J = min(4, n // 128)
    print(f"rep {rep}: workgroup 0 (us):", " | ".join(
        (f"wait D_{j} {us(8 * (j - 1) + 2, 8 * j):.1f}, " if j else "") + f"factor {us(8 * j, 8 * j + 1):.1f}, publish {us(8 * j + 1, 8 * j + 2):.1f}" for j in range(J)),
        f"| total {us(0, 8 * (J - 1) + 2):.1f}")
    for t in range(1, J):
        b = 32 + 16 * t
        print(f"   first slab of tile row {t}:", " | ".join(
            f"q={q}: product+store+publish {us(b + 4 * q, b + 4 * q + 1):.1f}, wait row {us(b + 4 * q + 1, b + 4 * q + 2):.1f}, product(+store+publish) {us(b + 4 * q + 2, b + 4 * q + 3):.1f}"
            + (f", then until W_{q + 1} {us(b + 4 * q + 3, b + 4 * q + 4):.1f}" if q + 1 < t else "") for q in range(t)))
ctx.close()
