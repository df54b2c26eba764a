#!/usr/bin/env python3
"""Generate tests/golden/golden_v2.json: BASELINE.json configs[0] -- N = 512, d = 1, RBF, `GaussianProcess::default`
(heuristics + constant-prior fit + scaled ADAM loop, <= 100 iterations), 64 predictions.

The reference cannot be built here (no Rust toolchain) and holds no vectors for this path, so the values come from the
CPU oracle (oracle/, statement-by-statement restatement; PARITY UNPINNED) and are cross-checked at generation against
dense numpy algebra for the final model.  Inputs are the deterministic SplitMix64 set of SURVEY.md section 8d
(friedrich_amd/synth.py, cfg = 1), so the fixture only needs the outputs.

    python tests/golden/make_golden_v2.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from friedrich_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    n, d, m, cfg = 512, 1, 64, 1
    X, y, Xq = synth.make_problem(n, d, cfg=cfg, m=m)
    O.set_threads(0)
    gp = O.OracleGP.default(X, y)
    mean, var = gp.predict(Xq), gp.predict_variance(Xq)
    # independent check of the final model: dense algebra on the fitted hyper-parameters
    params = gp.kernel_parameters()
    k = ("squared_exp", params[0], params[1])
    K = O.make_covariance_matrix(k, X, X) + gp.noise ** 2 * np.eye(n)
    Ks = O.make_covariance_matrix(k, X, Xq)
    want = gp.prior.c + Ks.T @ np.linalg.solve(K, y - gp.prior.c)
    assert np.max(np.abs(mean - want)) < 1e-8 * np.max(np.abs(want))
    rec = {"config": {"n": n, "d": d, "m": m, "cfg": cfg, "kernel": "squared_exp", "max_iter": 100,
                      "convergence_fraction": 0.05},
           "iterations": gp.iterations, "params": params.tolist(), "noise": gp.noise, "prior": gp.prior.c,
           "likelihood": gp.likelihood(), "mean": mean.tolist(), "var": var.tolist()}
    with open(os.path.join(HERE, "golden_v2.json"), "w") as f:
        json.dump({"config0_default": rec}, f, indent=1)
    print("config0_default: iterations", gp.iterations, "params", params, "noise", gp.noise)


if __name__ == "__main__":
    main()
