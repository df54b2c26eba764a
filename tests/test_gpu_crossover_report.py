"""Where does the device path start to pay?  (readme.md:7 of the reference: Bayesian-optimisation-sized problems; SURVEY
section 7 "hard parts": small problems.)  The reference's CPU path -- here its restatement, the oracle, on ONE host thread as
nalgebra runs -- against the device path through the C ABI, host pointers in and out (what the Rust shim would pass), for
N = 4 ... 2048: fit, predict of one point (mean + variance), predict of 64 points.  The table goes to
gpurun_out/crossover.json (copied to profiles/ and INTEGRATION.md section 4 by the builder); asserted are only the two ends:
at N = 4 the host wins (a kernel launch costs more than the whole problem), at N = 2048 the device wins every column."""
import json
import os
import time

import numpy as np
import pytest

from conftest import ROOT, rand_inputs
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _best(fn, reps):
    fn()
    b = 1e30
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        b = min(b, time.perf_counter() - t0)
    return 1e6 * b  # microseconds


def test_host_device_crossover_table(ctx):
    d = 4
    k = ("squared_exp", 0.9, 1.1)
    noise = 0.1
    rows = []
    O.set_threads(1)
    try:
        for n in (4, 16, 64, 128, 256, 512, 1024, 2048):
            X = rand_inputs(n, d, 10 + n)
            y = np.sin(X.sum(axis=1))
            Xq = rand_inputs(64, d, 11 + n)
            reps = 20 if n <= 256 else (5 if n <= 1024 else 2)
            gp = O.OracleGP(O.ZeroPrior(), k, noise, None, X, y)
            host_fit = _best(lambda: O.make_cholesky_cov_matrix(k, X, noise), reps)
            host_p1 = _best(lambda: (gp.predict(Xq[:1]), gp.predict_variance(Xq[:1])), reps)
            host_p64 = _best(lambda: (gp.predict(Xq), gp.predict_variance(Xq)), reps)
            chol = ctx.cholesky_from_inputs(k, X, noise)
            dev_fit = _best(lambda: chol.refactor(k, noise), reps)
            dev_p1 = _best(lambda: (chol.predict_mean(k, y, Xq[:1]), chol.predict_variance(k, Xq[:1])), reps)
            dev_p64 = _best(lambda: (chol.predict_mean(k, y, Xq), chol.predict_variance(k, Xq)), reps)
            chol.set_targets(y)
            dev_p1c = _best(lambda: (chol.predict_mean(k, None, Xq[:1]), chol.predict_variance(k, Xq[:1])), reps)
            chol.free()
            rows.append({"n": n, "host_fit_us": host_fit, "device_fit_us": dev_fit, "host_predict1_us": host_p1,
                         "device_predict1_us": dev_p1, "device_predict1_cached_alpha_us": dev_p1c, "host_predict64_us": host_p64,
                         "device_predict64_us": dev_p64})
            print({k_: (round(v, 1) if isinstance(v, float) else v) for k_, v in rows[-1].items()})
    finally:
        O.set_threads(0)
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "crossover.json"), "w") as f:
        json.dump({"what": "host (oracle restatement of the nalgebra path, 1 thread) vs device (C ABI, host pointers), microseconds, best of a "
                           "few repetitions; d = 4, RBF; predict = mean + variance", "rows": rows}, f, indent=1)
    assert rows[0]["host_fit_us"] < rows[0]["device_fit_us"] and rows[0]["host_predict1_us"] < rows[0]["device_predict1_us"]
    last = rows[-1]
    assert last["device_fit_us"] < last["host_fit_us"] and last["device_predict1_us"] < last["host_predict1_us"]
    assert last["device_predict64_us"] < last["host_predict64_us"]
