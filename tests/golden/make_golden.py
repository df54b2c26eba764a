#!/usr/bin/env python3
"""Generate tests/golden/golden_v1.npz.

The reference (friedrich 0.5.1) ships no golden vectors and cannot be built here (no Rust toolchain), so these
fixtures are produced by the CPU oracle (oracle/friedrich_oracle.c, the statement-by-statement restatement of the
reference) and every array is cross-checked against an independent implementation (scipy/LAPACK, dense numpy algebra)
before it is written.  They freeze the oracle: a later change to it that moves any value fails tests/test_golden.py.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np
import scipy.linalg as sl

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

CASES = [
    # name, kernel spec, n, d, m, noise, eps, n_add
    ("readme_1d", ("squared_exp", 1.0, 1.0), None, 1, None, 0.1, None, 4),
    ("se_d2", ("squared_exp", 0.6, 1.1), 64, 2, 9, 0.1, None, 8),
    ("matern2_d8", ("matern2", 1.2, 0.8), 96, 8, 16, 0.05, None, 32),
    ("sum_d3", ("sum", ("squared_exp", 0.8, 1.3), ("matern1", 0.5, 0.4)), 70, 3, 5, 0.2, None, 1),
    ("tanh_subst", ("hyper_tan", 1.0, 0.0), 48, 2, 0, 0.0, 1e-6, 0),
]


def inputs(name, n, d, m, seed):
    if name == "readme_1d":  # src/main.rs:16-17, 31-32
        X = np.array([[0.8], [1.2], [3.8], [4.2]])
        y = np.array([3.0, 4.0, -2.0, -2.0])
        Xq = np.array([[1.0], [2.0], [3.0]])
        Xadd = np.array([[0.0], [1.0], [2.0], [5.0]])
        yadd = np.array([2.0, 3.0, -1.0, -2.0])
        return X, y, Xq, Xadd, yadd
    rng = np.random.default_rng(seed)
    scale = 3.0 if name == "tanh_subst" else 1.0
    X = rng.random((n, d)) * scale
    y = np.sin(X.sum(axis=1)) + 0.1 * rng.standard_normal(n)
    Xq = rng.random((m, d)) * scale
    return X, y, Xq, None, None


def main():
    out, manifest = {}, {}
    for i, (name, kernel, n, d, m, noise, eps, n_add) in enumerate(CASES):
        X, y, Xq, Xadd, yadd = inputs(name, n, d, m, 100 + i)
        n = X.shape[0]
        prior = O.ConstantPrior(float(np.mean(y)))
        gp = O.OracleGP(prior, kernel, noise, eps, X, y)
        L = np.tril(gp.L)
        K = O.make_covariance_matrix(kernel, X, X) + noise * noise * np.eye(n)
        rec = {"X": X, "y": y, "L": L, "subst": gp.subst.astype(np.int64), "y_res": gp.y}
        if eps is None:
            assert np.max(np.abs(L - sl.cholesky(K, lower=True))) < 1e-12 * np.max(np.abs(L)), name
        else:
            assert len(gp.subst) > 0, name
        if Xq.shape[0] > 0:
            mean, var = gp.predict_mean_variance(Xq)
            Kinv = np.linalg.inv(K)
            Ks = O.make_covariance_matrix(kernel, X, Xq)
            assert np.max(np.abs(mean - (prior.c + Ks.T @ Kinv @ gp.y))) < 1e-9, name
            rec.update({"Xq": Xq, "mean": gp.predict(Xq), "var": gp.predict_variance(Xq), "mean2": mean, "var2": var,
                        "cov": gp.predict_covariance(Xq), "likelihood": np.array([gp.likelihood()])})
            pm, pc, pl = gp.sample_at(Xq)
            rec.update({"post_mean": pm, "post_cov": pc, "post_l": pl})
        if n_add:
            if Xadd is None:
                rng = np.random.default_rng(500 + i)
                Xadd = rng.random((n_add, X.shape[1]))
                yadd = np.sin(Xadd.sum(axis=1))
            gp.add_samples(Xadd, yadd)
            st, Lfull, _ = O.make_cholesky_cov_matrix(kernel, gp.X, noise)
            assert np.max(np.abs(np.tril(gp.L) - np.tril(Lfull))) < 1e-12 * np.max(np.abs(np.tril(Lfull))), name
            rec.update({"Xadd": Xadd, "L_added": np.tril(gp.L)})
        for key, val in rec.items():
            out[f"{name}/{key}"] = np.asarray(val)
        manifest[name] = {"kernel": kernel, "noise": noise, "eps": eps, "prior": prior.c, "keys": sorted(rec)}
    # GaussianProcess::default on the README set (heuristics + scaled ADAM), src/main.rs:18
    gp = O.OracleGP.default([[0.8], [1.2], [3.8], [4.2]], [3.0, 4.0, -2.0, -2.0])
    manifest["readme_default"] = {"iterations": gp.iterations, "params": gp.kernel_parameters().tolist(), "noise": gp.noise,
                                  "prior": gp.prior.c, "predict_1": float(gp.predict([[1.0]])[0]),
                                  "variance_1": float(gp.predict_variance([[1.0]])[0]), "likelihood": gp.likelihood()}
    np.savez_compressed(os.path.join(HERE, "golden_v1.npz"), **out)
    with open(os.path.join(HERE, "golden_v1.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    print("wrote", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "golden_v1.npz")), "bytes")


if __name__ == "__main__":
    main()
