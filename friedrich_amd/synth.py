"""Deterministic synthetic GP-regression data: SplitMix64 -> U[0,1), reproducible
bit-for-bit in C++/Python/Rust.  X is n x d column-major; y = sin(sum_c X[i,c]) + 0.05*sqrt(12)*(u - 1/2)."""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64_uniform(seed, start, count):
    """u_k = (mix(seed + (k+1)*GOLDEN) >> 11) * 2^-53 for k in [start, start+count)"""
    with np.errstate(over="ignore"):
        k = np.arange(start, start + count, dtype=np.uint64) + np.uint64(1)
        z = np.uint64(seed) + k * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def make_inputs(n, d, seed):
    """X[i, c] = u(seed, i*d + c), returned column-major (Fortran order)"""
    u = splitmix64_uniform(seed, 0, n * d).reshape(n, d)
    return np.asfortranarray(u)


def make_outputs(X, seed):
    n = X.shape[0]
    noise_u = splitmix64_uniform(seed ^ 0x5EEDFACE, 0, n)
    return np.sin(X.sum(axis=1)) + 0.05 * np.sqrt(12.0) * (noise_u - 0.5)


def make_problem(n, d, cfg=0, m=0):
    """(X, y, Xq) for benchmark config `cfg` (seed 0x5EED0000 + cfg; queries from seed + 1)"""
    seed = 0x5EED0000 + cfg
    X = make_inputs(n, d, seed)
    y = make_outputs(X, seed)
    Xq = make_inputs(m, d, seed + 1) if m > 0 else np.zeros((0, d), order="F")
    return X, y, Xq


def default_hyperparameters(X, y, ls=None):
    """friedrich's builder defaults (builder.rs:73; kernel.rs:594-600): ls = mean pairwise distance (pass it in
    when it was computed elsewhere), ampl = var(y), noise = 0.1*std(y), constant prior = mean(y)."""
    var = float(np.mean(y * y) - np.mean(y) ** 2)
    return {"ls": ls, "ampl": var, "noise": 0.1 * np.sqrt(var), "prior": float(np.mean(y))}
