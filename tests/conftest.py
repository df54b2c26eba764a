import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with `pytest -m gpu`")


@pytest.fixture(scope="session")
def ctx():
    """The product context.  On a GPU box a missing library / device is a hard failure, never a skip."""
    from friedrich_amd.device import Context

    c = Context()
    yield c
    c.close()


# kernel specs exercised everywhere (one per built-in kernel + both combinators)
ALL_KERNELS = [
    ("linear", 0.7),
    ("polynomial", 0.9, 0.5, 2.0),
    ("squared_exp", 0.8, 1.3),
    ("exponential", 0.9, 0.7),
    ("matern1", 1.1, 0.9),
    ("matern2", 0.7, 1.2),
    ("hyper_tan", 0.3, 0.1),
    ("multiquadric", 0.6),
    ("rational_quadratic", 1.5, 0.8),
    ("sum", ("squared_exp", 0.8, 1.3), ("linear", 0.2)),
    ("prod", ("matern2", 0.9, 1.1), ("squared_exp", 2.0, 0.5)),
    ("sum", ("prod", ("squared_exp", 1.0, 1.0), ("matern1", 2.0, 0.5)), ("exponential", 1.5, 0.2)),
]
# positive-definite-by-construction kernels for factorisation tests
PD_KERNELS = [
    ("squared_exp", 0.8, 1.3),
    ("matern2", 0.7, 1.2),
    ("matern1", 1.1, 0.9),
    ("exponential", 0.9, 0.7),
    ("sum", ("squared_exp", 0.8, 1.3), ("matern2", 0.5, 0.4)),
    ("prod", ("matern2", 0.9, 1.1), ("squared_exp", 2.0, 0.5)),
]


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = np.max(np.abs(b)) if b.size else 1.0
    if scale == 0.0:
        scale = 1.0
    return float(np.max(np.abs(a - b)) / scale) if b.size else 0.0


def rand_inputs(n, d, seed):
    rng = np.random.default_rng(seed)
    return np.asfortranarray(rng.random((n, d)))
