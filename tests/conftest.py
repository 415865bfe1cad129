import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def have_gpu() -> bool:
    import torch
    return torch.cuda.is_available()


def random_segments(p, B, K, seed):
    """SURVEY.md §8d kernel micro-benchmark law: random-but-physical nodes."""
    rng = np.random.default_rng(seed)
    x = np.zeros((B, K + 1, 14))
    x[..., 0] = rng.uniform(0.999, 1.0, (B, K + 1))
    x[..., 1:4] = rng.uniform(0, 1, (B, K + 1, 3))
    v = rng.uniform(-0.2, 0.2, (B, K + 1, 3))
    v[np.linalg.norm(v, axis=-1) < 0.01] = [0.05, -0.05, 0.02]
    x[..., 4:7] = v
    q = rng.normal(size=(B, K + 1, 4))
    x[..., 7:11] = q / np.linalg.norm(q, axis=-1, keepdims=True)
    x[..., 11:14] = rng.uniform(-0.1, 0.1, (B, K + 1, 3))
    mag = rng.uniform(p.Tmin, p.Tmax, (B, K + 1))
    d = np.zeros((B, K + 1, 3))
    d[..., 0] = 1.0
    d[..., 1:] = np.tan(np.radians(20.0)) * rng.uniform(-0.7, 0.7, (B, K + 1, 2))
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    u = mag[..., None] * d
    sigma = rng.uniform(0.5, 2.0, B)
    return x, u, sigma


@pytest.fixture(scope="session")
def aero_tables():
    z = np.load(os.path.join(GOLDEN, "lift_drag_tables.npz"))
    return z["drag"], z["lift"], z["torque"]
