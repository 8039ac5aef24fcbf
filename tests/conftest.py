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


@pytest.fixture(scope="session")
def gist_base():
    return np.fromfile(os.path.join(GOLDEN, "gist_1000.bin"), dtype=np.float32).reshape(1000, 960)


@pytest.fixture(scope="session")
def gist_test():
    return np.fromfile(os.path.join(GOLDEN, "gist_test.bin"), dtype=np.float32).reshape(1000, 960)


def gist_like(n, dim=960, seed=1806):
    """Synthetic gist-shaped rows: |N(mu_j, sigma_j)| clipped to [0, 0.8], 4 decimals."""
    stats = np.load(os.path.join(GOLDEN, "gist_dim_stats.npy"))  # [2][960]: mean, std
    rng = np.random.Generator(np.random.PCG64(seed))
    mu = np.resize(stats[0], dim).astype(np.float32)
    sd = np.resize(stats[1], dim).astype(np.float32)
    x = rng.standard_normal((n, dim), dtype=np.float32) * sd + mu
    np.abs(x, out=x)
    np.clip(x, 0, 0.8, out=x)
    return np.round(x, 4).astype(np.float32)


def gist_clustered(n, dim=960, seed=1806, clusters=1024, spread=0.15):
    """gist-like rows in `clusters` Gaussian clusters (centres = gist_like(seed 4711), noise spread x sigma_j inside): margins between
    neighbours far below the resolution of an 8-bit key -- bench.py's `--data clustered` in numpy"""
    stats = np.load(os.path.join(GOLDEN, "gist_dim_stats.npy"))
    sd = np.resize(stats[1], dim).astype(np.float32)
    centres = gist_like(clusters, dim=dim, seed=4711)
    rng = np.random.Generator(np.random.PCG64(seed))
    c = rng.integers(0, clusters, size=n)
    x = rng.standard_normal((n, dim), dtype=np.float32) * (sd * np.float32(spread)) + centres[c]
    np.abs(x, out=x)
    np.clip(x, 0, 0.8, out=x)
    return np.round(x, 4).astype(np.float32)
