import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle_c():
    """The plain-C oracle (built on demand; checker only)."""
    from oracle import c_oracle
    c_oracle.build()
    c_oracle.lib()
    return c_oracle


@pytest.fixture
def debug_set():
    """pcreg_debug_set(key, value) for the duration of one test: selects the other side of a certified fast path (the library
    reads no environment variable); every key touched is reset to 0 afterwards."""
    from pcreg_amd._lib import check, lib
    touched = []

    def _set(key: str, value: int = 1):
        check(lib().pcreg_debug_set(key.encode(), int(value)))
        touched.append(key)
    yield _set
    for key in touched:
        lib().pcreg_debug_set(key.encode(), 0)


@pytest.fixture(scope="session")
def oracle_py():
    from oracle import pcreg_oracle
    return pcreg_oracle


def rigid_case(n, seed, noise=0.02, outlier_frac=0.3, box=(30.0, 20.0, 25.0)):
    """Synthetic correspondences: pts1 = surface, pts2 = model points, with outliers."""
    from oracle import pcreg_oracle as o
    rng = np.random.default_rng(seed)
    P = rng.uniform(-1, 1, (n, 3)) * np.array(box) + np.array([40.0, 25.0, 50.0])
    R = o.eul2rotm(rng.uniform(-np.pi, np.pi, 3))
    t = rng.uniform(-10, 10, 3)
    pts2 = P
    pts1 = P @ R + t + rng.normal(0, noise, P.shape)      # [pts2,1]*T = pts1 with T = [R 0; t 1]
    k = int(outlier_frac * n)
    if k:
        bad = rng.choice(n, k, replace=False)
        pts1[bad] = rng.uniform(-1, 1, (k, 3)) * np.array(box) * 1.5 + np.array([40.0, 25.0, 50.0])
    T = np.eye(4)
    T[:3, :3] = R
    T[3, :3] = t
    return pts1, pts2, T
