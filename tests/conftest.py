import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def coracle():
    from oracle.clib import COracle
    return COracle()


@pytest.fixture(scope="session")
def reflib():
    from oracle import clib
    if not clib.ref_available():
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")
    return clib.RefLib()


@pytest.fixture(scope="session")
def refwrap():
    from oracle import clib
    if not clib.ref_available():
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")
    return clib.RefWrapLib()


@pytest.fixture(scope="session")
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda", 0)


GOLDEN = os.path.join(ROOT, "tests", "golden")


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def small_cloud(seed, n, scale=(2.0, 1.5, 0.7)):
    rng = np.random.default_rng(seed)
    return (rng.random((n, 3)) * np.asarray(scale) - 0.3).astype(np.float32)


def surface_cloud(seed, n_raw=60000, dl=0.03):
    """A small room fragment subsampled by the C oracle (surfaces give realistic neighbour counts)."""
    from d3feat_amd.utils.synthetic import room_fragment
    from oracle.clib import COracle
    raw = room_fragment(seed, n_raw=n_raw, edge=1.0)
    return COracle().grid_subsampling(raw, dl)
