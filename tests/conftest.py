import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh clone has no built artefacts (they are git-ignored): build the HIP library (hipcc cross-compiles without a GPU)
    and the oracle once, exactly as __graft_entry__.build() does.  Nothing is built when both are already there."""
    lib = os.path.join(ROOT, "motionplanning_5d_m_amd", "libcfs_hip.so")
    orc = os.path.join(ROOT, "oracle", "libcfs_oracle.so")
    if not (os.path.exists(lib) and os.path.exists(orc)):
        import __graft_entry__ as g
        g.build()


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "cfs_cases.npz"))


@pytest.fixture(scope="session")
def route_wp():
    return np.load(os.path.join(ROOT, "tests", "golden", "route_wp_200i_xori.npy"))


@pytest.fixture(scope="session")
def O():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def pkg():
    import motionplanning_5d_m_amd as m
    return m


@pytest.fixture(scope="session")
def gpu(pkg):
    """The HIP path must be the one that runs: no device -> fail loudly (never skip to a fallback)."""
    assert pkg.device_count() >= 1, "gpu-marked tests need a HIP device; libcfs_hip.so has no CPU fallback"
    return pkg
