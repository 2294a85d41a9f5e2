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


# ---- workloads and oracle answers shared by the GPU test modules (generated / solved once per session) ---------------------
class _OracleCache:
    """want / chaotic / moved_by of a workload per solver, computed on first use"""

    def __init__(self, O, s, bt):
        self.O, self.s, self.bt, self._c = O, s, bt, {}

    def __call__(self, mode):
        if mode not in self._c:
            from helpers import chaotic_problems
            O, s, bt = self.O, self.s, self.bt
            margin = bt.margin_cfs if mode == "CFS" else bt.margin_psg
            nz = bt.noise if (mode == "PSGCFS" and bt.noise is not None) else None
            want = O.optimizer_batch(O.robotproperty2("M200i"), mode, s.H, 5, bt.x_init, bt.xR1, s.QQ, bt.ff, bt.caug, s.Aaug, s.Baug,
                                     s.lim, s.MAX_input, bt.obs, margin, s.epsilon_O, s.MAX_O_ITER, s.alpha, noise=nz, nthreads=0)
            chaotic, moved_by = chaotic_problems(O, s, bt, mode, want)
            self._c[mode] = (want, chaotic, moved_by)
        return self._c[mode]


@pytest.fixture(scope="session")
def c3(gpu):
    """BASELINE config 3 (batch 1024), obstacle rejection through the GPU distance entry point"""
    from motionplanning_5d_m_amd import workloads
    return workloads.config3(lambda rb, th, ob: gpu.dist_arm(rb, th, ob)[0], B=1024)


@pytest.fixture(scope="session")
def c3_oracle(O, c3):
    return _OracleCache(O, *c3)


@pytest.fixture(scope="session")
def c4(route_wp):
    """the first 512 routes of BASELINE config 4's shape (H = 40, two obstacles, RRTstar_CFS cost matrices)"""
    from motionplanning_5d_m_amd import workloads
    return workloads.config4(route_wp, B=512)


@pytest.fixture(scope="session")
def c4_oracle(O, c4):
    return _OracleCache(O, *c4)
