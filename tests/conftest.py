import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import ndt_oracle
    ndt_oracle.build()
    return ndt_oracle


@pytest.fixture(scope="session")
def c1_world():
    """BASELINE.json configs[0]: 360-pt scan vs 5k-pt map, launch-file parameters."""
    from ndt_slam_amd import synth
    cfg = synth.CONFIGS["C1"]
    m = synth.make_map(cfg["n_map"], cfg["half"])
    sf = synth.ScanFactory(m, cfg["half"], cfg["n_scan"])
    return m, sf, cfg
