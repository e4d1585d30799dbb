import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def cfg():
    from uav_airvision_amd.config import ConfigEuRoC
    return ConfigEuRoC()


@pytest.fixture(scope='session')
def stream0(cfg):
    """A short seeded synthetic stereo+IMU stream shared by the tests."""
    from uav_airvision_amd.synth import SyntheticStream
    return SyntheticStream(cfg, seed=0, n_frames=6)


@pytest.fixture(scope='session')
def frames0(stream0):
    return [stream0.frame(k) for k in range(stream0.n_frames)]
