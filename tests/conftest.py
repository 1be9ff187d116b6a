import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def spec():
    from oracle import oracle as O
    return O.load_spec()


@pytest.fixture(scope="session")
def c4fm_1s():
    """1 s of seeded C4FM at 240 ksps, 30 dB SNR (BASELINE.json config 1 shape)."""
    from p25rx_amd import c4fm
    return c4fm.synth(1.0, seed=1, snr_db=30.0)
