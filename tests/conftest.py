import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# Handles created by the tests with non-default tables run the generic kernels unless a test ASKS for the specialised ones
# (specialize = REQUIRE / FORCE: tests/test_gpu_spec.py): the parity and fuzz suites create hundreds of handles with random
# tables, and a hipRTC build per table (seconds each) would triple the suite's time.  Read once by the library, at the first
# p25fe_create of the process.
os.environ.setdefault("P25FE_JIT", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "pin: compares with dumps of the real kchmck/p25rx under tests/golden/pin/ (tools/pin/README.md); "
                                       "skipped while that directory is empty")


@pytest.fixture(scope="session")
def spec():
    from oracle import oracle as O
    return O.load_spec()


@pytest.fixture(scope="session")
def c4fm_1s():
    """1 s of seeded C4FM at 240 ksps, 30 dB SNR (BASELINE.json config 1 shape)."""
    from p25rx_amd import c4fm
    return c4fm.synth(1.0, seed=1, snr_db=30.0)
