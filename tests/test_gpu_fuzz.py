"""A few fixed seeds of tests/fuzz_parity.py under pytest (the long runs are `python tests/fuzz_parity.py --minutes M`;
docs/MEASUREMENTS.md records them and what they found; the findings have tests of their own in test_gpu_recv_general.py /
test_gpu_host.py)."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("first", [1, 1001, 2001, 3001])
def test_random_scenes_bit_exact(first):
    import torch
    from oracle import oracle as O
    from p25rx_amd.frontend import FrontEnd as FE
    import fuzz_parity
    for seed in range(first, first + 12):
        assert fuzz_parity.run_scene(seed, O, FE, torch) > 10


@pytest.mark.parametrize("first", [5, 4005])
def test_random_aux_scenes(first):
    """ranges inside a buffer (offset / history / absolute grid), K0 on 1 - 3 channels, K6, NID with random bit errors"""
    import torch
    from oracle import oracle as O
    from p25rx_amd.frontend import FrontEnd as FE
    import fuzz_parity
    for seed in range(first, first + 10):
        assert fuzz_parity.run_aux_scene(seed, O, FE, torch) >= 3
