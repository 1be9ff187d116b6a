"""A few fixed seeds of tests/fuzz_parity.py under pytest (the long runs are `python tests/fuzz_parity.py --minutes M`;
docs/MEASUREMENTS.md records them)."""
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


@pytest.mark.parametrize("seed", [10573, 13804])
def test_scenes_the_long_runs_found(seed):
    """10573: tracking clock + time shards, the first detection of a shard governs past the end of its tile (the resolved
    dibit offsets were one too many: first_seg_end stopped at the tile's end).  13804: back-to-back sync words -- a streaming
    chunk holds more than n / 10 + 2 dibits (the host mirrors now size for the hard ceiling n / 6 + 2)."""
    import torch
    from oracle import oracle as O
    from p25rx_amd.frontend import FrontEnd as FE
    import fuzz_parity
    assert fuzz_parity.run_scene(seed, O, FE, torch) > 10
