"""A short randomised campaign inside the GPU suite (tests/fuzz_parity.py; AT_FUZZ_CASES enlarges it)."""
import os

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12])
def test_fuzz_campaign(seed):
    import fuzz_parity
    os.environ["AT_PACKED_MIN_ROUNDS"] = "0"
    try:
        n = fuzz_parity.run(int(os.environ.get("AT_FUZZ_CASES", "1500")), seed, verbose=False)
    finally:
        del os.environ["AT_PACKED_MIN_ROUNDS"]
    assert n >= 1500
