"""A short randomised campaign inside the GPU suite (tests/fuzz_parity.py; AT_FUZZ_CASES enlarges it).  Four seeds; 21 and 24 were picked
(tools/gpu/r03_g.sh: a scan over seeds with the kernel class of every batch printed) because their draw reaches the kernel families that
seeds 11 and 12 miss: the 4-lane groups, 16 / 19 rows per lane on the 16-lane groups, the 32-lane groups, the packed overlap kernel with
4 and 16 rows per lane, ragged frames and the ragged packed overlap."""
import os

import pytest

pytestmark = pytest.mark.gpu

SEEN = {}


@pytest.mark.parametrize("seed", [11, 12, 21, 24])
def test_fuzz_campaign(seed):
    import fuzz_parity
    os.environ["AT_PACKED_MIN_ROUNDS"] = "0"
    try:
        n = fuzz_parity.run(int(os.environ.get("AT_FUZZ_CASES", "1500")), seed, verbose=False, classes=SEEN)
    finally:
        del os.environ["AT_PACKED_MIN_ROUNDS"]
    assert n >= 1500


@pytest.mark.parametrize("aim", ["overlap and edit, scores only (the sweeps on the gap ramp)", "edit -u 1 (every form of the bit-parallel kernel)"])
def test_fuzz_campaign_aimed(aim):
    """Round 3's new sweeps get a campaign of their own: overlap without tracebacks and cell-by-cell edit distance (at_sweep.hip.h, RAMP),
    and the bit-parallel kernel with one alignment per lane in all four widths (AT_MYERS_LANE_MIN_PAIRS = 1: small batches too)."""
    import fuzz_parity
    env = ({"AT_FUZZ_MODES": "overlap,edit", "AT_FUZZ_TB": "0"} if aim.startswith("overlap")
           else {"AT_FUZZ_MODES": "edit", "AT_FUZZ_EDIT_UNIT": "1", "AT_MYERS_LANE_MIN_PAIRS": "1"})
    os.environ.update(env)
    seen = {}
    try:
        n = fuzz_parity.run(int(os.environ.get("AT_FUZZ_CASES", "1500")), 31 if aim.startswith("overlap") else 32, verbose=False, classes=seen)
    finally:
        for k in env:
            del os.environ[k]
    assert n >= 1500
    keys = " | ".join(seen)
    for family in (("overlap scores-only int32", "edit int32") if aim.startswith("overlap") else ("myers W2 64x1", "myers W16 64x1", "myers W32 64x1")):
        assert family in keys, (family, sorted(seen))
    if not aim.startswith("overlap"):
        assert sum(1 for k in seen if "myers" in k and "64x1" in k) >= 5, sorted(seen)   # (2, 3, 4, 5, 8, 16, 32 words per lane: most of them)


def test_fuzz_campaign_walk_kernel():
    """Round 4's walk kernel (at_walk16.hip.h) gets a campaign drawn inside its shape classes (tests/fuzz_walk_kernel.py): both group
    widths, teams of lanes on and off, batches in pieces, byte alphabets -- whole batches against the one-pass kernels, samples against
    the oracle."""
    import fuzz_walk_kernel
    saved = {k: os.environ.get(k) for k in ("AT_PACKED_MIN_ROUNDS", "AT_HOST_CHUNKS")}
    try:
        n = fuzz_walk_kernel.run(int(os.environ.get("AT_FUZZ_WALK_BATCHES", "120")), 41, verbose=False)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert n >= 120


def test_fuzz_campaign_reached_every_kernel_family():
    """(runs behind the four campaigns of this module) what their batches ran on, by at_last_config"""
    if len(SEEN) == 0:
        pytest.skip("the campaigns did not run in this session")
    keys = " | ".join(SEEN)
    for family in ("16x4 ", "8x8 ", "4x16 K16", "4x16 K19", "2x32 ", "1x64 ", "overlap packed16x4 1x64 K4", "overlap packed16x4 1x64 K16", " ragged",
                   "overlap packed16x4 1x64 K4 ragged", "int32", "myers", "fitj packed16x16"):
        assert family in keys, (family, sorted(SEEN))
