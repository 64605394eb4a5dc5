"""GPU suite: the routing users get.  tests/test_gpu_parity.py runs under AT_PACKED_MIN_ROUNDS=0 so that its small batches reach the
64-lane packed kernels; here nothing is set: small batches take whatever kernel align_device picks for them by default (mostly the
int32 kernel, the packed ones from one round of the resident waves up), and the results are the reference's all the same."""
import os
import random

import pytest

import oracle as O   # test infrastructure: the checker
from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def al():
    import aligntools.c_amd as A
    saved = {k: os.environ.pop(k) for k in ("AT_PACKED_MIN_ROUNDS", "AT_NO_PACKED", "AT_GROUP", "AT_TAIL_SPLIT") if k in os.environ}
    a = A.Aligner()
    yield a
    a.close()
    os.environ.update(saved)


def test_goldens_under_default_routing(al):
    from test_gpu_parity import _group, _run_group
    seen = set()
    for name in ("random_small.jsonl", "random_dna.jsonl", "known_answers.jsonl"):
        for key, cases in _group(load_golden(name)).items():
            _run_group(al, key, cases)
            seen.add(al.last_config.split(" store=")[0])
    assert any(c.startswith("int32") for c in seen) and any(c.startswith("packed16") for c in seen), seen


def test_batches_on_both_sides_of_the_packed_threshold(al):
    """Uniform global batches of 1024 x 1024 (64-lane packed kernel, two alignments per wave): 600 pairs are less than one round of the
    resident waves and stay on the int32 kernel, 7 000 take the packed one -- the same answers from both (sample against the oracle)."""
    rng = random.Random(77)
    dna = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    base = [(dna(1024), dna(1024)) for _ in range(40)]
    al.set_scoring(1, -1, -4, -1)
    small = al.align_batch("global", base * 15, render=False)
    assert "int32" in al.last_config, al.last_config
    big = al.align_batch("global", base * 175, render=False)
    assert "packed16 x4" in al.last_config and "1x64-lane" in al.last_config, al.last_config
    for k in range(40):
        r = O.align(O.GLOBAL, base[k][0], base[k][1], 1, -1, -4, -1)
        for res, q in ((small, k), (small, 40 * 14 + k), (big, k), (big, 40 * 174 + k)):
            assert (int(res["score"][q]), int(res["state"][q]), res["ops"][q]) == (r["score"], r["state"], r["ops"]), (k, q)
