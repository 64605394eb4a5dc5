import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: minutes on the GPU (the BASELINE configs at their stated sizes); part of -m gpu")


def load_golden(name):
    import json
    path = os.path.join(ROOT, "tests", "golden", name)
    with open(path) as fh:
        return [json.loads(line) for line in fh]


@pytest.fixture(scope="session")
def golden():
    return load_golden
