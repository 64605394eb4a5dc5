import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# torch bundles its own HIP runtime (libamdhip64.so.7 of ROCm 7.0 under torch/lib); libaligntools_hip.so is linked against
# the system one (ROCm 7.2, same soname).  Whichever is loaded first serves both: with torch first everything works, with
# the system runtime first torch reports "No HIP GPUs are available".  The tests that hand torch tensors to the C ABI
# therefore need torch's runtime in the process before the first at_init -- in whatever order the tests are selected.
try:
    import torch
    torch.cuda.is_available()
except Exception:   # no torch: only the tests that need it will fail
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    import json
    path = os.path.join(ROOT, "tests", "golden", name)
    with open(path) as fh:
        return [json.loads(line) for line in fh]


@pytest.fixture(scope="session")
def golden():
    return load_golden
