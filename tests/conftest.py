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


@pytest.fixture(scope="session")
def fake_rccl(tmp_path_factory):
    """Path of a stand-in for librccl.so whose collectives go through files in the job's rendezvous directory
    (tests/c/fake_rccl_files.c): AT_RCCL_LIB=<this> lets several ranks share the test box's one card."""
    import subprocess
    d = tmp_path_factory.mktemp("fake_rccl")
    so = str(d / "libfake_rccl.so")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    subprocess.run(["gcc", "-shared", "-fPIC", "-O1", "-std=c99", "-I" + os.path.join(rocm, "include"),
                    os.path.join(ROOT, "tests", "c", "fake_rccl_files.c"), "-o", so, "-L" + os.path.join(rocm, "lib"), "-lamdhip64",
                    "-Wl,-rpath," + os.path.join(rocm, "lib")], check=True)
    return so
