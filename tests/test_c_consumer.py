"""A plain C99 program (tests/c/abi_consumer.c) built against include/*.h with -pedantic -Werror and linked
against the two shared libraries: the boundary is usable from C exactly as INTEGRATION.md says.  The host-only
half runs here; the half that aligns runs on the GPU box."""
import os
import subprocess

import pytest

from conftest import ROOT

PKG = os.path.join(ROOT, "aligntools", "c_amd")


@pytest.fixture(scope="module")
def consumer(tmp_path_factory):
    import aligntools.c_amd.build as B
    B.build()
    exe = str(tmp_path_factory.mktemp("cabi") / "abi_consumer")
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O1", os.path.join(ROOT, "tests", "c", "abi_consumer.c"),
           "-I" + os.path.join(ROOT, "include"), "-L" + PKG, "-laligntools", "-laligntools_hip", "-Wl,-rpath," + PKG, "-o", exe]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    return exe


def test_c_consumer_host_only(consumer):
    p = subprocess.run([consumer, "nogpu"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "host-only ok" in p.stdout


@pytest.mark.gpu
def test_c_consumer_on_gpu(consumer):
    p = subprocess.run([consumer, "gpu"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "gpu ok" in p.stdout
