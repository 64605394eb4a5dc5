"""The boundary claim of INTEGRATION.md section 2, executed: the reference's own five `main_*` drivers and its `main()`, verbatim,
compile and link against include/aligntools.h + libaligntools.so once the DP functions are cut out of `src/alignment.h`.

Runs in the build container only (it needs /root/reference; nothing of the reference is copied into the repository or travels to
the GPU box -- the patched header is assembled in a temporary directory from LINE RANGES of the reference's file, and main.c is
copied there only because `#include "alignment.h"` looks beside the including file first).  The drivers' usage and error paths
then run as they do in the stock binary; a real alignment ends in the shim's "no HIP device" die() here, which shows that the
drivers reach the GPU path and nothing else.
"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"
PKG = os.path.join(ROOT, "aligntools", "c_amd")

# lines of /root/reference/src/alignment.h that go (1-based, inclusive): everything the shim's header declares or replaces
REMOVED = [
    (37, 48),     # matrix_t                      -> opaque `struct at_matrix` (aligntools.h)
    (51, 79),     # junction_t, opt_t, die        -> aligntools.h (same layouts, same texts)
    (90, 170),    # max5, init_opt, create_matrix, destory_matrix
    (172, 275),   # strrev, str_toupper, kstring_destory, kstring_read, isvalueinarray
    (280, 315),   # min3, edit_dist
    (372, 473),   # trace_back_gla, align_gla
    (558, 694),   # trace_back_fit_affine_jump, align_fit_affine_jump
    (766, 847),   # trace_back_local_affine, align_local_affine
    (896, 964),   # trace_back_overlap, align_overlap
]
KEPT_DRIVERS = ["main_edit_dist", "main_global_affine", "main_fit_affine_jump", "main_local_affine", "main_overlap"]

pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "alignment.h")), reason="needs /root/reference (build container only)")


@pytest.fixture(scope="module")
def patched(tmp_path_factory):
    if not (os.path.isfile(os.path.join(PKG, "libaligntools.so")) and os.path.isfile(os.path.join(PKG, "libaligntools_hip.so"))):
        pytest.skip("libaligntools*.so not built (python -c 'import __graft_entry__ as g; g.build()')")
    d = str(tmp_path_factory.mktemp("boundary"))
    lines = open(os.path.join(REF, "alignment.h"), encoding="latin1").read().split("\n")
    gone = set()
    for lo, hi in REMOVED:
        gone.update(range(lo, hi + 1))
    out = []
    for k, text in enumerate(lines, 1):
        if k in gone:
            continue
        out.append(text)
        if k == 24:   # behind the reference's own `typedef enum { true, false } bool;`
            out.append('#include "aligntools.h"   /* the patch: kstring_read, init_opt, die, align_*, edit_dist, trace_back_* -> libaligntools.so -> GPU */')
    body = "\n".join(out)
    for name in KEPT_DRIVERS:
        assert name in body, name
    for name in ("max5", "create_matrix", "strrev"):
        assert ("%s(" % name) not in body and ("%s (" % name) not in body, name
    open(os.path.join(d, "alignment.h"), "w", encoding="latin1").write(body)
    shutil.copy(os.path.join(REF, "main.c"), os.path.join(d, "main.c"))
    exe = os.path.join(d, "alignTools")
    cmd = ["gcc", "-g", "-O2", os.path.join(d, "main.c"), os.path.join(REF, "kstring.c"), "-I", d, "-I", REF, "-I", os.path.join(ROOT, "include"),
           "-o", exe, "-L", PKG, "-laligntools", "-laligntools_hip", "-lz", "-Wl,-rpath," + PKG]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def _run(exe, *args):
    return subprocess.run([exe] + list(args), capture_output=True, text=True, timeout=120, env=dict(os.environ, AT_QUIET_FIT="0"))


def test_reference_drivers_link_against_the_boundary(patched):
    r = subprocess.run(["nm", "-u", patched], capture_output=True, text=True)
    undefined = r.stdout
    for sym in ("align_gla", "align_local_affine", "align_fit_affine_jump", "align_overlap", "edit_dist", "kstring_read", "init_opt"):
        assert sym in undefined, sym     # the drivers' calls resolve in libaligntools.so, not in the translation unit


def test_usage_and_error_paths_of_the_patched_binary(patched, tmp_path):
    r = _run(patched)
    assert r.returncode == 1 and "Usage:   alignTools <command> [options]" in r.stderr
    r = _run(patched, "foo")
    assert r.returncode == 1 and "[main] unrecognized command 'foo'" in r.stderr
    for cmd in ("global", "local", "fit", "overlap", "edit"):
        r = _run(patched, cmd)
        assert r.returncode == 1 and "Usage" in r.stderr, (cmd, r.stderr)
    r = _run(patched, "local", str(tmp_path / "nofile.fa"))
    assert r.returncode == 255 and "FATAL ERROR: Can't open" in r.stderr
    one = tmp_path / "one.fa"
    one.write_text(">a\nACGT\n")
    r = _run(patched, "global", str(one))
    assert r.returncode == 255 and "fail to read sequence" in r.stderr
    three = tmp_path / "three.fa"
    three.write_text(">a\nACGT\n>b\nACGA\n>c\nAC\n")
    r = _run(patched, "global", str(three))
    assert r.returncode == 255 and "more than 2 sequences" in r.stderr


def test_an_alignment_goes_to_the_gpu_path_or_dies(patched, tmp_path):
    """With a GPU the patched binary prints the reference's answer; without one it dies in the shim ("no CPU fallback")."""
    two = tmp_path / "two.fa"
    two.write_text(">a\nPLEASANTLY\n>b\nMEANLY\n")
    r = _run(patched, "local", "-m", "2", "-u", "-2", "-o", "-5", "-e", "-2", str(two))
    if r.returncode == 0:
        assert r.stdout == "score=4.000000\nLEA\nMEA\n"
    else:
        assert r.returncode == 255 and "FATAL ERROR" in r.stderr and "no CPU fallback" in r.stderr, r.stderr
