"""Build the gfx950 shim (libaligntools_hip.so) and the C host (alignTools CLI) in-tree."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libaligntools_hip.so")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources if os.path.exists(s))


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-j8", "-C", HERE] + (["-B"] if force else []), capture_output=True, text=True)
    if verbose or r.returncode:
        print(r.stdout[-4000:])
        print(r.stderr[-4000:])
    if r.returncode:
        raise RuntimeError("aligntools.c_amd: build failed (make -C %s)" % HERE)
    return LIB
