"""CPU suite: the C-ABI library loads, exports every symbol include/aligntools_hip.h
declares, its host helpers (pack / render) work, and it FAILS LOUDLY without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import aligntools.c_amd as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from aligntools.c_amd import build
    build.build()
    return A.load_library()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "aligntools_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(at_[a-z_]+)\s*\(", hdr)))
    assert declared == sorted(A.ABI_SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None


def test_reference_named_surface_exported(lib):
    """libaligntools.so exports every function include/aligntools.h declares -- the five kernels, init_opt / die / kstring_*, the batch
    reader, and the traceback half: at_fill_matrix, the four trace_back_*() and destory_matrix (alignment.h:372, 558, 766, 896, 153)."""
    hdr = open(os.path.join(ROOT, "include", "aligntools.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b([a-z_]+[a-z0-9_]*)\s*\([^;{]*\)\s*;", hdr)))
    for need in ("align_gla", "align_local_affine", "align_fit_affine_jump", "align_overlap", "edit_dist", "trace_back_gla", "trace_back_local_affine",
                 "trace_back_fit_affine_jump", "trace_back_overlap", "destory_matrix", "at_fill_matrix", "init_opt", "die", "kstring_read", "kstring_destory"):
        assert need in declared, (need, declared)
    host = C.CDLL(os.path.join(ROOT, "aligntools", "c_amd", "libaligntools.so"))
    for name in declared:
        assert getattr(host, name) is not None, name


def _no_gpu():
    try:
        import torch
        return not torch.cuda.is_available()
    except Exception:
        return True


@pytest.mark.skipif(not _no_gpu(), reason="a GPU is visible")
def test_fails_loudly_without_gpu(lib):
    with pytest.raises(A.AlignToolsError) as ei:
        A.Aligner()
    assert ei.value.code == -2 and "no CPU fallback" in str(ei.value)
    with pytest.raises(A.AlignToolsError):
        A.align_local_affine("ACGT", "ACGT")


def test_pack_2bit_and_8bit(lib):
    words, w1, w2, l1, l2, bits = A.pack_pairs([(b"ACGTACGTACGTACGTA", b"TTTT"), (b"G", b"CA")])
    assert bits == 2
    # base k at bits [2k, 2k+1]: ACGT -> 0,1,2,3 -> 0b11100100 = 0xE4 per 4 bases
    assert words[w1[0]] == 0xE4E4E4E4 and words[w1[0] + 1] == 0
    assert words[w2[0]] == 0xFF
    assert words[w1[1]] == 2 and words[w2[1]] == 0b0001
    words, w1, w2, l1, l2, bits = A.pack_pairs([(b"PLEASANTLY", b"MEANLY")])
    assert bits == 8
    assert bytes(words[w1[0]:w1[0] + 3].tobytes()[:10]) == b"PLEASANTLY"
    assert bytes(words[w2[0]:w2[0] + 2].tobytes()[:6]) == b"MEANLY"


def test_render_matches_reference_strings(lib):
    # PLEASANTLY / MEANLY local: 3 diagonal steps ending at (4,3): LEA / MEA
    r1 = C.create_string_buffer(8)
    r2 = C.create_string_buffer(8)
    assert lib.at_render(bytes([0, 0, 0]), 3, b"PLEASANTLY", 4, b"MEANLY", 3, r1, r2) == 0
    assert (r1.value, r2.value) == (b"LEA", b"MEA")
    # ops inconsistent with the sequences are rejected, not read out of bounds
    assert lib.at_render(bytes([0, 0, 0]), 3, b"PL", 2, b"ME", 2, r1, r2) != 0


def test_render_roundtrip_against_oracle(lib):
    import oracle as O
    from conftest import load_golden
    al = object.__new__(A.Aligner)
    al._lib = lib
    for c in load_golden("random_small.jsonl")[:300]:
        if c["mode"] == "edit":
            continue
        r = O.align(O.MODE_NAMES[c["mode"]], c["s1"], c["s2"], c["m"], c["u"], c["o"], c["e"], c["j"], c["use_jump"], c["sites"])
        a, b = A.Aligner.render(al, r["ops"], c["s1"].encode("latin1"), r["end_i"], c["s2"].encode("latin1"), r["end_j"])
        assert (a, b) == (c["r1"], c["r2"])


@pytest.mark.gpu
def test_one_hip_runtime_whatever_the_import_order():
    """The shim first, torch second (the order that used to leave torch without a GPU), and the other way round: both see
    the device and one alignment comes out right (aligntools.c_amd._one_hip_runtime)."""
    import subprocess
    import sys
    code = r'''
import sys
sys.path.insert(0, %r)
first = sys.argv[1]
if first == "shim":
    import aligntools.c_amd as A
    al = A.Aligner()
    import torch
else:
    import torch
    torch.cuda.is_available()
    import aligntools.c_amd as A
    al = A.Aligner()
assert torch.cuda.is_available(), "torch lost the GPU"
x = torch.arange(8, device="cuda").sum().item()
al.set_scoring(2, -2, -5, -2, -10, False, [])
r = al.align_batch("local", [("PLEASANTLY", "MEANLY")])
assert x == 28 and int(r["score"][0]) == 4 and (r["r1"][0], r["r2"][0]) == ("LEA", "MEA")
maps = open("/proc/self/maps").read()
libs = sorted(set(l.split()[-1] for l in maps.splitlines() if "libamdhip64" in l))
assert len(libs) == 1, libs
print("ok", first, libs[0])
''' % ROOT
    for first in ("shim", "torch"):
        p = subprocess.run([sys.executable, "-c", code, first], capture_output=True, text=True, timeout=600)
        assert p.returncode == 0 and "ok " + first in p.stdout, p.stdout[-1500:] + p.stderr[-1500:]
