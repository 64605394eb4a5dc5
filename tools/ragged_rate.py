"""Host-entry rate on a RAGGED batch (every pair its own lengths -> int32 kernel, largest pairs first), against the same
number of cells in uniform 150 x 150 pairs (packed kernel).  100k pairs, local, tracebacks."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aligntools.c_amd as A
from aligntools.c_amd.synth import synth_pairs_blob

n = 100000
rng = np.random.default_rng(5)
al = A.Aligner(0)
al.set_scoring(2, -2, -5, -2)
lib = A.load_library()
p = lambda a: a.ctypes.data_as(C.c_void_p)
for name, lo in (("uniform 150x150", 150), ("ragged 100..150 x 100..150", 100), ("ragged 30..150 x 30..150", 30)):
    len1 = rng.integers(lo, 151, n).astype(np.int32)
    len2 = rng.integers(lo, 151, n).astype(np.int32)
    blob = synth_pairs_blob(0x5EED0002, n, 150, 150).reshape(-1).copy()
    off1 = np.arange(n, dtype=np.int64) * 300
    off2 = off1 + 150
    score, ei, ej, st, nops = (np.zeros(n, np.int32) for _ in range(5))
    ops = np.zeros(n * 300 + 64, np.uint8)
    ts = []
    for it in range(5):
        t0 = time.perf_counter()
        rc = lib.at_align_batch(al._h, A.MODE_LOCAL, n, p(blob), p(off1), p(len1), p(off2), p(len2), 1, p(score), p(ei), p(ej), p(st),
                                p(ops), p(off1), p(nops))
        ts.append(time.perf_counter() - t0)
        assert rc == 0
    t = min(ts[1:])
    cells = float((len1.astype(np.int64) * len2).sum())
    print("%-28s %.2f ms = %.0f GCUPS host path (%s)" % (name, t * 1e3, cells / t / 1e9, al.last_config[:60]))
