"""The host-buffer entry at_align_batch with the caller's buffers in pinned (page-locked) host memory against pageable ones
(C2 workload): what a caller gains by allocating its sequence blob and result arrays with hipHostMalloc / torch pin_memory."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aligntools.c_amd as A
from aligntools.c_amd.synth import synth_pairs_blob

n, l1, l2 = 100000, 150, 150
base = synth_pairs_blob(0x5EED0002, n, l1, l2).reshape(-1).copy()


def arr(x, pinned):
    if not pinned:
        return x.copy()
    t = torch.from_numpy(x.copy()).pin_memory()
    keep.append(t)
    return t.numpy()


al = A.Aligner(0)
al.set_scoring(2, -2, -5, -2)
lib = A.load_library()
p = lambda a: a.ctypes.data_as(C.c_void_p)
for pinned in (False, True):
    keep = []
    blob = arr(base, pinned)
    off1 = arr(np.arange(n, dtype=np.int64) * (l1 + l2), pinned)
    off2 = arr(off1 + l1, pinned)
    len1 = arr(np.full(n, l1, dtype=np.int32), pinned)
    len2 = arr(np.full(n, l2, dtype=np.int32), pinned)
    score, ei, ej, st, nops = (arr(np.zeros(n, np.int32), pinned) for _ in range(5))
    ops = arr(np.zeros(n * (l1 + l2) + 64, np.uint8), pinned)
    for tb in (1, 0):
        ts = []
        for it in range(6):
            t0 = time.perf_counter()
            rc = lib.at_align_batch(al._h, A.MODE_LOCAL, n, p(blob), p(off1), p(len1), p(off2), p(len2), tb, p(score), p(ei), p(ej), p(st),
                                    p(ops) if tb else None, p(off1) if tb else None, p(nops) if tb else None)
            ts.append(time.perf_counter() - t0)
            assert rc == 0
        t = min(ts[1:])
        print("at_align_batch, %s caller buffers, traceback=%d: %.2f ms per 100k pairs = %.1f GCUPS" % ("pinned" if pinned else "pageable", tb, t * 1e3, n * l1 * l2 / t / 1e9), flush=True)
