"""PCIe-inclusive rate of the host-buffer entry at_align_batch (pack on host + H2D + kernel + D2H), C2 workload.
Not the bench `value` (which has inputs resident in HBM); quoted in DESIGN.md section 7."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aligntools.c_amd as A
from aligntools.c_amd.synth import synth_pairs_blob

n, l1, l2 = 100000, 150, 150
blob = synth_pairs_blob(0x5EED0002, n, l1, l2).reshape(-1).copy()
off1 = np.arange(n, dtype=np.int64) * (l1 + l2)
off2 = off1 + l1
len1 = np.full(n, l1, dtype=np.int32)
len2 = np.full(n, l2, dtype=np.int32)
score = np.zeros(n, np.int32); ei = np.zeros(n, np.int32); ej = np.zeros(n, np.int32); st = np.zeros(n, np.int32); nops = np.zeros(n, np.int32)
ops = np.zeros(n * (l1 + l2) + 64, np.uint8)
al = A.Aligner(0)
al.set_scoring(2, -2, -5, -2)
lib = A.load_library()
p = lambda a: a.ctypes.data_as(C.c_void_p)
for tb in (1, 0):
    ts = []
    for it in range(6):
        t0 = time.perf_counter()
        rc = lib.at_align_batch(al._h, A.MODE_LOCAL, n, p(blob), p(off1), p(len1), p(off2), p(len2), tb, p(score), p(ei), p(ej), p(st),
                                p(ops) if tb else None, p(off1) if tb else None, p(nops) if tb else None)
        ts.append(time.perf_counter() - t0)
        assert rc == 0
    t = min(ts[1:])
    print("at_align_batch host path, traceback=%d: %.2f ms per 100k pairs = %.1f GCUPS (%s)" % (tb, t * 1e3, n * l1 * l2 / t / 1e9, al.last_config))
