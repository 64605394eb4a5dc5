"""Edit distance with unit mismatch cost (`edit -u 1`) through the host entry: bit-parallel kernel (at_myers.hip.h)
against the cell-by-cell kernel (AT_MYERS=0).  10k pairs of 1000 x 1000 and 100k pairs of 150 x 150."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aligntools.c_amd as A
from aligntools.c_amd.synth import synth_pairs_blob

lib = A.load_library()
p = lambda a: a.ctypes.data_as(C.c_void_p)
for n, l in ((10000, 1000), (100000, 150)):
    blob = synth_pairs_blob(0x5EED0007, n, l, l).reshape(-1).copy()
    off1 = np.arange(n, dtype=np.int64) * 2 * l
    off2 = off1 + l
    len1 = np.full(n, l, dtype=np.int32)
    len2 = np.full(n, l, dtype=np.int32)
    score = np.zeros(n, np.int32)
    for myers in ("1", "0"):
        os.environ["AT_MYERS"] = myers
        al = A.Aligner(0)
        al.set_scoring(1, 1, -5, -1)
        ts = []
        for it in range(5):
            t0 = time.perf_counter()
            rc = lib.at_align_batch(al._h, A.MODE_EDIT, n, p(blob), p(off1), p(len1), p(off2), p(len2), 0, p(score), None, None, None,
                                    None, None, None)
            ts.append(time.perf_counter() - t0)
            assert rc == 0
        t = min(ts[1:])
        print("%6d x %4d^2  AT_MYERS=%s  %.2f ms = %.0f GCUPS host path, checksum %d (%s)" % (n, l, myers, t * 1e3, n * l * l / t / 1e9,
                                                                                           int(score.astype(np.int64).sum()), al.last_config[:40]))
        al.close()
