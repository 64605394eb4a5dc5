"""Host-entry rate by alphabet on C2's shape (100k x 150 x 150, local, tracebacks): pure ACGT (2-bit words), reads with a
few N (byte words), protein (byte words).  All three run on the packed kernel."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aligntools.c_amd as A
from aligntools.c_amd.synth import synth_pairs_blob

n, l1, l2 = 100000, 150, 150
rng = np.random.default_rng(3)
al = A.Aligner(0)
al.set_scoring(2, -2, -5, -2)
lib = A.load_library()
p = lambda a: a.ctypes.data_as(C.c_void_p)
base = synth_pairs_blob(0x5EED0002, n, l1, l2).reshape(-1).copy()
withn = base.copy()
withn[rng.integers(0, len(withn), 2000)] = ord("N")
prot = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)[rng.integers(0, 20, len(base))].copy()
off1 = np.arange(n, dtype=np.int64) * (l1 + l2)
off2 = off1 + l1
len1 = np.full(n, l1, dtype=np.int32)
len2 = np.full(n, l2, dtype=np.int32)
score, ei, ej, st, nops = (np.zeros(n, np.int32) for _ in range(5))
ops = np.zeros(n * (l1 + l2) + 64, np.uint8)
for name, blob in (("ACGT", base), ("ACGT + 2000 N", withn), ("protein", prot)):
    ts = []
    for it in range(5):
        t0 = time.perf_counter()
        rc = lib.at_align_batch(al._h, A.MODE_LOCAL, n, p(blob), p(off1), p(len1), p(off2), p(len2), 1, p(score), p(ei), p(ej), p(st),
                                p(ops), p(off1), p(nops))
        ts.append(time.perf_counter() - t0)
        assert rc == 0
    t = min(ts[1:])
    print("%-14s %.2f ms = %.0f GCUPS host path (%s)" % (name, t * 1e3, n * l1 * l2 / t / 1e9, al.last_config[:75]))
