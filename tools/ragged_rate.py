"""Host-entry rate (at_align_batch: upload, GPU packing, sweep, download) on RAGGED batches -- every pair its own lengths,
packed kernels in frames -- against uniform batches of the workload's nominal shape.  100k pairs, tracebacks.

    local   150 x 150        vs   100..150 x 100..150 and 30..150 x 30..150
    fit -s  150 x 500 (C4)   vs   100..150 x 400..500
    global  150 x 150        vs   100..150 x 100..150
    local   250 x 250        vs   200..250 x 200..250          global  300 x 300   vs   250..300 x 250..300
    local / global 500 x 500 vs   400..500 x 400..500          overlap 1000 x 1000 vs  800..1000 x 800..1000   (20k pairs)
AT_RAGGED_PACKED=0 keeps ragged batches on the int32 kernel (the rate before frames).
"""
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
lib = A.load_library()
p = lambda a: a.ctypes.data_as(C.c_void_p)
CASES = [
    ("local", (2, -2, -5, -2, -10), False, [], 150, 150, [("uniform 150x150", 150, 150), ("ragged 100..150 x 100..150", 100, 100), ("ragged 30..150 x 30..150", 30, 30)]),
    ("fit", (2, -2, -5, -1, -10), True, [100, 200, 300, 400], 150, 500, [("uniform 150x500 (C4)", 150, 500), ("ragged 100..150 x 400..500", 100, 400)]),
    ("global", (1, -1, -4, -1, -10), False, [], 150, 150, [("uniform 150x150", 150, 150), ("ragged 100..150 x 100..150", 100, 100)]),
    ("local", (2, -2, -5, -2, -10), False, [], 250, 250, [("uniform 250x250", 250, 250), ("ragged 200..250 x 200..250", 200, 200)]),
    ("global", (1, -1, -4, -1, -10), False, [], 300, 300, [("uniform 300x300", 300, 300), ("ragged 250..300 x 250..300", 250, 250)]),
    # reads of 305 .. 608 bases: frames on the 32-lane groups (round 3); ragged overlap on the packed overlap kernel (n = 20 000)
    ("local", (2, -2, -5, -2, -10), False, [], 500, 500, [("uniform 500x500", 500, 500), ("ragged 400..500 x 400..500", 400, 400)]),
    ("global", (1, -1, -4, -1, -10), False, [], 500, 500, [("uniform 500x500", 500, 500), ("ragged 400..500 x 400..500", 400, 400)]),
    ("overlap", (1, -2, -5, -1, -10), False, [], 1000, 1000, [("uniform 1000x1000", 1000, 1000), ("ragged 800..1000 x 800..1000", 800, 800)]),
]
if len(sys.argv) > 1:   # e.g. `ragged_rate.py 250 300`: only the cases of those read lengths
    CASES = [c for c in CASES if str(c[4]) in sys.argv[1:]]
N_ALL = n
for mode, sc, uj, sites, L1, L2, rows in CASES:
    n = N_ALL if L1 * L2 <= 100000 else N_ALL // 5          # (the long-read cases: 20 000 pairs)
    al.set_scoring(*sc, uj, sites)
    blob = synth_pairs_blob(0x5EED0002, n, L1, L2).reshape(-1).copy()
    off1 = np.arange(n, dtype=np.int64) * (L1 + L2)
    off2 = off1 + L1
    base = None
    for name, lo1, lo2 in rows:
        len1 = rng.integers(lo1, L1 + 1, n).astype(np.int32)
        len2 = rng.integers(lo2, L2 + 1, n).astype(np.int32)
        score, ei, ej, st, nops = (np.zeros(n, np.int32) for _ in range(5))
        ops = np.zeros(n * (L1 + L2) + 64, np.uint8)
        ts = []
        for it in range(5):
            t0 = time.perf_counter()
            rc = lib.at_align_batch(al._h, A.MODES[mode], n, p(blob), p(off1), p(len1), p(off2), p(len2), 1, p(score), p(ei), p(ej), p(st),
                                    p(ops), p(off1), p(nops))
            ts.append(time.perf_counter() - t0)
            assert rc == 0, lib.at_last_error(al._h)
        t = min(ts[1:])
        cells = float((len1.astype(np.int64) * len2).sum())
        rate = cells / t / 1e9
        base = base or rate
        print("%-7s %-28s %7.2f ms = %5.0f GCUPS host path, %3.0f %% of uniform (%s)" % (mode + (" -s" if uj else ""), name, t * 1e3, rate, 100 * rate / base,
                                                                                   al.last_config[:90]), flush=True)
