#!/bin/bash
# round 3, call S: the chunks of a host batch send their inputs up one after the other (AT_HOST_ORDERED_UPLOADS) -- parity, then the rates A/B
set -e
export TMPDIR=/tmp
O=gpurun_out/r03s
mkdir -p $O
python3 -m pytest tests/test_gpu_parity.py tests/test_default_routing.py tests/test_cli.py -x -q -m gpu -k "chunk or host or routing or batch or large or cli" 2>&1 | tail -3
cat > /tmp/hp.py <<'PY'
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import aligntools.c_amd as A
from aligntools.c_amd.synth import synth_pairs_blob
n, l1, l2 = 100000, 150, 150
blob = synth_pairs_blob(0x5EED0002, n, l1, l2).reshape(-1).copy()
off1 = np.arange(n, dtype=np.int64) * (l1 + l2); off2 = off1 + l1
len1 = np.full(n, l1, dtype=np.int32); len2 = np.full(n, l2, dtype=np.int32)
score, ei, ej, st, nops = (np.zeros(n, np.int32) for _ in range(5))
ops = np.zeros(n * (l1 + l2) + 64, np.uint8)
al = A.Aligner(0); al.set_scoring(2, -2, -5, -2); lib = A.load_library()
p = lambda a: a.ctypes.data_as(C.c_void_p)
for tb in (1, 0):
    ts = []
    for it in range(60):
        t0 = time.perf_counter()
        rc = lib.at_align_batch(al._h, A.MODE_LOCAL, n, p(blob), p(off1), p(len1), p(off2), p(len2), tb, p(score), p(ei), p(ej), p(st), p(ops) if tb else None, p(off1) if tb else None, p(nops) if tb else None)
        ts.append(time.perf_counter() - t0); assert rc == 0
    ts = np.array(ts[8:]) * 1e3
    print("ordered=%s chunks=%s tb=%d: min %.2f median %.2f mean %.2f max %.2f ms = %.0f GCUPS at the median" % (os.environ.get("AT_HOST_ORDERED_UPLOADS", "1"), os.environ.get("AT_HOST_CHUNKS", "6"), tb, ts.min(), np.median(ts), ts.mean(), ts.max(), n * l1 * l2 / np.median(ts) / 1e6), flush=True)
PY
for ord in 1 0 1 0; do AT_HOST_ORDERED_UPLOADS=$ord python3 /tmp/hp.py 2>/dev/null; done
for ch in 4 8 12; do AT_HOST_CHUNKS=$ch python3 /tmp/hp.py 2>/dev/null; done
AT_HOST_TRACE=1 python3 /tmp/hp.py > /dev/null 2> $O/trace_ordered.txt
