#!/bin/bash
# round 3, call Z: bit-parallel edit distance all-vs-all (pairs enumerated in the kernel): parity, rate against the cell-by-cell kernel
set -e
export TMPDIR=/tmp
O=gpurun_out/r03z
mkdir -p $O
python3 -m pytest tests/test_gpu_parity.py tests/test_cli.py -x -q -m gpu -k "all_vs_all or allpairs or edit" 2>&1 | tail -3
python3 - <<'PY'
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import aligntools.c_amd as A
from aligntools.c_amd.synth import synth_pairs_blob
nreads, L = 4000, 1000
blob = synth_pairs_blob(0x5EED0005, nreads // 2, L, L).reshape(nreads, L)
words, woff, _w2, lens, _l2, bits = A.pack_pairs([(bytes(r), b"") for r in blob])
dev = torch.device("cuda", 0)
d_words = torch.from_numpy(words.view(np.int32)).to(dev); d_woff = torch.from_numpy(woff).to(dev); d_len = torch.from_numpy(lens).to(dev)
total = nreads * (nreads - 1) // 2
res = torch.zeros((5, total), dtype=torch.int32, device=dev)
al = A.Aligner(0); al.set_scoring(1, 1, -5, -1)
out = {}
for myers in ("1", "0"):
    os.environ["AT_MYERS"] = myers
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        al.align_allpairs_device(A.MODES["edit"], nreads, d_words.data_ptr(), bits, d_woff.data_ptr(), d_len.data_ptr(), L, 0, total, False,
                                 res[0].data_ptr(), res[1].data_ptr(), res[2].data_ptr(), res[3].data_ptr(), 0, 0, res[4].data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize(); t = time.perf_counter() - t0
    out[myers] = res[0].clone()
    print("AT_MYERS=%s: %d reads of %d bases all-vs-all, %d pairs: %.1f ms = %.0f GCUPS (%s)" % (myers, nreads, L, total, t * 1e3, total * L * L / t / 1e9, al.last_config[:70]), flush=True)
assert torch.equal(out["1"], out["0"]), "bit-parallel and cell-by-cell all-vs-all disagree"
print("equal on all", total, "pairs")
PY
