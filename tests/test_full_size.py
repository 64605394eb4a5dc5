"""GPU suite: BASELINE configs[3] and configs[4] at the sizes they state, on ONE MI355X (they are quoted for 8: one card holds
them too) -- C4, fit with the jump state over 10 000 000 pairs of 150 x 500 in one launch; C5, overlap all-vs-all over 50 000
reads of 1 kbp = 1 249 975 000 pairs streamed in slices.  The oracle finishes samples in seconds; every pair is covered by
properties that do not need it (a traceback re-scores to its reported score; the same pairs through other slice sizes give the
same bits).  Inputs are generated in HBM by the torch forms of the synthetic generators (tests/test_synth.py: bit-identical to
the numpy ones)."""
import random

import numpy as np
import pytest

import oracle as O   # test infrastructure: the checker

pytestmark = [pytest.mark.gpu, pytest.mark.slow]


def test_c4_ten_million_pairs_in_one_launch():
    """alignment.h:596-694 + :558-592 at C4's full 10 M pairs: 64-bit slot offsets beyond 2^32, 625 000 work items through the
    work queue, pointer slots reused 300 times.  Every pair: ends in row l1 at a column < l2, and its ops (END -> START) re-score
    to the reported score -- matches, mismatches, affine gaps, one gamma per run of JUMP ops -- and reach row 0; 500 sampled pairs
    equal the oracle bit for bit (score, end cell, state, ops)."""
    import torch
    import aligntools.c_amd as A
    from aligntools.c_amd import synth as S
    n, l1, l2 = 10_000_000, 150, 500
    m, u, o, e, g = 2, -2, -5, -1, -10
    sites = [100, 200, 300, 400]
    seed = 0x5EED0004
    dev = torch.device("cuda", 0)
    w1, w2 = (l1 + 15) // 16 + 1, (l2 + 15) // 16 + 1
    d_words = torch.empty((n, w1 + w2), dtype=torch.int32, device=dev)
    d_codes = torch.empty((n, l1 + l2), dtype=torch.uint8, device=dev)       # kept for the re-scoring
    slab = 500_000
    for lo in range(0, n, slab):
        c = S.workload_codes_torch("fit", True, seed, slab, l1, l2, lo, dev)
        d_codes[lo:lo + slab] = c
        d_words[lo:lo + slab, :w1] = S.pack2_torch(c[:, :l1])
        d_words[lo:lo + slab, w1:] = S.pack2_torch(c[:, l1:])
        del c
    base = torch.arange(n, dtype=torch.int64, device=dev) * (w1 + w2)
    d_woff1, d_woff2 = base, base + w1
    d_len1 = torch.full((n,), l1, dtype=torch.int32, device=dev)
    d_len2 = torch.full((n,), l2, dtype=torch.int32, device=dev)
    d_ops_off = torch.arange(n, dtype=torch.int64, device=dev) * (l1 + l2)
    d_ops = torch.zeros(n * (l1 + l2) + 64, dtype=torch.uint8, device=dev)
    res = torch.zeros((5, n), dtype=torch.int32, device=dev)
    al = A.Aligner(0)
    al.set_scoring(m, u, o, e, g, True, sites)
    al.align_batch_device(A.MODE_FIT, n, d_words.data_ptr(), 2, d_woff1.data_ptr(), d_len1.data_ptr(), d_woff2.data_ptr(), d_len2.data_ptr(),
                          l1, l2, True, True, res[0].data_ptr(), res[1].data_ptr(), res[2].data_ptr(), res[3].data_ptr(),
                          d_ops.data_ptr(), d_ops_off.data_ptr(), res[4].data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert "packed16" in al.last_config and "8-lane" in al.last_config, al.last_config
    score, ei, ej, st, nops = (res[k] for k in range(5))
    assert int(nops.min()) >= l1 and int(nops.max()) <= l1 + l2 and int(score.min()) > -(1 << 30)
    assert bool((ei == l1).all()) and bool(((ej >= 0) & (ej < l2)).all())
    # ---- every pair: the ops re-score to the score (slabs of 250 000 pairs, on the GPU: plain tensor arithmetic, no kernel of ours) ----
    ops2d = d_ops[:n * (l1 + l2)].view(n, l1 + l2)
    col = torch.arange(l1 + l2, device=dev)[None, :]
    bad = 0
    for lo in range(0, n, 250_000):
        hi = lo + 250_000
        a = ops2d[lo:hi].to(torch.int16)
        valid = col < nops[lo:hi, None]
        di = ((a <= 1) & valid).to(torch.int32)
        dj = ((a != 1) & valid).to(torch.int32)
        i = ei[lo:hi, None] - torch.cumsum(di, 1, dtype=torch.int32)
        j = ej[lo:hi, None] - torch.cumsum(dj, 1, dtype=torch.int32)
        walked_off = ((i < 0) | (j < 0)) & valid
        mid = (a == 0) & valid
        c1 = torch.gather(d_codes[lo:hi, :l1], 1, i.clamp(0, l1 - 1).long())
        c2 = torch.gather(d_codes[lo:hi, l1:], 1, j.clamp(0, l2 - 1).long())
        sc = torch.where(mid, torch.where(c1 == c2, m, u), 0).sum(1, dtype=torch.int32)
        start = torch.ones_like(valid)
        start[:, 1:] = a[:, 1:] != a[:, :-1]
        gap = ((a == 1) | (a == 2)) & valid
        jump = (a == 3) & valid
        sc = sc + o * (gap & start).sum(1, dtype=torch.int32) + e * (gap & ~start).sum(1, dtype=torch.int32) + g * (jump & start).sum(1, dtype=torch.int32)
        rows_left = ei[lo:hi] - di.sum(1, dtype=torch.int32)                     # fit: the walk ends in row 0 (:561 while(i>0))
        bad += int(((sc != score[lo:hi]) | (rows_left != 0) | walked_off.any(1)).sum())
        del a, valid, di, dj, i, j, mid, c1, c2, sc, start, gap, jump
    assert bad == 0
    # ---- 500 sampled pairs against the oracle, from the first, the last and random places ----
    rng = random.Random(4)
    sample = sorted(set([0, 1, n - 2, n - 1] + [rng.randrange(n) for _ in range(496)]))
    idx = torch.tensor(sample, dtype=torch.int64, device=dev)
    asc = np.frombuffer(b"ACGT", dtype=np.uint8)[d_codes[idx].cpu().numpy()]
    h = res[:, idx].cpu().numpy()
    hops = ops2d[idx].cpu().numpy()
    jumps = 0
    for q, k in enumerate(sample):
        r = O.align(O.FIT, asc[q, :l1].tobytes(), asc[q, l1:].tobytes(), m, u, o, e, g, True, sites)
        assert (int(h[0, q]), int(h[1, q]), int(h[2, q]), int(h[3, q]), hops[q, :h[4, q]].tobytes()) == \
               (r["score"], r["end_i"], r["end_j"], r["state"], r["ops"]), k
        jumps += b"\x03" in r["ops"]
    assert jumps > 50          # the jump state is used


def test_c5_all_vs_all_fifty_thousand_reads_streamed():
    """alignment.h:926-964 at C5's full size: 50 000 reads of 1 kbp, all 1 249 975 000 ordered pairs a < b, scores and end cells
    through at_align_allpairs_stream in slices of 8 Mi pairs (device and host memory bounded by the slice: the whole result would
    be 20 GB).  The callback sees every pair once, in order; three windows of 40 M pairs -- the first, one across the middle,
    the last -- are kept and equal, bit for bit, what the same pairs give through slices of another size; pairs sampled from the
    first, a middle and the last slice equal the oracle."""
    import aligntools.c_amd as A
    from aligntools.c_amd.synth import synth_pairs_blob
    nreads, L = 50_000, 1000
    blob = synth_pairs_blob(0x5EED0005, nreads // 2, L, L).reshape(-1).copy()      # 25 000 rows of two reads
    # a few real overlaps among the unrelated reads: read 2k + 1 starts with the last 100 .. 800 bases of read 2k, every 40th base changed
    prng = random.Random(77)
    for k in list(range(0, 400, 2)) + list(range(nreads - 400, nreads, 2)) + [prng.randrange(nreads // 2) * 2 for _ in range(300)]:
        ov = prng.randint(100, 800)
        piece = blob[(k + 1) * L - ov:(k + 1) * L].copy()
        piece[::40] = ord("A")
        blob[(k + 1) * L:(k + 1) * L + ov] = piece
    off = np.arange(nreads, dtype=np.int64) * L
    lens = np.full(nreads, L, dtype=np.int32)
    total = nreads * (nreads - 1) // 2
    assert total == 1_249_975_000
    al = A.Aligner(0)
    al.set_scoring(1, -2, -5, -1)
    W = 40_000_000
    windows = [0, total // 2 - W // 2, total - W]
    kept = [np.zeros((3, W), dtype=np.int32) for _ in windows]
    state = {"next": 0, "slices": 0, "sum": 0, "neg": 0}

    def on_slice(first, sc, ei, ej, st):
        assert first == state["next"]
        n = len(sc)
        state["next"] = first + n
        state["slices"] += 1
        state["sum"] += int(sc.sum(dtype=np.int64))
        state["neg"] += int((sc < 0).sum()) + int((ei != L).sum()) + int(((ej < 0) | (ej >= L)).sum())
        for w, lo in enumerate(windows):
            a, b = max(first, lo), min(first + n, lo + W)
            if a < b:
                for row, x in enumerate((sc, ei, ej)):
                    kept[w][row, a - lo:b - lo] = x[a - first:b - first]
    chunk = 8 << 20
    al.align_allpairs_stream("overlap", blob, off, lens, 0, total, chunk, on_slice)
    assert state["next"] == total and state["slices"] == (total + chunk - 1) // chunk and state["neg"] == 0
    assert "rows/lane=16" in al.last_config and "slices" in al.last_config, al.last_config
    # the same windows through slices of 3 000 000 pairs (does not divide anything)
    for w, lo in enumerate(windows):
        again = np.zeros((3, W), dtype=np.int32)

        def on2(first, sc, ei, ej, st, again=again, lo=lo):
            for row, x in enumerate((sc, ei, ej)):
                again[row, first - lo:first - lo + len(x)] = x
        al.align_allpairs_stream("overlap", blob, off, lens, lo, W, 3_000_000, on2)
        assert (again == kept[w]).all(), w
    # the oracle on pairs of the first, a middle and the last slice
    rng = random.Random(5)
    for w, lo in enumerate(windows):
        for q in [0, W - 1] + [rng.randrange(W) for _ in range(14)]:
            p = lo + q
            a = 0
            # row of pair p in the strict upper triangle
            a = int((2 * nreads - 1 - ((2 * nreads - 1) ** 2 - 8 * p) ** 0.5) / 2)
            while a > 0 and a * (2 * nreads - a - 1) // 2 > p:
                a -= 1
            while (a + 1) * (2 * nreads - a - 2) // 2 <= p:
                a += 1
            b = p - a * (2 * nreads - a - 1) // 2 + a + 1
            ref = O.align(O.OVERLAP, blob[a * L:(a + 1) * L].tobytes(), blob[b * L:(b + 1) * L].tobytes(), 1, -2, -5, -1)
            assert (int(kept[w][0, q]), int(kept[w][1, q]), int(kept[w][2, q])) == (ref["score"], ref["end_i"], ref["end_j"]), (p, a, b)

    # ---- the same triangle with a threshold (at_set_min_score, the bit-parallel overlap filter of at_myers.hip.h): pairs proven below
    # T are not swept.  The whole triangle in well under a minute; in the three windows every pair the filter let through carries the
    # exact results of the run above, every pair it stopped scores below T there, and nothing that reaches T was stopped.
    import time
    T = 30
    kept2 = [np.zeros((2, W), dtype=np.int32) for _ in windows]
    st2 = {"next": 0, "swept": 0, "hits": 0}

    def on3(first, sc, ei, ej, st):
        assert first == st2["next"]
        n = len(sc)
        st2["next"] = first + n
        st2["swept"] += int((st == 2).sum())
        st2["hits"] += int(((st == 2) & (sc >= T)).sum())
        assert (((st == 0) & (sc < T)) | (st == 2)).all()
        for w, lo in enumerate(windows):
            a, b = max(first, lo), min(first + n, lo + W)
            if a < b:
                kept2[w][0, a - lo:b - lo] = sc[a - first:b - first]
                kept2[w][1, a - lo:b - lo] = st[a - first:b - first]
    al.set_min_score(T)
    t0 = time.time()
    al.align_allpairs_stream("overlap", blob, off, lens, 0, total, chunk, on3)
    dt = time.time() - t0
    al.set_min_score(None)
    assert "overlap filter" in al.last_config, al.last_config
    assert st2["next"] == total
    assert dt < 60.0, dt                                   # (the unthresholded triangle takes 110 s)
    assert st2["swept"] < total // 100, st2["swept"]       # unrelated reads are stopped by the bound
    nhit = 0
    for w in range(len(windows)):
        swept = kept2[w][1] == 2
        assert (kept2[w][0][swept] == kept[w][0][swept]).all(), w
        assert (kept[w][0][~swept] < T).all(), w
        assert (kept2[w][0][~swept] >= kept[w][0][~swept]).all(), w
        nhit += int((kept[w][0] >= T).sum())
    assert nhit >= 300 and st2["hits"] >= nhit              # the planted overlaps of the first and last reads are found
    print("C5 triangle with --min-score %d: %.1f s, %d of %d pairs swept, %d score >= T" % (T, dt, st2["swept"], total, st2["hits"]))
    al.close()
