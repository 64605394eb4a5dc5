"""GPU suite (-m gpu): the HIP path, called through the C ABI, against
  (1) the golden vectors the real reference produced (tests/golden), bit-exact
      score and both gapped strings, and
  (2) the oracle restatement on fresh seeded inputs (ops, end cell, start state).
"""
import hashlib
import os
import random
from collections import defaultdict

import numpy as np
import pytest

import oracle as O
from conftest import load_golden

pytestmark = pytest.mark.gpu


def _md5(s):
    return hashlib.md5(s.encode("latin1")).hexdigest()


@pytest.fixture(scope="module")
def al():
    import os
    import aligntools.c_amd as A
    before = os.environ.get("AT_PACKED_MIN_ROUNDS")
    os.environ["AT_PACKED_MIN_ROUNDS"] = "0"   # small test batches must still reach the 64-lane packed kernels
    a = A.Aligner()
    yield a
    a.close()
    if before is None:                          # (the default routing is tests/test_default_routing.py's subject)
        os.environ.pop("AT_PACKED_MIN_ROUNDS", None)
    else:
        os.environ["AT_PACKED_MIN_ROUNDS"] = before


def _group(cases):
    g = defaultdict(list)
    for c in cases:
        g[(c["mode"], c["m"], c["u"], c["o"], c["e"], c["j"], c["use_jump"], tuple(c["sites"]))].append(c)
    return g


def _run_group(al, key, cases):
    mode, m, u, o, e, j, uj, sites = key
    al.set_scoring(m, u, o, e, j, uj, list(sites))
    res = al.align_batch(mode, [(c["s1"], c["s2"]) for c in cases])
    for k, c in enumerate(cases):
        assert int(res["score"][k]) == c["score"], (key, c["tag"], k, c["s1"][:40], c["s2"][:40])
        if mode == "edit":
            continue
        if "r1" in c:
            assert res["r1"][k] == c["r1"] and res["r2"][k] == c["r2"], (key, c["tag"], k)
        else:
            assert len(res["r1"][k]) == c["rlen"]
            assert _md5(res["r1"][k]) == c["r1_md5"] and _md5(res["r2"][k]) == c["r2_md5"], (key, c["tag"])
    if mode != "edit":
        # the same strings rendered on the GPU (at_render.hip.h) instead of by at_render on the host
        gs = al.align_batch_strings(mode, [(c["s1"], c["s2"]) for c in cases])
        assert gs["r1"] == res["r1"] and gs["r2"] == res["r2"], key
        assert (gs["score"] == res["score"]).all() and (gs["nops"] == res["nops"]).all()


@pytest.mark.parametrize("name", ["random_small.jsonl", "random_dna.jsonl", "known_answers.jsonl", "known_answers_big.jsonl"])
def test_hip_matches_reference_goldens(al, name):
    groups = _group(load_golden(name))
    for key, cases in groups.items():
        _run_group(al, key, cases)


def test_hip_matches_oracle_ops_and_end_cells(al):
    """ops (END->START), end cell and start state equal the restatement's."""
    rng = random.Random(4242)
    for mode in ("global", "local", "fit", "overlap"):
        for sc in ((2, -2, -5, -2), (1, -1, -1, -1), (1, -2, -5, -1)):
            pairs = []
            for _ in range(120):
                l1, l2 = rng.randint(1, 200), rng.randint(2, 260)
                if mode == "fit":
                    l1 = min(l1, l2)
                s1 = "".join(rng.choice("ACGT") for _ in range(l1))
                if rng.random() < 0.5:
                    s2 = "".join(rng.choice("ACGT") for _ in range(l2))
                else:
                    s2 = ("".join(rng.choice("ACGT") for _ in range(rng.randint(0, 30))) + s1[: rng.randint(1, l1)] +
                          "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 30))))
                    if mode == "fit" and len(s2) < len(s1):
                        s2 = s2 + s1
                pairs.append((s1, s2))
            uj = mode == "fit" and sc[0] == 2
            sites = [50, 100, 101, 150] if uj else []
            al.set_scoring(*sc, -7, uj, sites)
            res = al.align_batch(mode, pairs)
            for k, (s1, s2) in enumerate(pairs):
                r = O.align(O.MODE_NAMES[mode], s1, s2, *sc, -7, uj, sites)
                assert r["rc"] == 0
                assert int(res["score"][k]) == r["score"]
                assert (int(res["end_i"][k]), int(res["end_j"][k]), int(res["state"][k])) == (r["end_i"], r["end_j"], r["state"])
                assert res["ops"][k] == r["ops"]


def test_reference_named_surface(al):
    import aligntools.c_amd as A
    opt = A.opt_t(m=2, u=-2, o=-5, e=-2)
    assert A.align_local_affine("PLEASANTLY", "MEANLY", opt) == (4.0, "LEA", "MEA")
    assert A.edit_dist("kitten", "sitting", A.opt_t(u=1)) == 3
    with pytest.raises(A.AlignToolsError) as ei:
        A.align_fit_affine_jump("ACGTACGT", "ACG")
    assert "first sequence must be shorter" in str(ei.value)


@pytest.mark.parametrize("mode", ["local", "global", "fit", "fitj"])
def test_packed16_uniform_batches_match_oracle(al, mode):
    """Uniform-shape DNA batches take the packed two-pairs-per-wave kernel; every score,
    end cell, start state and ops string must equal the oracle's (odd batch sizes,
    1..4 rows per lane, two strips, related pairs with long tracebacks)."""
    rng = random.Random(77)
    shapes = [(150, 150), (40, 90), (100, 64), (130, 200), (250, 260), (300, 320), (1, 5), (64, 64), (65, 8)]
    use_jump = mode == "fitj"
    if use_jump:
        mode = "fit"
    for (l1, l2) in shapes:
        if mode == "fit" and l1 > l2:
            continue
        sites = sorted(rng.sample(range(l2 + 3), min(5, l2))) if use_jump else []
        for sc in ((2, -2, -5, -2), (1, -1, -1, -1), (1, -2, -5, -1), (0, 0, 0, 0)):
            n = rng.choice([1, 2, 7, 33])
            pairs = []
            for _ in range(n):
                s1 = "".join(rng.choice("ACGT") for _ in range(l1))
                style = rng.random()
                if style < 0.4:
                    s2 = "".join(rng.choice("ACGT") for _ in range(l2))
                else:
                    body = list(s1)
                    for _m in range(rng.randint(0, 8)):
                        p = rng.randrange(len(body))
                        body[p] = rng.choice("ACGT")
                    if rng.random() < 0.5 and len(body) > 4:
                        p = rng.randrange(len(body) - 2)
                        del body[p:p + rng.randint(1, 2)]
                    s2 = ("".join(rng.choice("ACGT") for _ in range(rng.randint(0, 6))) + "".join(body))
                    s2 = (s2 + "".join(rng.choice("ACGT") for _ in range(l2)))[:l2]
                pairs.append((s1, s2))
            jp = rng.choice([-10, -3, 0])
            al.set_scoring(*sc, jp, use_jump, sites)
            res = al.align_batch(mode, pairs)
            assert "packed16" in al.last_config, al.last_config
            for k, (s1, s2) in enumerate(pairs):
                r = O.align(O.MODE_NAMES[mode], s1, s2, *sc, jp, use_jump, sites)
                assert r["rc"] == 0
                assert int(res["score"][k]) == r["score"], (mode, l1, l2, sc, k)
                assert (int(res["end_i"][k]), int(res["end_j"][k]), int(res["state"][k])) == (r["end_i"], r["end_j"], r["state"]), (mode, l1, l2, sc, k)
                assert res["ops"][k] == r["ops"], (mode, l1, l2, sc, k)


def _rescore(ops, s1, s2, ei, ej, m, u, o, e, mode):
    """Independent check: walk the ops (END->START) and add up the score the path implies."""
    i, j, sc, prev = ei, ej, 0, None
    for op in ops:
        if op == 0:
            i -= 1; j -= 1
            sc += m if s1[i] == s2[j] else u
        else:
            if op == 1:
                i -= 1
            else:
                j -= 1
            sc += e if prev == op else o     # interior gap of length k costs o + e*(k-1)
        prev = op
    return sc, i, j


def _rescore_np(ops, s1, s2, ei, ej, m, u, o, e, g, mode):
    """The same check for every mode, vectorised: the score the ops (END -> START) imply from the end cell, and where the
    walk ends.  Gap runs cost o + e (k - 1) (overlap: o per op; a JUMP run: g once), the padding run with which a global
    alignment reaches the origin along a border costs o + e k (alignment.h:429-441 against :456,460)."""
    a = np.frombuffer(ops, dtype=np.uint8).astype(np.int64)
    if a.size == 0:
        return 0, ei, ej
    b1, b2 = np.frombuffer(s1, dtype=np.uint8), np.frombuffer(s2, dtype=np.uint8)
    di, dj = (a <= 1).astype(np.int64), (a != 1).astype(np.int64)
    i, j = ei - np.cumsum(di), ej - np.cumsum(dj)           # the cell each op leaves the walk in
    assert i.min() >= 0 and j.min() >= 0
    mid = a == 0
    sc = int(np.where(b1[i[mid]] == b2[j[mid]], m, u).sum())
    start = np.ones(a.size, dtype=bool)
    start[1:] = a[1:] != a[:-1]                             # first op of a run
    if mode == "overlap":
        sc += o * int((~mid).sum())
    else:
        gap = (a == 1) | (a == 2)
        sc += o * int((gap & start).sum()) + e * int((gap & ~start).sum()) + g * int(((a == 3) & start).sum())
        if mode == "global":
            # ops behind the first arrival on a border are the padding run: one more e than an interior gap of its length
            on_border = np.flatnonzero((i == 0) | (j == 0))
            if on_border.size and on_border[0] + 1 < a.size:
                sc += e
    return sc, int(i[-1]), int(j[-1])


def test_full_size_c2_batch_properties(al):
    """BASELINE configs[1] at full size (100k x 150x150 local): size-independent properties on every pair
    -- the ops string re-scores to the reported score, starts inside the matrix, ends in a HOME cell --
    and bit-exact equality with the oracle on a seeded sample."""
    from aligntools.c_amd.synth import synth_pairs_blob
    n, l1, l2 = 100000, 150, 150
    blob = synth_pairs_blob(0x5EED0002, n, l1, l2)
    pairs = [(row[:l1].tobytes(), row[l1:].tobytes()) for row in blob]
    al.set_scoring(2, -2, -5, -2)
    res = al.align_batch("local", pairs, render=False)
    assert "packed16" in al.last_config
    assert (res["score"] >= 0).all() and (res["end_i"] >= 1).all() and (res["end_j"] <= l2).all()
    bad = 0
    for k in range(0, n, 7):
        s1, s2 = pairs[k]
        sc, i, j = _rescore(res["ops"][k], s1, s2, int(res["end_i"][k]), int(res["end_j"][k]), 2, -2, -5, -2, "local")
        # the HOME cell is emitted as an aligned pair: its own contribution is clamped away
        first = res["ops"][k][-1:]
        home = (2 if s1[i] == s2[j] else -2) if first == b"\x00" else 0
        if not (sc - home <= int(res["score"][k]) <= sc - min(home, 0)) or i < 0 or j < 0:
            bad += 1
    assert bad == 0
    rng = random.Random(3)
    for k in rng.sample(range(n), 300):
        r = O.align(O.LOCAL, pairs[k][0], pairs[k][1], 2, -2, -5, -2)
        assert (int(res["score"][k]), int(res["end_i"][k]), int(res["end_j"][k]), res["ops"][k]) == (r["score"], r["end_i"], r["end_j"], r["ops"])


def test_edge_cases_empty_ragged_and_ranges(al):
    import aligntools.c_amd as A
    # empty sequences where the reference is defined: global pads, edit counts, overlap of an empty s1
    al.set_scoring(1, -1, -4, -1)
    res = al.align_batch("global", [("", "ACG"), ("AC", ""), ("", "")])
    assert res["score"].tolist() == [-4 - 3, -4 - 2, 0]
    assert (res["r1"], res["r2"]) == (["---", "AC", ""], ["ACG", "--", ""])
    assert al.align_batch("edit", [("", "ACGT"), ("ACG", ""), ("", "")])["score"].tolist() == [4, 3, 0]
    assert al.align_batch("overlap", [("", "ACGT")])["score"].tolist() == [0]
    # outside the reference's domain: reported, not guessed
    for mode, pair in (("local", ("", "ACG")), ("overlap", ("ACG", "")), ("fit", ("", "ACGT"))):
        with pytest.raises(A.AlignToolsError) as ei:
            al.align_batch(mode, [pair])
        assert ei.value.code == -4
    # ragged batch (int32 kernel), lengths straddling the 64*K strip edges, vs the oracle
    rng = random.Random(11)
    pairs = [("".join(rng.choice("ACGT") for _ in range(l1)), "".join(rng.choice("ACGT") for _ in range(l2)))
             for l1, l2 in [(1, 1), (63, 300), (64, 1), (65, 129), (255, 256), (256, 255), (257, 40), (513, 700), (2, 1100)]]
    al.set_scoring(2, -2, -5, -2)
    for mode in ("global", "local", "overlap", "edit"):
        res = al.align_batch(mode, pairs)
        assert "int32" in al.last_config
        for k, (a, b) in enumerate(pairs):
            r = O.align(O.MODE_NAMES[mode], a, b, 2, -2, -5, -2)
            assert int(res["score"][k]) == r["score"]
            if mode != "edit":
                assert res["ops"][k] == r["ops"]
    # scores too large for the byte LUT of the 2-bit kernels -> 8-bit kernels, same answers
    al.set_scoring(20, -30, -50, -10)
    res = al.align_batch("local", pairs[:6])
    assert "bits=8" in al.last_config
    for k, (a, b) in enumerate(pairs[:6]):
        r = O.align(O.LOCAL, a, b, 20, -30, -50, -10)
        assert int(res["score"][k]) == r["score"] and res["ops"][k] == r["ops"]
    # scores that could leave the exact integer range are refused
    al.set_scoring(1 << 20, -(1 << 20), -5, -1)
    with pytest.raises(A.AlignToolsError) as ei:
        al.align_batch("global", pairs[:2])
    assert ei.value.code == -3


def test_long_sequences(al):
    """5 000 x 6 500 bases (20 strips of 256 rows, a 16 MB pointer matrix in the wave's HBM slot), unrelated and related,
    every mode with its traceback, against the oracle."""
    rng = random.Random(77)
    dna = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    a = dna(5000)
    t = list(a)
    for _ in range(300):
        q = rng.randrange(len(t))
        r = rng.random()
        if r < 0.5:
            t[q] = rng.choice("ACGT")
        elif r < 0.75:
            del t[q]
        else:
            t.insert(q, rng.choice("ACGT"))
    rel = (dna(700) + "".join(t) + dna(2000))[:6500]
    pairs = [(a, rel), (a, dna(6500))]
    for mode, sc, uj, sites in (("global", (1, -1, -4, -1, -10), False, []), ("local", (2, -2, -5, -2, -10), False, []),
                                ("fit", (2, -2, -5, -1, -10), False, []), ("fit", (2, -2, -5, -1, -10), True, [1000, 3000, 5000]),
                                ("overlap", (1, -2, -5, -1, -10), False, []), ("edit", (1, 1, -5, -1, -10), False, [])):
        al.set_scoring(*sc, uj, sites)
        res = al.align_batch(mode, pairs, render=False)
        for k, (x, y) in enumerate(pairs):
            r = O.align(O.MODE_NAMES[mode], x, y, *sc, uj, sites)
            assert int(res["score"][k]) == r["score"], (mode, uj, k)
            if mode != "edit":
                assert (int(res["end_i"][k]), int(res["end_j"][k]), int(res["state"][k]), res["ops"][k]) == \
                       (r["end_i"], r["end_j"], r["state"], r["ops"]), (mode, uj, k)


@pytest.mark.parametrize("mode", ["local", "global", "fit"])
def test_packed16_score_range_extremes(al, mode):
    """The packed kernel keeps 16*score in int16 with a -32768 sentinel: drive it to the edges of the
    range the host admits -- all-mismatch and all-match pairs, one-sided gaps, the largest eligible
    shape (two strips of 4 rows per lane) -- and require exact agreement with the fp64 oracle."""
    rng = random.Random(2026)
    for (l1, l2) in [(150, 150), (200, 208), (330, 330), (100, 500), (600, 640), (1024, 1024)]:
        if mode != "fit" and (l1, l2) == (100, 500):
            continue
        a_run, c_run = "A" * l1, "C" * l2
        rnd1 = "".join(rng.choice("ACGT") for _ in range(l1))
        pairs = [(a_run, c_run),                                   # nothing matches: deepest negatives
                 (a_run, ("A" * l2)),                              # everything matches: highest positives
                 (rnd1, (rnd1 * 4)[:l2]),                          # long exact diagonals
                 (rnd1, (rnd1[l1 // 2:] + rnd1 * 4)[:l2]),         # a long gap first
                 ("AC" * (l1 // 2) + "A" * (l1 % 2), ("CA" * l2)[:l2]),   # tie-heavy periodic
                 (a_run, ("C" * (l2 // 2) + "A" * l2)[:l2])]
        for sc in ((2, -2, -5, -2), (1, -2, -5, -1), (1, -1, -4, -1)):
            al.set_scoring(*sc)
            res = al.align_batch(mode, pairs)
            if "packed16" not in al.last_config:                  # shape/scoring not admitted: nothing to test here
                continue
            if (l1, l2) == (1024, 1024) and sc == (1, -1, -4, -1):
                # global / fit: scores x4 (at_sweep16 TS = 2); local has no downward drift and still fits x16
                assert ("packed16 x16" if mode == "local" else "packed16 x4") in al.last_config
            for k, (s1, s2) in enumerate(pairs):
                r = O.align(O.MODE_NAMES[mode], s1, s2, *sc)
                assert r["rc"] == 0
                assert int(res["score"][k]) == r["score"], (mode, l1, l2, sc, k)
                assert (int(res["end_i"][k]), int(res["end_j"][k]), int(res["state"][k])) == (r["end_i"], r["end_j"], r["state"]), (mode, l1, l2, sc, k)
                assert res["ops"][k] == r["ops"], (mode, l1, l2, sc, k)


def test_score_only_batches(al):
    """want_traceback = 0 (the TB = false kernels: no pointer matrix at all): scores, end cells and start
    states still equal the oracle's, for the packed and the int32 kernels."""
    rng = random.Random(9)
    uniform = [("".join(rng.choice("ACGT") for _ in range(150)), "".join(rng.choice("ACGT") for _ in range(180))) for _ in range(37)]
    ragged = [("".join(rng.choice("ACGT") for _ in range(rng.randint(1, 300))), "".join(rng.choice("ACGT") for _ in range(rng.randint(300, 400))))
              for _ in range(21)]
    for pairs, want in ((uniform, "packed16"), (ragged, "int32")):
        for mode, uj in (("local", False), ("global", False), ("fit", False), ("fit", True), ("overlap", False), ("edit", False)):
            al.set_scoring(2, -2, -5, -2, -6, uj, [100, 200])
            res = al.align_batch(mode, pairs, traceback=False)
            if mode in ("local", "global", "fit"):
                assert want in al.last_config, (mode, al.last_config)
            assert "ops" not in res
            for k, (a, b) in enumerate(pairs):
                r = O.align(O.MODE_NAMES[mode], a, b, 2, -2, -5, -2, -6, uj, [100, 200])
                assert int(res["score"][k]) == r["score"], (mode, uj, k)
                if mode != "edit":
                    assert (int(res["end_i"][k]), int(res["end_j"][k]), int(res["state"][k])) == (r["end_i"], r["end_j"], r["state"])


def test_score_only_long_pairs_sixteen_rows_per_lane(al):
    """Scores-only launches of the x4 packed kernels take 16 rows per lane (strips of 1 024 rows) where that is fewer
    lane-steps than strips of 256: 600, 1 024 and 1 600 rows, global and local, against the oracle (the same batches with
    tracebacks keep 4 rows per lane and must agree)."""
    rng = random.Random(1616)
    dna = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    seen16 = False
    for l1, l2 in ((600, 700), (1024, 1024), (1600, 900)):
        pairs = []
        for k in range(10):
            a = dna(l1)
            b = (a[: l1 // 2] + dna(25) + a[l1 // 2 + 40:] + dna(l2))[:l2] if k % 2 else dna(l2)
            pairs.append((a, b))
        for mode, sc in (("global", (1, -1, -4, -1)), ("local", (2, -2, -5, -2))):
            al.set_scoring(*sc)
            res = al.align_batch(mode, pairs, traceback=False)
            assert "packed16" in al.last_config, al.last_config
            if "packed16 x4" in al.last_config:        # (where the scores still fit x16 the kernels keep 4 rows per lane)
                assert "rows/lane=16" in al.last_config, al.last_config
                seen16 = True
            full = al.align_batch(mode, pairs, render=False)
            for k, (a, b) in enumerate(pairs):
                r = O.align(O.MODE_NAMES[mode], a, b, *sc)
                assert int(res["score"][k]) == r["score"] == int(full["score"][k]), (l1, mode, k)
                assert (int(res["end_i"][k]), int(res["end_j"][k])) == (r["end_i"], r["end_j"]), (l1, mode, k)
    assert seen16


def test_packed_overlap_with_tracebacks(al):
    """Uniform overlap batches with tracebacks run on the packed kernel (scores x4, LEFT / DIAGONAL / RIGHT tags): 4 and 16
    rows per lane, one and several strips, true overlaps (suffix of s1 = prefix of s2, with errors), unrelated pairs and empty
    results, 2-bit and byte alphabets, against the oracle; scores-only batches and scorings outside the 16-bit range stay on
    the int32 kernel."""
    rng = random.Random(926)
    for l1, l2, alpha in ((40, 60, "ACGT"), (150, 150, "ACGT"), (256, 200, "ACGTN"), (300, 300, "ACGT"), (1000, 1000, "ACGT"), (1100, 700, "ACGT")):
        dna = lambda n: "".join(rng.choice(alpha) for _ in range(n))
        pairs = []
        for k in range(9):
            a = dna(l1)
            if k % 3 == 0:
                b = dna(l2)
            else:
                ov = rng.randint(5, min(l1, l2) - 2)
                t = list(a[l1 - ov:])
                for _ in range(ov // 20):
                    q = rng.randrange(len(t))
                    x = rng.random()
                    if x < 0.5:
                        t[q] = rng.choice(alpha)
                    elif x < 0.75 and len(t) > 2:
                        del t[q]
                    else:
                        t.insert(q, rng.choice(alpha))
                b = ("".join(t) + dna(l2))[:l2]
            pairs.append((a, b))
        for sc in ((1, -2, -5, -1), (2, -1, -1, -1), (1, -1, 0, 0)):
            al.set_scoring(*sc)
            res = al.align_batch("overlap", pairs, render=False)
            assert "packed16 x4" in al.last_config, (l1, l2, sc, al.last_config)
            for k, (a, b) in enumerate(pairs):
                r = O.align(O.OVERLAP, a, b, *sc)
                assert (int(res["score"][k]), int(res["end_i"][k]), int(res["end_j"][k]), int(res["state"][k]), res["ops"][k]) == \
                       (r["score"], r["end_i"], r["end_j"], r["state"], r["ops"]), (l1, l2, sc, k)
            res = al.align_batch("overlap", pairs, traceback=False)
            assert "int32" in al.last_config
    al.set_scoring(3, -4, -9, -1)
    pairs = [("".join(rng.choice("ACGT") for _ in range(1000)),) * 2 for _ in range(4)]
    res = al.align_batch("overlap", pairs, render=False)           # 4 * (9 * 1000 + 3 * 1000) > 2^15: int32
    assert "int32" in al.last_config
    for k, (a, b) in enumerate(pairs):
        r = O.align(O.OVERLAP, a, b, 3, -4, -9, -1)
        assert (int(res["score"][k]), res["ops"][k]) == (r["score"], r["ops"])


def test_deep_lane_kernels_overlap_scores_and_edit(al):
    """Overlap without tracebacks and edit run with 8 or 16 rows per lane once the first sequence is longer than 256:
    one strip (257..1024 rows), several strips (> 1024), ragged batches, row l1 anywhere inside its lane, both alphabets."""
    rng = random.Random(21)
    dna = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    prot = lambda n: "".join(rng.choice("ACDEFGHIKL") for _ in range(n))
    def related(a, gen):
        t = list(a)
        for _ in range(max(1, len(t) // 25)):
            q = rng.randrange(len(t))
            r = rng.random()
            if r < 0.4:
                t[q] = gen(1)
            elif r < 0.7:
                del t[q]
            else:
                t.insert(q, gen(1))
        return "".join(t)
    batches = []
    for gen in (dna, prot):
        lens = [257, 300, 511, 512, 513, 640, 999, 1000, 1001, 1023, 1024, 1025, 1500, 2049]
        pairs = []
        for n in lens:
            a = gen(n)
            pairs.append((a, gen(rng.randint(200, 1100))))
            pairs.append((a, related(a[n // 3:], gen) + gen(50)))      # a suffix of s1 opens s2: a real overlap
        batches.append(pairs)
        batches.append([(gen(1000), gen(1000)) for _ in range(9)])     # uniform 1 kbp reads (the C5 shape)
    for pairs in batches:
        for mode, sc in (("overlap", (1, -2, -5, -1)), ("overlap", (2, -3, -4, -1)), ("edit", (1, 1, -5, -1)), ("edit", (1, -2, -5, -1))):
            al.set_scoring(*sc)
            res = al.align_batch(mode, pairs, traceback=False)
            if mode == "edit" and sc[1] == 1 and all(set(a + b) <= set("ACGT") for a, b in pairs):
                assert "myers" in al.last_config, al.last_config          # unit costs on DNA: the bit-parallel kernel
            else:
                assert "rows/lane=8" in al.last_config or "rows/lane=16" in al.last_config, al.last_config
            for k, (a, b) in enumerate(pairs):
                r = O.align(O.MODE_NAMES[mode], a, b, *sc)
                assert int(res["score"][k]) == r["score"], (mode, sc, k, len(a), len(b))
                if mode == "overlap":
                    assert (int(res["end_i"][k]), int(res["end_j"][k])) == (r["end_i"], r["end_j"]), (mode, sc, k, len(a), len(b))


@pytest.mark.parametrize("cfg", ["C3", "C4", "C5"])
def test_baseline_shapes_at_scale(al, cfg):
    """BASELINE configs[2..4] at sizes that fill the chip for several rounds of the work queue (pointer slots reused,
    launches of the host entry in flight side by side): a seeded sample of every batch equals the oracle bit for bit --
    half of the pairs unrelated, half related (long tracebacks with gaps)."""
    from aligntools.c_amd.synth import synth_pairs_blob, mutate_pairs
    mode, l1, l2, n, sc, uj, sites, nsample = {
        "C3": ("global", 1024, 1024, 10000, (1, -1, -4, -1, -10), False, [], 200),     # the config's own 10 000 pairs
        "C4": ("fit", 150, 500, 40000, (2, -2, -5, -1, -10), True, [100, 200, 300, 400], 150),
        "C5": ("overlap", 1000, 1000, 3000, (1, -2, -5, -1, -10), False, [], 200),
    }[cfg]
    blob = synth_pairs_blob(0x5EED0100 + len(cfg) + ord(cfg[1]), n, l1, l2)
    rel = mutate_pairs(blob, l1, l2, 7, sub=0.06)
    rng = random.Random(ord(cfg[1]))
    pairs = []
    for k in range(n):
        row = rel[k] if k % 2 else blob[k]
        s1, s2 = row[:l1].tobytes(), row[l1:].tobytes()
        if k % 4 == 1:                       # an indel as well, so that the related pairs carry gaps
            q = rng.randrange(10, l1 - 10)
            s1 = s1[:q] + s1[q + 3:] + b"ACG"
        pairs.append((s1, s2))
    al.set_scoring(*sc, uj, sites)
    res = al.align_batch(mode, pairs, render=False)
    assert (res["nops"] >= 0).all()
    for k in rng.sample(range(n), nsample):
        r = O.align(O.MODE_NAMES[mode], pairs[k][0], pairs[k][1], *sc, uj, sites)
        assert (int(res["score"][k]), int(res["end_i"][k]), int(res["end_j"][k]), int(res["state"][k]), res["ops"][k]) == \
               (r["score"], r["end_i"], r["end_j"], r["state"], r["ops"]), (cfg, k)
    # size-independent properties of EVERY pair: the ops re-score to the reported score, start inside the matrix at the
    # mode's end cell and end where the mode's traceback must (global: the origin; fit: row 0; overlap: column 0)
    m_, u_, o_, e_, g_ = sc
    for k in range(n):
        ei, ej = int(res["end_i"][k]), int(res["end_j"][k])
        assert ei == l1 and (ej == l2 if mode == "global" else 0 <= ej < l2), (cfg, k, ei, ej)
        got, i0, j0 = _rescore_np(res["ops"][k], pairs[k][0], pairs[k][1], ei, ej, m_, u_, o_, e_, g_, mode)
        assert got == int(res["score"][k]), (cfg, k, got, int(res["score"][k]))
        assert (i0, j0) == (0, 0) if mode == "global" else (i0 == 0 if mode == "fit" else j0 == 0), (cfg, k, i0, j0)
    # the same batch under a tie-heavy scoring (1 / -1 / -1 / -1: most cells have several equal candidates), where a wrong first-wins
    # order gives another path of the SAME score -- invisible to the re-scoring above, visible to the oracle
    tie = (1, -1, -1, -1, -2)
    al.set_scoring(*tie, uj, sites)
    res = al.align_batch(mode, pairs, render=False)
    assert (res["nops"] >= 0).all()
    for k in rng.sample(range(n), max(nsample, 200)):
        r = O.align(O.MODE_NAMES[mode], pairs[k][0], pairs[k][1], *tie, uj, sites)
        assert (int(res["score"][k]), int(res["end_i"][k]), int(res["end_j"][k]), int(res["state"][k]), res["ops"][k]) == \
               (r["score"], r["end_i"], r["end_j"], r["state"], r["ops"]), (cfg, "tie-heavy", k)


def test_device_entry_refuses_broken_uniform_promise(al):
    """at_align_batch_device(uniform_shape = 1) with a pair of another length: the packed kernel must not sweep it with
    the batch extents; the pairs of that work item come back as domain errors, every other pair is unaffected."""
    import torch
    import aligntools.c_amd as A
    rng = random.Random(5)
    n, l1, l2 = 64, 150, 150
    pairs = [("".join(rng.choice("ACGT") for _ in range(l1)).encode(), "".join(rng.choice("ACGT") for _ in range(l2)).encode()) for _ in range(n)]
    dev = torch.device("cuda", 0)
    words, woff1, woff2, len1, len2, bits = A.pack_pairs(pairs)
    len1 = len1.copy()
    len1[37] = 149
    t = lambda a: torch.from_numpy(a).to(dev)
    d_words, d_woff1, d_woff2, d_len1, d_len2 = t(words.view(np.int32)), t(woff1), t(woff2), t(len1), t(len2)
    d_ops_off = t(np.arange(n, dtype=np.int64) * (l1 + l2))
    d_res = torch.zeros((4, n), dtype=torch.int32, device=dev)
    d_nops = torch.zeros(n, dtype=torch.int32, device=dev)
    d_ops = torch.zeros(n * (l1 + l2) + 64, dtype=torch.uint8, device=dev)
    al.set_scoring(2, -2, -5, -2)
    al.align_batch_device(A.MODE_LOCAL, n, d_words.data_ptr(), bits, d_woff1.data_ptr(), d_len1.data_ptr(), d_woff2.data_ptr(),
                          d_len2.data_ptr(), l1, l2, True, True, d_res[0].data_ptr(), d_res[1].data_ptr(), d_res[2].data_ptr(),
                          d_res[3].data_ptr(), d_ops.data_ptr(), d_ops_off.data_ptr(), d_nops.data_ptr(),
                          torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert "packed16" in al.last_config
    score, nops = d_res[0].cpu().numpy(), d_nops.cpu().numpy()
    refused = np.flatnonzero(score == np.iinfo(np.int32).min)
    assert 37 in refused and len(refused) <= 16 and (nops[refused] == -1).all()   # a work item: up to 16 pairs
    for k in range(n):
        if k not in refused:
            assert int(score[k]) == O.align(O.LOCAL, pairs[k][0], pairs[k][1], 2, -2, -5, -2)["score"]
    # ragged entry with bounds below the real lengths: the int32 kernel refuses the pairs instead of overrunning the
    # regions sized from the bounds (pair 37, 149 long, fits max_len1 = 149)
    d_res.zero_()
    al.align_batch_device(A.MODE_LOCAL, n, d_words.data_ptr(), bits, d_woff1.data_ptr(), d_len1.data_ptr(), d_woff2.data_ptr(),
                          d_len2.data_ptr(), 149, l2, False, True, d_res[0].data_ptr(), d_res[1].data_ptr(), d_res[2].data_ptr(),
                          d_res[3].data_ptr(), d_ops.data_ptr(), d_ops_off.data_ptr(), d_nops.data_ptr(),
                          torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    score, nops = d_res[0].cpu().numpy(), d_nops.cpu().numpy()
    assert "int32" in al.last_config
    assert [k for k in range(n) if score[k] != np.iinfo(np.int32).min] == [37] and nops[37] >= 0 and (np.delete(nops, 37) == -1).all()
    assert int(score[37]) == O.align(O.LOCAL, pairs[37][0][:149], pairs[37][1], 2, -2, -5, -2)["score"]


def test_packed_kernels_on_byte_alphabets(al):
    """Uniform batches whose sequences are not pure ACGT (reads with N, lower case, protein) run on the packed kernels with
    byte sequence words (compare instead of the 2-bit score LUT): every mode, with tracebacks, against the oracle."""
    rng = random.Random(123)
    for alpha, l1, l2 in (("ACGTN", 150, 150), ("ACDEFGHIKLMNPQRSTVWY", 90, 120), ("acgtACGT", 64, 200), ("ACGTN", 300, 340)):
        pairs = []
        for k in range(120):
            a = "".join(rng.choice(alpha) for _ in range(l1))
            if k % 2:
                t = list(a)
                for _ in range(l1 // 20):
                    q = rng.randrange(len(t))
                    r = rng.random()
                    if r < 0.5:
                        t[q] = rng.choice(alpha)
                    elif r < 0.75:
                        del t[q]
                    else:
                        t.insert(q, rng.choice(alpha))
                b = ("".join(rng.choice(alpha) for _ in range(20)) + "".join(t) + "".join(rng.choice(alpha) for _ in range(l2)))[:l2]
            else:
                b = "".join(rng.choice(alpha) for _ in range(l2))
            pairs.append((a, b))
        for mode, sc, uj, sites in (("local", (2, -2, -5, -2, -10), False, []), ("global", (1, -1, -4, -1, -10), False, []),
                                    ("fit", (2, -2, -5, -1, -10), False, []), ("fit", (2, -2, -5, -1, -8), True, [10, 50, 100])):
            al.set_scoring(*sc, uj, sites)
            res = al.align_batch(mode, pairs, render=False)
            assert "packed16" in al.last_config and "bits=8" in al.last_config, al.last_config
            for k, (x, y) in enumerate(pairs):
                r = O.align(O.MODE_NAMES[mode], x, y, *sc, uj, sites)
                assert (int(res["score"][k]), int(res["end_i"][k]), int(res["end_j"][k]), int(res["state"][k]), res["ops"][k]) == \
                       (r["score"], r["end_i"], r["end_j"], r["state"], r["ops"]), (alpha, mode, uj, k)


@pytest.mark.parametrize("maxl", [304, 208, 152])
def test_ragged_local_batches_in_frames(al, maxl):
    """Ragged local batches of reads (l1 <= 304) run on the packed kernel in frames: sorted into buckets of similar size,
    every alignment keeping its own extents inside its bucket's frame.  All lengths from 1 up, unrelated and related
    pairs, score / end cell / ops against the oracle; a batch with one longer read falls back to the int32 kernel."""
    rng = random.Random(91)
    dna = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    pairs = []
    for k in range(3000):
        l1 = rng.choice([1, 2, 15, 16, 17, 40, 41, 48, 49, 56, 57, 63, 64, 65, 80, 81, 96, 104, 105, 112, 113, 128, 129, 150, 152, 160, 161, 207, 208,
                         209, 250, 256, 257, 300, 304]) if k % 3 == 0 else rng.randint(1, 304 if k % 2 else 208)
        l1 = min(l1, maxl)
        l2 = rng.randint(1, 260)
        a = dna(l1)
        if k % 2:
            t = list(a)
            for _ in range(max(1, l1 // 15)):
                q = rng.randrange(len(t))
                r = rng.random()
                if r < 0.4:
                    t[q] = rng.choice("ACGT")
                elif r < 0.7 and len(t) > 1:
                    del t[q]
                else:
                    t.insert(q, rng.choice("ACGT"))
            b = (dna(rng.randint(0, 30)) + "".join(t) + dna(l2))[:max(l2, 1)]
        else:
            b = dna(l2)
        pairs.append((a, b))
    for sc in ((2, -2, -5, -2), (1, -1, -1, -1), (3, -1, 0, 0)):
        al.set_scoring(*sc)
        for tb in (True, False):
            res = al.align_batch("local", pairs, traceback=tb, render=False)
            assert "ragged frames" in al.last_config and "4x16-lane" in al.last_config, al.last_config
            for k, (a, b) in enumerate(pairs):
                r = O.align(O.LOCAL, a, b, *sc)
                assert (int(res["score"][k]), int(res["end_i"][k]), int(res["end_j"][k])) == (r["score"], r["end_i"], r["end_j"]), (sc, tb, k)
                if tb:
                    assert res["ops"][k] == r["ops"], (sc, k)
    al.set_scoring(2, -2, -5, -2)
    longer = pairs[:200] + [(dna(609), dna(100))]          # (305 .. 608 bases: the 32-lane frames, test_ragged_long_reads_in_32_lane_frames)
    res = al.align_batch("local", longer, render=False)
    assert "int32" in al.last_config
    r = O.align(O.LOCAL, longer[-1][0], longer[-1][1], 2, -2, -5, -2)
    assert int(res["score"][-1]) == r["score"] and res["ops"][-1] == r["ops"]


@pytest.mark.parametrize("mode", ["global", "fit", "fitj"])
def test_ragged_global_and_fit_batches_in_frames(al, mode):
    """Ragged global / fit / fit -s batches of reads (l1 <= 304) run on the packed kernels in frames whose work items hold
    reads of one length: 8-lane groups up to 152 bases, 16-lane groups beyond; l2 differs inside an item.  Length mixes
    around every rows-per-lane class edge, a few lengths that occur once (their items are padded with repeats), related and
    unrelated pairs, both alphabets, with and without tracebacks -- score / end cell / start state / ops against the
    oracle; a batch with one read of 305 bases falls back to the int32 kernel."""
    rng = random.Random(4471)
    uj = mode == "fitj"
    m = "fit" if uj else mode
    for alpha, lens in (("ACGT", [1, 2, 39, 40, 41, 48, 49, 56, 57, 64, 65, 80, 81, 104, 105, 128, 129, 150, 151, 152, 153, 160, 161, 200, 208, 209, 250, 256, 257, 300, 304]),
                        ("ACGTN", [30, 100, 150, 180, 250, 290])):
        dna = lambda n: "".join(rng.choice(alpha) for _ in range(n))
        pairs = []
        for k in range(1400):
            l1 = rng.choice(lens) if k % 4 else rng.randint(1, 304)
            l2 = rng.randint(max(l1, 2), max(l1, 2) + rng.choice([0, 5, 60, 300]))
            a = dna(l1)
            if k % 2:
                t = list(a)
                for _ in range(max(1, l1 // 15)):
                    q = rng.randrange(len(t))
                    r = rng.random()
                    if r < 0.4:
                        t[q] = rng.choice(alpha)
                    elif r < 0.7 and len(t) > 1:
                        del t[q]
                    else:
                        t.insert(q, rng.choice(alpha))
                b = (dna(rng.randint(0, 30)) + "".join(t) + dna(l2))[:l2]
            else:
                b = dna(l2)
            pairs.append((a, b))
        sites = [20, 100, 250, 400]
        for sc in ((2, -2, -5, -1, -10), (1, -1, -1, -1, -3)):
            al.set_scoring(*sc, uj, sites)
            for tb in (True, False):
                res = al.align_batch(m, pairs, traceback=tb, render=False)
                assert "equal-l1 work items" in al.last_config, al.last_config
                for k, (a, b) in enumerate(pairs):
                    r = O.align(O.MODE_NAMES[m], a, b, *sc, uj, sites)
                    assert (int(res["score"][k]), int(res["end_i"][k]), int(res["end_j"][k]), int(res["state"][k])) == \
                           (r["score"], r["end_i"], r["end_j"], r["state"]), (mode, alpha, sc, tb, k, len(a), len(b))
                    if tb:
                        assert res["ops"][k] == r["ops"], (mode, alpha, sc, k, len(a), len(b))
    al.set_scoring(2, -2, -5, -1, -10, uj, sites)
    longer = pairs[:200] + [(dna(513), dna(600))]          # (beyond the longest one-strip frame of global, 512, and of fit, 416)
    res = al.align_batch(m, longer, render=False)
    assert "int32" in al.last_config
    r = O.align(O.MODE_NAMES[m], longer[-1][0], longer[-1][1], 2, -2, -5, -1, -10, uj, sites)
    assert int(res["score"][-1]) == r["score"] and res["ops"][-1] == r["ops"]


@pytest.mark.parametrize("mode", ["local", "global", "fit", "fitj"])
def test_ragged_long_reads_in_32_lane_frames(al, mode):
    """Ragged batches with reads of 305 .. 608 bases (local), .. 512 (global), .. 416 (fit, fit -s) stay on the packed kernels: frames
    on two groups of 32 lanes, one strip of 32 x 10 / 12 / 13 / 16 / 19 rows, next to the 8- and 16-lane frames of the shorter reads
    of the same batch.  Lengths on both sides of every class edge, lengths that occur once (padded items), related and unrelated
    pairs, both alphabets, with and without tracebacks: score / end cell / start state / ops against the oracle."""
    rng = random.Random(6080 + len(mode))
    uj = mode == "fitj"
    m = "fit" if uj else mode
    top = {"local": 608, "global": 512, "fit": 416, "fitj": 416}[mode]
    edges = [x for x in (150, 300, 304, 305, 306, 319, 320, 321, 383, 384, 385, 415, 416, 417, 500, 511, 512, 513, 600, 607, 608) if x <= top]
    for alpha in ("ACGT", "ACGTN"):
        dna = lambda n: "".join(rng.choice(alpha) for _ in range(n))
        pairs = []
        for k in range(260 if alpha == "ACGT" else 120):
            l1 = rng.choice(edges) if k % 3 else rng.randint(250, top)
            l2 = rng.randint(max(l1, 2), max(l1, 2) + rng.choice([0, 7, 90])) if m == "fit" else rng.randint(max(1, l1 - 120), l1 + 120)
            a = dna(l1)
            if k % 2:
                t = list(a)
                for _ in range(max(1, l1 // 20)):
                    q = rng.randrange(len(t))
                    r = rng.random()
                    if r < 0.4:
                        t[q] = rng.choice(alpha)
                    elif r < 0.7 and len(t) > 1:
                        del t[q]
                    else:
                        t.insert(q, rng.choice(alpha))
                b = (dna(rng.randint(0, 20)) + "".join(t) + dna(l2))[:l2]
            else:
                b = dna(l2)
            pairs.append((a, b))
        sites = [20, 100, 250, 400]
        # (global and fit matrices drift downwards: at 500 bases only mild scorings keep 16 x the score range inside 16 bits, packed_ok)
        sc = (2, -2, -5, -1, -10) if mode == "local" and alpha == "ACGTN" else (1, -1, -2, -1, -4)
        al.set_scoring(*sc, uj, sites)
        for tb in (True, False):
            res = al.align_batch(m, pairs, traceback=tb, render=False)
            assert "ragged frames" in al.last_config, al.last_config
            for k, (a, b) in enumerate(pairs):
                r = O.align(O.MODE_NAMES[m], a, b, *sc, uj, sites)
                assert (int(res["score"][k]), int(res["end_i"][k]), int(res["end_j"][k]), int(res["state"][k])) == \
                       (r["score"], r["end_i"], r["end_j"], r["state"]), (mode, alpha, tb, k, len(a), len(b))
                if tb:
                    assert res["ops"][k] == r["ops"], (mode, alpha, k, len(a), len(b))
        # every pair long: the launch itself names the 32-lane groups
        longs = [p for p in pairs if len(p[0]) > 304]
        res = al.align_batch(m, longs, render=False)
        assert "2x32-lane groups" in al.last_config and "ragged frames" in al.last_config, al.last_config
        for k, (a, b) in enumerate(longs[:40]):
            r = O.align(O.MODE_NAMES[m], a, b, *sc, uj, sites)
            assert (int(res["score"][k]), res["ops"][k]) == (r["score"], r["ops"]), (mode, alpha, k)


def test_ragged_overlap_with_tracebacks_on_the_packed_kernel(al):
    """Ragged overlap batches with tracebacks (reads of up to 1 024 bases, scores within 16 bits) run on the packed overlap kernel in
    frames: the two alignments of a wavefront share l1 (the end cells lie in row l1), 4 rows per lane up to 256 bases and 16 beyond,
    each alignment scanning row l1 up to its own column l2 - 1.  True overlaps with errors, unrelated reads and empty results, both
    alphabets, against the oracle; without tracebacks, or with a read of 1 025 bases, the batch runs on the int32 kernel."""
    rng = random.Random(9264)
    for alpha in ("ACGT", "ACGTN"):
        dna = lambda n: "".join(rng.choice(alpha) for _ in range(n))
        pairs = []
        for k in range(300 if alpha == "ACGT" else 120):
            l1 = rng.choice([1, 2, 60, 255, 256, 257, 400, 800, 1000, 1023, 1024]) if k % 3 == 0 else rng.randint(1, 1024)
            l2 = rng.randint(1, 1024)
            a = dna(l1)
            if k % 4:
                ov = rng.randint(1, min(l1, l2))
                t = list(a[l1 - ov:])
                for _ in range(ov // 25):
                    q = rng.randrange(len(t))
                    x = rng.random()
                    if x < 0.5:
                        t[q] = rng.choice(alpha)
                    elif x < 0.75 and len(t) > 2:
                        del t[q]
                    else:
                        t.insert(q, rng.choice(alpha))
                b = ("".join(t) + dna(l2))[:l2]
            else:
                b = dna(l2)
            pairs.append((a, b))
        for sc in ((1, -2, -5, -1), (2, -1, -1, -1)):
            al.set_scoring(*sc)
            res = al.align_batch("overlap", pairs, render=False)
            assert "packed16 x4" in al.last_config and "ragged frames" in al.last_config, al.last_config
            for k, (a, b) in enumerate(pairs):
                r = O.align(O.OVERLAP, a, b, *sc)
                assert (int(res["score"][k]), int(res["end_i"][k]), int(res["end_j"][k]), res["ops"][k]) == \
                       (r["score"], r["end_i"], r["end_j"], r["ops"]), (alpha, sc, k, len(a), len(b))
        res = al.align_batch("overlap", pairs, traceback=False)
        assert "int32" in al.last_config
        res = al.align_batch("overlap", pairs[:100] + [(dna(1025), dna(30))], render=False)
        assert "int32" in al.last_config
        r = O.align(O.OVERLAP, pairs[7][0], pairs[7][1], 2, -1, -1, -1)
        assert int(res["score"][7]) == r["score"] and res["ops"][7] == r["ops"]


def test_chunked_host_entry(al):
    """Batches of >= 32k pairs go through the host entry as chunks on helper handles and threads: the results equal
    the one-piece run (AT_HOST_CHUNKS=1), ragged shapes included, and an error names the pair by its batch index."""
    import os
    import aligntools.c_amd as A
    rng = random.Random(33)
    n = 40000
    pairs = [("".join(rng.choice("ACGT") for _ in range(rng.randint(20, 60))), "".join(rng.choice("ACGT") for _ in range(rng.randint(60, 90))))
             for _ in range(n)]
    al.set_scoring(2, -2, -5, -2)
    for mode in ("local", "fit"):
        chunked = al.align_batch(mode, pairs, render=False)
        assert "chunks" in al.last_config
        os.environ["AT_HOST_CHUNKS"] = "1"
        try:
            whole = al.align_batch(mode, pairs, render=False)
            assert "chunks" not in al.last_config
        finally:
            del os.environ["AT_HOST_CHUNKS"]
        for key in ("score", "end_i", "end_j", "state", "nops"):
            assert (chunked[key] == whole[key]).all(), (mode, key)
        assert chunked["ops"] == whole["ops"]
        gs = al.align_batch_strings(mode, pairs[:33000])
        for k in rng.sample(range(33000), 200):
            r = O.align(O.MODE_NAMES[mode], pairs[k][0], pairs[k][1], 2, -2, -5, -2)
            assert (int(gs["score"][k]), gs["r1"][k], gs["r2"][k]) == (r["score"], r["r1"], r["r2"]), (mode, k)
    bad = list(pairs)
    bad[39999] = ("", "ACGT")
    with pytest.raises(A.AlignToolsError) as ei:
        al.align_batch("local", bad)
    assert ei.value.code == -4 and "39999" in str(ei.value)


def test_all_vs_all_mode(al):
    """BASELINE config 'overlap all-vs-all': ordered pairs (a < b) enumerated on the GPU from a linear triangle
    index (no per-pair descriptors), split in two ranges like two ranks would; every pair equals the oracle."""
    import numpy as np
    import torch
    import aligntools.c_amd as A
    rng = random.Random(31)
    reads = []
    base = "".join(rng.choice("ACGT") for _ in range(400))
    for k in range(41):
        st = rng.randint(0, 200)
        reads.append(base[st:st + rng.randint(80, 200)] + "".join(rng.choice("ACGT") for _ in range(rng.randint(0, 30))))
    n = len(reads)
    words, woff, _w2, lens, _l2, bits = A.pack_pairs([(r.encode(), b"") for r in reads])
    dev = torch.device("cuda", 0)
    d_words = torch.from_numpy(words.view(np.int32)).to(dev)
    d_woff = torch.from_numpy(woff).to(dev)
    d_len = torch.from_numpy(lens).to(dev)
    total = n * (n - 1) // 2
    maxl = int(lens.max())
    for mode in ("overlap", "local"):
        al.set_scoring(1, -2, -5, -1)
        got = {}
        for first, cnt in ((0, 300), (300, total - 300)):
            res = torch.zeros((5, cnt), dtype=torch.int32, device=dev)
            ops = torch.zeros(cnt * 2 * maxl + 64, dtype=torch.uint8, device=dev)
            ops_off = torch.arange(cnt, dtype=torch.int64, device=dev) * (2 * maxl)
            al.align_allpairs_device(A.MODES[mode], n, d_words.data_ptr(), bits, d_woff.data_ptr(), d_len.data_ptr(), maxl,
                                     first, cnt, True, res[0].data_ptr(), res[1].data_ptr(), res[2].data_ptr(), res[3].data_ptr(),
                                     ops.data_ptr(), ops_off.data_ptr(), res[4].data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            r, o = res.cpu().numpy(), ops.cpu().numpy()
            for p in range(cnt):
                got[first + p] = (int(r[0, p]), int(r[1, p]), int(r[2, p]), bytes(o[p * 2 * maxl: p * 2 * maxl + int(r[4, p])]))
        t = 0
        for a in range(n):
            for b in range(a + 1, n):
                ref = O.align(O.MODE_NAMES[mode], reads[a], reads[b], 1, -2, -5, -1)
                assert got[t] == (ref["score"], ref["end_i"], ref["end_j"], ref["ops"]), (mode, a, b)
                t += 1
        assert t == total
    # edit distance with unit costs over the same triangle: the bit-parallel kernel enumerates the pairs itself (one alignment per lane)
    al.set_scoring(1, 1, -5, -1)
    res = torch.zeros((5, total), dtype=torch.int32, device=dev)
    al.align_allpairs_device(A.MODES["edit"], n, d_words.data_ptr(), bits, d_woff.data_ptr(), d_len.data_ptr(), maxl,
                             0, total, False, res[0].data_ptr(), res[1].data_ptr(), res[2].data_ptr(), res[3].data_ptr(),
                             0, 0, res[4].data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert "myers" in al.last_config and "64x1-lane" in al.last_config, al.last_config
    r = res.cpu().numpy()
    t = 0
    for a in range(n):
        for b in range(a + 1, n):
            assert (int(r[0, t]), int(r[1, t]), int(r[2, t])) == (O.align(O.EDIT, reads[a], reads[b], 1, 1, -5, -1)["score"], len(reads[a]), len(reads[b])), (a, b)
            t += 1


def test_gpu_rendering_device_entry_full_size(al):
    """at_render_batch_device on the full C2 batch, buffers resident in HBM (the bench.py arrangement): every string
    equals what at_render makes of the same ops on the host, and r1 / r2 without their gaps are the aligned
    substrings of s1 / s2."""
    import torch
    import aligntools.c_amd as A
    from aligntools.c_amd.synth import synth_pairs_blob
    n, l1, l2 = 100000, 150, 150
    blob = synth_pairs_blob(0x5EED0002, n, l1, l2)
    # every 5th pair: s2 = mutated s1, so that long alignments with gaps are in the batch
    rng = random.Random(11)
    pairs = []
    for k, row in enumerate(blob):
        s1 = row[:l1].tobytes()
        s2 = row[l1:].tobytes()
        if k % 5 == 0:
            t = bytearray(s1)
            for _ in range(6):
                q = rng.randrange(len(t))
                r = rng.random()
                if r < 0.4:
                    t[q] = rng.choice(b"ACGT")
                elif r < 0.7:
                    del t[q]
                else:
                    t.insert(q, rng.choice(b"ACGT"))
            s2 = (bytes(t) + s2)[:l2]
        pairs.append((s1, s2))
    dev = torch.device("cuda", 0)
    words, woff1, woff2, len1, len2, bits = A.pack_pairs(pairs)
    assert bits == 2
    t = lambda a: torch.from_numpy(a).to(dev)
    d_words, d_woff1, d_woff2, d_len1, d_len2 = t(words.view(np.int32)), t(woff1), t(woff2), t(len1), t(len2)
    d_ops_off = t(np.arange(n, dtype=np.int64) * (l1 + l2))
    d_str_off = t(np.arange(n, dtype=np.int64) * (l1 + l2 + 1))
    d_res = torch.zeros((4, n), dtype=torch.int32, device=dev)
    d_nops = torch.zeros(n, dtype=torch.int32, device=dev)
    d_ops = torch.zeros(n * (l1 + l2) + 64, dtype=torch.uint8, device=dev)
    d_r1 = torch.full((n * (l1 + l2 + 1) + 64,), 0x7f, dtype=torch.uint8, device=dev)
    d_r2 = torch.full((n * (l1 + l2 + 1) + 64,), 0x7f, dtype=torch.uint8, device=dev)
    al.set_scoring(2, -2, -5, -2)
    stream = torch.cuda.current_stream().cuda_stream
    al.align_batch_device(A.MODE_LOCAL, n, d_words.data_ptr(), bits, d_woff1.data_ptr(), d_len1.data_ptr(), d_woff2.data_ptr(),
                          d_len2.data_ptr(), l1, l2, True, True, d_res[0].data_ptr(), d_res[1].data_ptr(), d_res[2].data_ptr(),
                          d_res[3].data_ptr(), d_ops.data_ptr(), d_ops_off.data_ptr(), d_nops.data_ptr(), stream)
    al.render_batch_device(n, d_words.data_ptr(), bits, d_woff1.data_ptr(), d_woff2.data_ptr(), d_res[1].data_ptr(),
                           d_res[2].data_ptr(), d_ops.data_ptr(), d_ops_off.data_ptr(), d_nops.data_ptr(),
                           d_r1.data_ptr(), d_r2.data_ptr(), d_str_off.data_ptr(), True, stream)
    cap = n * 80
    d_packed = torch.full((cap,), 0x7f, dtype=torch.uint8, device=dev)
    d_poff = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    al.compact_ops_device(n, d_ops.data_ptr(), d_ops_off.data_ptr(), d_nops.data_ptr(), d_packed.data_ptr(), cap,
                          d_poff.data_ptr(), stream)
    torch.cuda.synchronize()
    res = d_res.cpu().numpy()
    nops = d_nops.cpu().numpy()
    ops = d_ops.cpu().numpy()
    # CIGAR compaction: offsets are the exclusive scan of nops, the payload is the slots' used parts back to back
    poff = d_poff.cpu().numpy()
    want = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(nops, out=want[1:])
    assert (poff == want).all() and want[-1] <= cap
    slots = ops[:n * (l1 + l2)].reshape(n, l1 + l2)
    mask = np.arange(l1 + l2)[None, :] < nops[:, None]
    assert (d_packed.cpu().numpy()[:want[-1]] == slots[mask]).all()
    r1 = d_r1.cpu().numpy().tobytes()
    r2 = d_r2.cpu().numpy().tobytes()
    assert nops.max() > 100            # the related pairs align end to end
    for k in range(n):
        so, nk = k * (l1 + l2 + 1), int(nops[k])
        a, b = r1[so:so + nk], r2[so:so + nk]
        assert r1[so + nk] == 0 and r2[so + nk] == 0
        ei, ej = int(res[1][k]), int(res[2][k])
        ga, gb = a.replace(b"-", b""), b.replace(b"-", b"")
        assert pairs[k][0][ei - len(ga):ei] == ga and pairs[k][1][ej - len(gb):ej] == gb, k
        if k % 16 == 0:
            ha, hb = al.render(ops[k * (l1 + l2):k * (l1 + l2) + nk].tobytes(), pairs[k][0], ei, pairs[k][1], ej)
            assert ha.encode() == a and hb.encode() == b, k
    for k in rng.sample(range(n), 200):
        r = O.align(O.LOCAL, pairs[k][0], pairs[k][1], 2, -2, -5, -2)
        so, nk = k * (l1 + l2 + 1), int(nops[k])
        assert (r["r1"], r["r2"]) == (r1[so:so + nk].decode(), r2[so:so + nk].decode())
    # d_str_off = NULL: the strings go to the ops offsets, without terminators (what bench.py --render does)
    d_r1.fill_(0x7f)
    d_r2.fill_(0x7f)
    al.render_batch_device(n, d_words.data_ptr(), bits, d_woff1.data_ptr(), d_woff2.data_ptr(), d_res[1].data_ptr(),
                           d_res[2].data_ptr(), d_ops.data_ptr(), d_ops_off.data_ptr(), d_nops.data_ptr(),
                           d_r1.data_ptr(), d_r2.data_ptr(), None, False, stream)
    torch.cuda.synchronize()
    q1 = d_r1.cpu().numpy().tobytes()
    q2 = d_r2.cpu().numpy().tobytes()
    for k in range(0, n, 3):
        so, oo, nk = k * (l1 + l2 + 1), k * (l1 + l2), int(nops[k])
        assert q1[oo:oo + nk] == r1[so:so + nk] and q2[oo:oo + nk] == r2[so:so + nk]
        assert nk == l1 + l2 or q1[oo + nk] == 0x7f           # nothing written behind the string


def test_all_vs_all_scores_long_reads(al):
    """The C5 arrangement proper: all-vs-all overlap SCORES over reads of about 1 kbp (16 rows per lane, one strip, no
    pointer matrix), pairs enumerated on the GPU; score and end cell of every pair equal the oracle's."""
    import torch
    import aligntools.c_amd as A
    rng = random.Random(32)
    genome = "".join(rng.choice("ACGT") for _ in range(6000))
    reads = []
    for k in range(24):
        st = rng.randint(0, 5000)
        r = list(genome[st:st + rng.randint(700, 1000)])
        for _ in range(len(r) // 40):                       # sequencing errors
            q = rng.randrange(len(r))
            x = rng.random()
            if x < 0.5:
                r[q] = rng.choice("ACGT")
            elif x < 0.75:
                del r[q]
            else:
                r.insert(q, rng.choice("ACGT"))
        reads.append("".join(r))
    n = len(reads)
    words, woff, _w2, lens, _l2, bits = A.pack_pairs([(r.encode(), b"") for r in reads])
    dev = torch.device("cuda", 0)
    d_words, d_woff, d_len = (torch.from_numpy(x).to(dev) for x in (words.view(np.int32), woff, lens))
    total = n * (n - 1) // 2
    res = torch.zeros((4, total), dtype=torch.int32, device=dev)
    al.set_scoring(1, -2, -5, -1)
    al.align_allpairs_device(A.MODE_OVERLAP, n, d_words.data_ptr(), bits, d_woff.data_ptr(), d_len.data_ptr(), int(lens.max()),
                             0, total, False, res[0].data_ptr(), res[1].data_ptr(), res[2].data_ptr(), res[3].data_ptr(),
                             None, None, None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert "rows/lane=16" in al.last_config, al.last_config
    r = res.cpu().numpy()
    t = 0
    positive = 0
    for a in range(n):
        for b in range(a + 1, n):
            ref = O.align(O.OVERLAP, reads[a], reads[b], 1, -2, -5, -1)
            assert (int(r[0, t]), int(r[1, t]), int(r[2, t])) == (ref["score"], ref["end_i"], ref["end_j"]), (a, b)
            positive += ref["score"] > 50
            t += 1
    assert positive >= 5       # real overlaps are in the set


def test_all_vs_all_streamed_in_slices(al):
    """at_align_allpairs_stream: the triangle in slices with bounded memory.  The same pairs through slices of three different
    sizes (one of them not dividing anything, one larger than the range) and through the single-shot device entry give the same
    scores and end cells; a sub-range starts at the right pair; the callback sees every pair once, in order."""
    import torch
    import aligntools.c_amd as A
    rng = random.Random(77)
    n = 300
    reads = [bytes(rng.choice(b"ACGT") for _ in range(rng.randint(60, 140))) for _ in range(n)]
    lens = np.array([len(r) for r in reads], dtype=np.int32)
    off = np.zeros(n, dtype=np.int64)
    off[1:] = np.cumsum(lens[:-1])
    blob = np.frombuffer(b"".join(reads) + b"\0", dtype=np.uint8).copy()
    total = n * (n - 1) // 2
    al.set_scoring(1, -2, -5, -1)

    def run(first, npairs, chunk):
        got = np.full((4, npairs), -12345, dtype=np.int32)
        seen = []

        def on_slice(f, sc, ei, ej, st):
            seen.append((f, len(sc)))
            for row, x in enumerate((sc, ei, ej, st)):
                got[row, f - first:f - first + len(x)] = x
        al.align_allpairs_stream("overlap", blob, off, lens, first, npairs, chunk, on_slice)
        assert seen[0][0] == first and sum(x[1] for x in seen) == npairs
        assert all(seen[k][0] + seen[k][1] == seen[k + 1][0] for k in range(len(seen) - 1))
        return got, len(seen)
    full, ns = run(0, total, 7001)
    assert ns == (total + 7000) // 7001 and "slices" in al.last_config
    for chunk in (1 << 20, 4096):
        again, _ = run(0, total, chunk)
        assert (again == full).all()
    part, _ = run(12345, 20000, 3333)
    assert (part == full[:, 12345:32345]).all()
    # the single-shot device entry on the same reads
    words, woff, _w2, l1, _l2, bits = A.pack_pairs([(r, b"") for r in reads])
    dev = torch.device("cuda", 0)
    d_words, d_woff, d_len = (torch.from_numpy(x).to(dev) for x in (words.view(np.int32), woff, l1))
    res = torch.zeros((4, total), dtype=torch.int32, device=dev)
    al.align_allpairs_device(A.MODE_OVERLAP, n, d_words.data_ptr(), bits, d_woff.data_ptr(), d_len.data_ptr(), int(lens.max()), 0, total, False,
                             res[0].data_ptr(), res[1].data_ptr(), res[2].data_ptr(), res[3].data_ptr(), None, None, None,
                             torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert (res.cpu().numpy()[:3] == full[:3]).all()
    # and the oracle on a sample
    for q in [0, total - 1] + [rng.randrange(total) for _ in range(60)]:
        a = 0
        while (a + 1) * (2 * n - a - 2) // 2 <= q:
            a += 1
        b = q - a * (2 * n - a - 1) // 2 + a + 1
        ref = O.align(O.OVERLAP, reads[a], reads[b], 1, -2, -5, -1)
        assert (int(full[0, q]), int(full[1, q]), int(full[2, q])) == (ref["score"], ref["end_i"], ref["end_j"]), (q, a, b)
    # a callback that raises stops the sweep and surfaces in Python
    with pytest.raises(ZeroDivisionError):
        al.align_allpairs_stream("overlap", blob, off, lens, 0, total, 5000, lambda *a: 1 // 0)


def test_bit_parallel_edit_distance(al):
    """`edit -u 1` (unit mismatch cost: Levenshtein) runs on the bit-parallel kernel (at_myers.hip.h): every word / lane
    boundary of l1 (32, 1024, 2048, 4096, 8192, 16384 rows), ragged and uniform batches, unrelated and related pairs, empty sequences,
    against the oracle; any other -u keeps the cell-by-cell kernel."""
    rng = random.Random(404)
    dna = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    def related(a):
        t = list(a)
        for _ in range(max(1, len(t) // 10)):
            q = rng.randrange(len(t))
            r = rng.random()
            if r < 0.4:
                t[q] = rng.choice("ACGT")
            elif r < 0.7 and len(t) > 1:
                del t[q]
            else:
                t.insert(q, rng.choice("ACGT"))
        return "".join(t)
    lens = [1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 511, 512, 513, 1000, 1023, 1024, 1025, 1500, 2047, 2048, 2049, 3000, 4096, 4097, 5000, 8192, 8193, 12000, 16384, 16385, 20000]
    ragged = []
    for n in lens:
        a = dna(n)
        ragged.append((a, dna(rng.randint(1, min(2 * n + 5, 3000)))))
        ragged.append((a, related(a)))
        ragged.append((a, a))
    ragged += [("", "ACGT"), ("ACG", ""), ("", "")]
    al.set_scoring(1, 1, -5, -1)
    res = al.align_batch("edit", ragged)
    assert "myers" in al.last_config and "words/lane=32" in al.last_config, al.last_config   # (20 000 rows: 32 lanes x 32 words)
    for k, (a, b) in enumerate(ragged):
        assert int(res["score"][k]) == O.align(O.EDIT, a, b, 1, 1, -5, -1)["score"], (k, len(a), len(b))
    # reads of up to 256 bases: one alignment per LANE (64 per wavefront), 2 / 3 / 4 / 5 words for up to 64 / 96 / 128 / 160 bases, 8 beyond; a second sequence too
    # long for 64 windows in LDS keeps the eight-lane groups
    for l1, l2, want, groups in ((150, 150, 5, "64x1-lane"), (160, 90, 5, "64x1-lane"), (161, 300, 8, "64x1-lane"), (256, 256, 8, "64x1-lane"),
                                 (33, 2000, 2, "64x1-lane"), (64, 70, 2, "64x1-lane"), (65, 50, 3, "64x1-lane"), (96, 120, 3, "64x1-lane"), (97, 97, 4, "64x1-lane"),
                                 (128, 200, 4, "64x1-lane"), (129, 129, 5, "64x1-lane"), (40, 4000, 1, "8x8-lane"), (1000, 1000, 1, "2x32-lane"), (1024, 700, 1, "2x32-lane"),
                                 (1025, 1100, 2, "2x32-lane")):
        uniform = [(dna(l1), dna(l2)) if k % 2 else (lambda a: (a, (related(a) + dna(l2))[:l2]))(dna(l1)) for k in range(70 if l1 > 256 else 200)]
        res = al.align_batch("edit", uniform)
        assert "words/lane=%d" % want in al.last_config and groups in al.last_config, al.last_config
        for k, (a, b) in enumerate(uniform):
            assert int(res["score"][k]) == O.align(O.EDIT, a, b, 1, 1, -5, -1)["score"], (l1, l2, k)
    # ragged short reads, every length 1 .. 256 with every word boundary, one alignment per lane each with its own lengths
    short = [(dna(n), related(dna(n)) if n % 3 else dna(rng.randint(1, 300))) for n in range(1, 257)] + [("", "ACGT"), ("ACG", "")]
    res = al.align_batch("edit", short)
    assert "64x1-lane" in al.last_config and "words/lane=8" in al.last_config, al.last_config
    for k, (a, b) in enumerate(short):
        assert int(res["score"][k]) == O.align(O.EDIT, a, b, 1, 1, -5, -1)["score"], (k, len(a), len(b))
    # reads of 257 .. 1 024 bases: one alignment per lane as well (16 / 32 words) once the batch has a wavefront per CU -- here forced
    os.environ["AT_MYERS_LANE_MIN_PAIRS"] = "1"
    try:
        for l1, l2, want in ((257, 300, 16), (300, 300, 16), (512, 200, 16), (513, 600, 32), (1000, 1000, 32), (1024, 700, 32)):
            uniform = [(dna(l1), dna(l2)) if k % 2 else (lambda a: (a, (related(a) + dna(l2))[:l2]))(dna(l1)) for k in range(70)]
            res = al.align_batch("edit", uniform)
            assert "words/lane=%d" % want in al.last_config and "64x1-lane" in al.last_config, al.last_config
            for k, (a, b) in enumerate(uniform):
                assert int(res["score"][k]) == O.align(O.EDIT, a, b, 1, 1, -5, -1)["score"], (l1, l2, k)
        longer = [(dna(n), related(dna(n)) if n % 3 else dna(rng.randint(1, 1200))) for n in list(range(250, 1025, 7)) + [511, 512, 513, 991, 992, 993, 1023, 1024]]
        longer += [("", "ACGT"), ("ACG", ""), (dna(1024), dna(1))]
        res = al.align_batch("edit", longer)
        assert "64x1-lane" in al.last_config and "words/lane=32" in al.last_config, al.last_config
        for k, (a, b) in enumerate(longer):
            assert int(res["score"][k]) == O.align(O.EDIT, a, b, 1, 1, -5, -1)["score"], (k, len(a), len(b))
    finally:
        del os.environ["AT_MYERS_LANE_MIN_PAIRS"]
    al.set_scoring(1, 2, -5, -1)          # another mismatch cost: the DP kernel
    res = al.align_batch("edit", ragged[:20])
    assert "int32" in al.last_config
    al.set_scoring(1, 1, -5, -1)
    for l1, l2 in ((32769, 100), (500, 125000)):                  # longer than the bit-parallel kernel takes / than its LDS windows hold
        a, b = dna(l1), dna(l2)
        res = al.align_batch("edit", [(a, b)])
        assert "int32" in al.last_config and int(res["score"][0]) == O.align(O.EDIT, a, b, 1, 1, -5, -1)["score"], al.last_config


def test_device_entry_detects_uniform_batches_itself(al):
    """at_align_batch_device(uniform_shape = 0) on a batch that does have one shape: the device checks the lengths and the
    packed launch runs (the int32 launch queued behind it is a no-op); with one odd pair the roles swap.  Same results."""
    import torch
    import aligntools.c_amd as A
    rng = random.Random(808)
    n, l1, l2 = 5000, 120, 140
    dna = lambda k: "".join(rng.choice("ACGT") for _ in range(k)).encode()
    pairs = [(dna(l1), dna(l2)) for _ in range(n)]
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(a).to(dev)
    for odd in (False, True):
        if odd:
            pairs[1234] = (dna(l1 - 7), dna(l2))
        words, woff1, woff2, len1, len2, bits = A.pack_pairs(pairs)
        d_words, d_woff1, d_woff2, d_len1, d_len2 = t(words.view(np.int32)), t(woff1), t(woff2), t(len1), t(len2)
        d_ops_off = t(np.arange(n, dtype=np.int64) * (l1 + l2))
        d_res = torch.zeros((4, n), dtype=torch.int32, device=dev)
        d_nops = torch.zeros(n, dtype=torch.int32, device=dev)
        d_ops = torch.zeros(n * (l1 + l2) + 64, dtype=torch.uint8, device=dev)
        for mode, sc in (("local", (2, -2, -5, -2)), ("global", (1, -1, -4, -1)), ("fit", (2, -2, -5, -1))):
            al.set_scoring(*sc)
            d_res.fill_(-77)
            al.align_batch_device(A.MODES[mode], n, d_words.data_ptr(), bits, d_woff1.data_ptr(), d_len1.data_ptr(), d_woff2.data_ptr(),
                                  d_len2.data_ptr(), l1, l2, False, True, d_res[0].data_ptr(), d_res[1].data_ptr(), d_res[2].data_ptr(),
                                  d_res[3].data_ptr(), d_ops.data_ptr(), d_ops_off.data_ptr(), d_nops.data_ptr(),
                                  torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            assert al.last_config.startswith("auto:") and "packed16" in al.last_config and "int32" in al.last_config, al.last_config
            r, nops, ops = d_res.cpu().numpy(), d_nops.cpu().numpy(), d_ops.cpu().numpy()
            for k in list(range(0, n, 97)) + [1234]:
                ref = O.align(O.MODE_NAMES[mode], pairs[k][0], pairs[k][1], *sc)
                got = (int(r[0, k]), int(r[1, k]), int(r[2, k]), int(r[3, k]), ops[k * (l1 + l2):k * (l1 + l2) + int(nops[k])].tobytes())
                assert got == (ref["score"], ref["end_i"], ref["end_j"], ref["state"], ref["ops"]), (odd, mode, k)


def test_local_packed_with_large_scores(al):
    """Local matrices hold no -inf and do not drift downwards (M >= 0, L and U >= o), so the 16-bit range check of the
    packed kernel only has to cover m * min(l1, l2): BLAST-like and minimap2-like scorings stay on the packed kernels."""
    rng = random.Random(5150)
    dna = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    for sc, l1, l2 in (((5, -4, -10, -1), 150, 150), ((2, -8, -12, -2), 150, 160), ((2, -6, -5, -3), 100, 250), ((10, -9, -20, -5), 150, 150),
                       ((1, -3, -5, -2), 1000, 1100)):
        pairs = []
        for k in range(80):
            a = dna(l1)
            if k % 2:
                t = list(a)
                for _ in range(l1 // 15):
                    q = rng.randrange(len(t))
                    r = rng.random()
                    if r < 0.5:
                        t[q] = rng.choice("ACGT")
                    elif r < 0.75:
                        del t[q]
                    else:
                        t.insert(q, rng.choice("ACGT"))
                b = (dna(rng.randint(0, 20)) + "".join(t) + dna(l2))[:l2]
            else:
                b = dna(l2)
            pairs.append((a, b))
        al.set_scoring(*sc)
        res = al.align_batch("local", pairs, render=False)
        assert "packed16" in al.last_config, (sc, al.last_config)
        for k, (a, b) in enumerate(pairs):
            r = O.align(O.LOCAL, a, b, *sc)
            assert (int(res["score"][k]), int(res["end_i"][k]), int(res["end_j"][k]), res["ops"][k]) == (r["score"], r["end_i"], r["end_j"], r["ops"]), (sc, k)
    for sc, want in (((20, -30, -50, -10), "packed16 x4"),      # 20 * 150 * 16 > 2^15: scores x4 (and byte words: no byte LUT)
                     ((200, -300, -500, -100), "int32")):        # 200 * 150 * 4 > 2^15: the int32 kernel
        al.set_scoring(*sc)
        pairs = [(dna(150), dna(150)) for _ in range(70)]
        res = al.align_batch("local", pairs, render=False)
        assert want in al.last_config, (sc, al.last_config)
        for k, (a, b) in enumerate(pairs):
            r = O.align(O.LOCAL, a, b, *sc)
            assert (int(res["score"][k]), res["ops"][k]) == (r["score"], r["ops"]), (sc, k)


@pytest.mark.parametrize("mode", ["global", "fit"])
def test_packed_range_check_harsh_mismatch(al, mode):
    """Global / fit values are bounded below by |u| * min(l1, l2) + |e| * max(l1, l2) + openings (a max over paths is at least
    one path): scorings with a harsh mismatch still fit the packed kernel's 16 bits.  Worst cases included: nothing matches,
    one-sided gaps, everything matches."""
    rng = random.Random(7788)
    dna = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    for sc, l1, l2 in (((2, -8, -12, -2), 150, 150), ((1, -9, -3, -1), 150, 170), ((3, -6, -8, -4), 120, 130), ((2, -8, -12, -2), 100, 250)):
        if mode == "global" and l2 > 200:
            continue
        a = dna(l1)
        pairs = [("A" * l1, "C" * l2), ("A" * l1, "A" * l2), (a, (a[l1 // 2:] + a)[:l2].ljust(l2, "T")), (a, ("G" * (l2 - l1) + a)[:l2]),
                 ("AC" * (l1 // 2), ("CA" * l2)[:l2])]
        pairs += [(dna(l1), dna(l2)) for _ in range(60)]
        pairs = [(x[:l1].ljust(l1, "A"), y[:l2].ljust(l2, "C")) for x, y in pairs]
        for uj in ((False, True) if mode == "fit" else (False,)):
            al.set_scoring(*sc, -9, uj, [20, 60, 100])
            res = al.align_batch(mode, pairs, render=False)
            assert "packed16" in al.last_config, (sc, l1, l2, al.last_config)
            for k, (x, y) in enumerate(pairs):
                r = O.align(O.MODE_NAMES[mode], x, y, *sc, -9, uj, [20, 60, 100])
                assert (int(res["score"][k]), int(res["end_i"][k]), int(res["end_j"][k]), int(res["state"][k]), res["ops"][k]) == \
                       (r["score"], r["end_i"], r["end_j"], r["state"], r["ops"]), (mode, sc, uj, k)


@pytest.mark.parametrize("mode", ["local", "global", "fit", "fitj"])
def test_packed_32_lane_groups(al, mode):
    """Reads of 209..304 bases (250- and 300-base reads) run as four groups of 16 lanes with 16 / 19 rows per lane, 305..608 as two
    groups of 32 lanes with 10 / 12 / 13 / 16 / 19 rows per lane, and with AT_GROUP=32 all of 209..416 on the 32-lane groups (7 / 8 / 10 / 13 rows):
    every class edge, both alphabets, related and unrelated pairs, with and without tracebacks, against the oracle."""
    rng = random.Random(3232)
    uj = mode == "fitj"
    m = "fit" if uj else mode
    import os
    for l1, l2, alpha, force32 in ((209, 209, "ACGT", 0), (224, 260, "ACGT", 1), (225, 225, "ACGTN", 1), (250, 250, "ACGT", 0), (250, 250, "ACGT", 1),
                                   (256, 300, "ACGT", 0), (257, 257, "ACGT", 0), (257, 257, "ACGT", 1), (300, 300, "ACGT", 0), (300, 300, "ACGTN", 1),
                                   (304, 2000, "ACGT", 0), (305, 305, "ACGT", 0), (320, 400, "ACGTN", 0), (321, 321, "ACGT", 0), (321, 321, "ACGT", 1),
                                   (384, 400, "ACGT", 0), (385, 385, "ACGTN", 0), (416, 416, "ACGT", 0), (417, 430, "ACGT", 0), (512, 512, "ACGT", 0),
                                   (513, 520, "ACGT", 0), (608, 700, "ACGTN", 0), (609, 609, "ACGT", 0)):
        pairs = []
        for k in range(24):
            a = "".join(rng.choice(alpha) for _ in range(l1))
            if k % 2:
                t = list(a)
                for _ in range(l1 // 18):
                    q = rng.randrange(len(t))
                    r = rng.random()
                    if r < 0.5:
                        t[q] = rng.choice(alpha)
                    elif r < 0.75:
                        del t[q]
                    else:
                        t.insert(q, rng.choice(alpha))
                b = ("".join(rng.choice(alpha) for _ in range(rng.randint(0, 25))) + "".join(t) + "".join(rng.choice(alpha) for _ in range(l2)))[:l2]
            else:
                b = "".join(rng.choice(alpha) for _ in range(l2))
            pairs.append((a, b))
        sc = (2, -2, -5, -2, -9) if mode != "global" else (1, -2, -4, -1, -9)
        al.set_scoring(*sc, uj, [50, 150, 250])
        if force32:
            os.environ["AT_GROUP"] = "32"
        try:
            res = al.align_batch(m, pairs, render=False)
            cfg = al.last_config
            res0 = al.align_batch(m, pairs, traceback=False, render=False)
        finally:
            os.environ.pop("AT_GROUP", None)
        assert "packed16" in cfg, (l1, l2, cfg)
        if "packed16 x16" in cfg:      # (scores x4 when x16 would leave 16 bits: one 64-lane group)
            hi32 = 416 if force32 else {"local": 608, "global": 512}.get(mode, 416)   # (16 rows per lane: local and global; 19: local)
            want = "4x16-lane" if l1 <= 304 and not force32 else "2x32-lane" if l1 <= hi32 else "1x64-lane"
            assert want + " groups" in cfg, (l1, l2, cfg)
        for k, (x, y) in enumerate(pairs):
            r = O.align(O.MODE_NAMES[m], x, y, *sc, uj, [50, 150, 250])
            assert (int(res["score"][k]), int(res["end_i"][k]), int(res["end_j"][k]), int(res["state"][k]), res["ops"][k]) == \
                   (r["score"], r["end_i"], r["end_j"], r["state"], r["ops"]), (mode, l1, l2, k)
            assert (int(res0["score"][k]), int(res0["end_i"][k]), int(res0["end_j"][k]), int(res0["state"][k])) == \
                   (r["score"], r["end_i"], r["end_j"], r["state"]), (mode, l1, l2, k, "scores only")


@pytest.mark.parametrize("mode", ["local", "global", "fit", "fitj"])
def test_packed_8_lane_groups(al, mode):
    """Reads of up to 152 bases run as eight groups of 8 lanes (16 alignments per wavefront), 5 / 7 / 10 / 13 / 16 / 19 rows per
    lane: every class edge, both alphabets, related and unrelated pairs, batch sizes that leave the last work item partly
    empty, against the oracle.  19 rows per lane exercise the second chain of the local arg-max (rows 16..18 of a lane)."""
    rng = random.Random(808)
    uj = mode == "fitj"
    m = "fit" if uj else mode
    for l1, l2, alpha, n in ((1, 9, "ACGT", 3), (4, 4, "ACGT", 70), (36, 36, "ACGT", 40), (37, 500, "ACGT", 65), (40, 120, "ACGT", 17), (41, 41, "ACGTN", 33), (48, 48, "ACGT", 20), (49, 60, "ACGT", 17),
                             (52, 52, "ACGTN", 64), (53, 53, "ACGT", 33), (56, 70, "ACGT", 16), (57, 57, "ACGT", 31),
                             (64, 64, "ACGT", 33), (65, 65, "ACGTN", 18), (64, 300, "ACGTN", 70), (76, 76, "ACGT", 64), (75, 90, "ACGT", 31), (77, 77, "ACGT", 33),
                             (80, 80, "ACGT", 48), (81, 150, "ACGTN", 19), (104, 104, "ACGT", 23), (105, 130, "ACGT", 32), (128, 128, "ACGT", 35),
                             (129, 500, "ACGT", 21), (150, 150, "ACGT", 50), (150, 500, "ACGT", 37), (152, 152, "ACGTN", 18), (151, 153, "ACGT", 5)):
        pairs = []
        for k in range(n):
            a = "".join(rng.choice(alpha) for _ in range(l1))
            if k % 2:
                t = list(a)
                for _ in range(l1 // 18):
                    q = rng.randrange(len(t))
                    r = rng.random()
                    if r < 0.5:
                        t[q] = rng.choice(alpha)
                    elif r < 0.75 and len(t) > 1:
                        del t[q]
                    else:
                        t.insert(q, rng.choice(alpha))
                b = ("".join(rng.choice(alpha) for _ in range(rng.randint(0, 25))) + "".join(t) + "".join(rng.choice(alpha) for _ in range(l2)))[:l2]
            else:
                b = "".join(rng.choice(alpha) for _ in range(l2))
            pairs.append((a, b))
        sc = (2, -2, -5, -2, -9) if mode != "global" else (1, -2, -4, -1, -9)
        sites = [5, 50, 150, 250]
        al.set_scoring(*sc, uj, sites)
        want = [O.align(O.MODE_NAMES[m], x, y, *sc, uj, sites) for x, y in pairs]
        # reads of up to 76 bases run on sixteen groups of 4 lanes (32 alignments per wavefront, 9 / 10 / 13 / 16 / 19 rows per lane);
        # AT_GROUP=8 keeps them on the 8-lane groups (5 / 6 / 7 / 8 / 10 rows per lane)
        for force8 in ((False, True) if l1 <= 76 else (False,)):
            for tb in (True, False):
                if force8:
                    os.environ["AT_GROUP"] = "8"
                try:
                    res = al.align_batch(m, pairs, traceback=tb, render=False)
                finally:
                    os.environ.pop("AT_GROUP", None)
                assert ("16x4-lane groups" if l1 <= 76 and not force8 else "8x8-lane groups") in al.last_config, (l1, l2, al.last_config)
                for k, r in enumerate(want):
                    assert (int(res["score"][k]), int(res["end_i"][k]), int(res["end_j"][k]), int(res["state"][k])) == \
                           (r["score"], r["end_i"], r["end_j"], r["state"]), (mode, l1, l2, k, tb, force8)
                    if tb:
                        assert res["ops"][k] == r["ops"], (mode, l1, l2, k, force8)


@pytest.mark.parametrize("mode", ["local", "global", "fit", "fitj"])
def test_sliver_of_a_batch_on_64_lane_groups(al, mode, monkeypatch):
    """A uniform batch that fills the resident waves a whole number of times plus a sliver (less than a quarter of a round) ends on
    items of two 32-lane groups: the sliver's pairs follow the 4- / 8- / 16-lane items in the same launch and work queue (the
    kernel's second argument; at_hip.hip, align_device).  AT_TAIL_SPLIT=0 keeps every pair on the narrow groups.
    The split batch equals the unsplit one everywhere and the oracle on the sliver and on a sample of the rest."""
    import re
    monkeypatch.setenv("AT_HOST_CHUNKS", "1")      # the host entry would cut the batch into chunks side by side: one launch here
    rng = random.Random(4242)
    uj = mode == "fitj"
    m = "fit" if uj else mode
    sc = (2, -2, -5, -2, -9) if mode != "global" else (1, -2, -4, -1, -9)
    sites = [5, 30, 44]
    al.set_scoring(*sc, uj, sites)
    for l1, l2 in ((60, 64), (170, 200)):
        def mk():
            a = "".join(rng.choice("ACGT") for _ in range(l1))
            if rng.random() < 0.6:
                t = list(a)
                for _ in range(1 + l1 // 15):
                    q = rng.randrange(len(t))
                    r = rng.random()
                    if r < 0.5:
                        t[q] = rng.choice("ACGT")
                    elif r < 0.75 and len(t) > 1:
                        del t[q]
                    else:
                        t.insert(q, rng.choice("ACGT"))
                b = ("".join(rng.choice("ACGT") for _ in range(rng.randint(0, 9))) + "".join(t) + "".join(rng.choice("ACGT") for _ in range(l2)))[:l2]
            else:
                b = "".join(rng.choice("ACGT") for _ in range(l2))
            return a, b
        probe = [mk() for _ in range(64)]
        fresh = [mk() for _ in range(700)]
        base = [mk() for _ in range(2048)]
        for tb in (True, False):
            al.align_batch(m, probe * 2048, traceback=tb, render=False)    # one full-size call tells the grid of this kernel on this chip
            g = re.search(r"(\d+)x(\d+)-lane groups \((\d+) pairs/wave\).* grid=(\d+)", al.last_config)
            assert g and int(g.group(2)) in (4, 8, 16), al.last_config
            per_wave, grid = int(g.group(3)), int(g.group(4))
            n = (grid + 37) * per_wave - 3                              # one round, then 37 work items, the last one not full
            pairs = (base * (n // 2048 + 1))[:n - 700] + fresh
            monkeypatch.setenv("AT_TAIL_SPLIT", "0")
            whole = al.align_batch(m, pairs, traceback=tb, render=False)
            assert "32-lane items" not in al.last_config
            monkeypatch.setenv("AT_TAIL_SPLIT", "1")
            res = al.align_batch(m, pairs, traceback=tb, render=False)
            assert "+ last %d pairs as 32-lane items" % (37 * per_wave - 3) in al.last_config, al.last_config
            for key in ("score", "end_i", "end_j", "state"):
                assert (np.asarray(res[key]) == np.asarray(whole[key])).all(), (mode, l1, key)
            if tb:
                assert res["ops"] == whole["ops"], (mode, l1)
            for k in list(range(n - 700, n)) + list(range(0, n - 700, 97)):
                x, y = pairs[k]
                r = O.align(O.MODE_NAMES[m], x, y, *sc, uj, sites)
                assert (int(res["score"][k]), int(res["end_i"][k]), int(res["end_j"][k]), int(res["state"][k])) == \
                       (r["score"], r["end_i"], r["end_j"], r["state"]), (mode, l1, k, tb)
                if tb:
                    assert res["ops"][k] == r["ops"], (mode, l1, k)


def test_short_read_against_long_target_falls_back_to_int32(al):
    """A read of up to 208 bases against a second sequence too long for the packed kernels' LDS window (more than ~4 000
    bases) has no packed instantiation: single pair, uniform batch and ragged batch all run on the int32 kernel and match
    the oracle (round-1 advisor finding: they used to fail with AT_ERR_RANGE)."""
    rng = random.Random(5150)
    al.set_scoring(2, -2, -5, -2, -10, False, [])
    for l2 in (5000, 20000):
        for n, ragged in ((1, False), (9, False), (70, True)):
            pairs = []
            for k in range(n):
                a = "".join(rng.choice("ACGT") for _ in range(150 - (k % 7 if ragged else 0)))
                t = "".join(rng.choice("ACGT") for _ in range(l2 - (k * 13 if ragged else 0)))
                q = rng.randrange(len(t) - 200)
                t = t[:q] + a[10:140] + t[q + 130:]
                pairs.append((a, t))
            for mode in ("local", "fit"):
                res = al.align_batch(mode, pairs, render=False)
                for k in range(0, n, max(1, n // 6)):
                    r = O.align(O.MODE_NAMES[mode], pairs[k][0], pairs[k][1], 2, -2, -5, -2, -10, False, [])
                    assert (int(res["score"][k]), int(res["end_i"][k]), int(res["end_j"][k]), res["ops"][k]) == \
                           (r["score"], r["end_i"], r["end_j"], r["ops"]), (mode, l2, n, k, al.last_config)


def test_host_entry_writes_only_its_slots(al):
    """at_align_batch writes exactly nops[k] bytes of pair k's slot and nothing between or behind the slots, whatever order
    the slots are in (round-1 advisor finding: the whole span used to be copied back); at_align_batch_strings likewise
    (nops[k] characters and the terminating 0)."""
    import aligntools.c_amd as A
    rng = random.Random(99)
    pairs = [(A._b("".join(rng.choice("ACGT") for _ in range(60))), A._b("".join(rng.choice("ACGT") for _ in range(80)))) for _ in range(40)]
    al.set_scoring(2, -2, -5, -2, -10, False, [])
    ref = al.align_batch("local", pairs, render=True)
    n = len(pairs)
    blob, off1, len1, off2, len2 = A._flatten(pairs)
    slot = 60 + 80 + 1
    ops_off = np.array([(n - 1 - k) * (slot + 37) + 11 for k in range(n)], dtype=np.int64)   # reversed, with gaps
    total = int(ops_off.max()) + slot + 100
    for strings in (False, True):
        score, ei, ej, st, nops = (np.zeros(n, dtype=np.int32) for _ in range(5))
        b1 = np.full(total, 0xEE, dtype=np.uint8)
        b2 = np.full(total, 0xEE, dtype=np.uint8)
        if strings:
            al._check(al._lib.at_align_batch_strings(al._h, A.MODES["local"], n, A._ptr(blob), A._ptr(off1), A._ptr(len1), A._ptr(off2), A._ptr(len2),
                                                     A._ptr(score), A._ptr(ei), A._ptr(ej), A._ptr(st), A._ptr(b1), A._ptr(b2), A._ptr(ops_off), A._ptr(nops)))
        else:
            al._check(al._lib.at_align_batch(al._h, A.MODES["local"], n, A._ptr(blob), A._ptr(off1), A._ptr(len1), A._ptr(off2), A._ptr(len2), 1,
                                             A._ptr(score), A._ptr(ei), A._ptr(ej), A._ptr(st), A._ptr(b1), A._ptr(ops_off), A._ptr(nops)))
        assert (score == ref["score"]).all() and (nops == ref["nops"]).all()
        written = np.zeros(total, dtype=bool)
        for k in range(n):
            o, c = int(ops_off[k]), int(nops[k])
            if strings:
                assert bytes(b1[o:o + c]).decode("latin1") == ref["r1"][k] and bytes(b2[o:o + c]).decode("latin1") == ref["r2"][k], k
                assert b1[o + c] == 0 and b2[o + c] == 0
                written[o:o + c + 1] = True
            else:
                assert bytes(b1[o:o + c]) == ref["ops"][k], k
                written[o:o + c] = True
        assert (b1[~written] == 0xEE).all(), "bytes outside the slots were written"
        if strings:
            assert (b2[~written] == 0xEE).all()


@pytest.mark.parametrize("case", [
    # mode, jump state, l1, l2, pairs, scoring, sites: what it reaches
    ("global", False, 1024, 1024, 300, (1, -1, -4, -1, -10), []),             # 64 lanes x 16 rows, scores x4 (C3's shape)
    ("local", False, 1000, 1024, 300, (2, -2, -5, -2, -10), []),              # scores x4, HOME inside a block
    ("local", False, 700, 640, 400, (1, -1, -1, -1, -10), []),                # scores x16, tie-heavy, l2 < l1
    ("fit", False, 700, 1024, 300, (1, -1, -4, -1, -10), []),
    ("fit", True, 620, 660, 300, (1, -1, -4, -1, -6), [100, 450, 451, 600]),  # scores x16 with the jump plane on the 64-lane group
    ("global", False, 620, 3000, 120, (1, -1, -1, -1, -10), []),              # 190 column blocks per lane
    ("local", False, 150, 150, 700, (2, -2, -5, -2, -10), []),                # AT_TWO_PASS=2 only: 8 lanes x 19 rows (C2's shape)
    ("fit", True, 150, 500, 500, (2, -2, -5, -1, -10), [100, 200, 300, 400]), # ... C4's shape
    ("global", False, 129, 140, 500, (1, -1, -1, -1, -10), []),
    ("fit", True, 140, 300, 500, (1, -1, -1, -1, -2), [7, 50, 51, 120]),
])
@pytest.mark.parametrize("walk_kernel", [0, 1])
def test_two_pass_tracebacks(al, case, walk_kernel, monkeypatch):
    """Two-pass tracebacks (at_sweep16.hip.h, CK kernels: the scores-only sweep leaves checkpoints, the pointers are rebuilt block by
    block where the walks go) against the one-pass kernels on the whole batch -- score, end cell, start state, ops -- and against the
    oracle on a sample; related and unrelated pairs, batches that leave the last work item partly empty.  walk_kernel = 1: pass 2 as a
    kernel of its own (at_walk16.hip.h, AT_TP_SPLIT=1: one walker per half-lane)."""
    mode, uj, l1, l2, n, sc, sites = case
    monkeypatch.setenv("AT_TP_SPLIT", str(walk_kernel))
    rng = random.Random(l1 * 7919 + l2)

    def mk(related):
        a = "".join(rng.choice("ACGT") for _ in range(l1))
        if not related:
            return a, "".join(rng.choice("ACGT") for _ in range(l2))
        t = list(a)
        for _ in range(1 + l1 // 15):
            q = rng.randrange(len(t))
            r = rng.random()
            if r < 0.5:
                t[q] = rng.choice("ACGT")
            elif r < 0.75 and len(t) > 1:
                del t[q]
            else:
                t.insert(q, rng.choice("ACGT"))
        return a, ("".join(rng.choice("ACGT") for _ in range(rng.randint(0, max(0, l2 - l1)))) + "".join(t) + "".join(rng.choice("ACGT") for _ in range(l2)))[:l2]

    pairs = [mk(k % 2 == 1) for k in range(n - 3)]
    al.set_scoring(*sc, uj, sites)
    monkeypatch.setenv("AT_HOST_CHUNKS", "1")
    monkeypatch.setenv("AT_TWO_PASS", "2")
    two = al.align_batch(mode, pairs, traceback=True, render=False)
    assert "two-pass" in al.last_config and ("walk kernel" in al.last_config) == bool(walk_kernel), al.last_config
    monkeypatch.setenv("AT_TWO_PASS", "0")
    one = al.align_batch(mode, pairs, traceback=True, render=False)
    assert "two-pass" not in al.last_config, al.last_config
    for key in ("score", "end_i", "end_j", "state", "nops"):
        assert (np.asarray(two[key]) == np.asarray(one[key])).all(), (case, key)
    assert two["ops"] == one["ops"], case
    for k in range(0, len(pairs), max(1, len(pairs) // (12 if l1 > 300 else 60))):
        r = O.align(O.MODE_NAMES[mode], pairs[k][0], pairs[k][1], *sc, uj, sites)
        assert (int(two["score"][k]), int(two["end_i"][k]), int(two["end_j"][k]), int(two["state"][k])) == \
               (r["score"], r["end_i"], r["end_j"], r["state"]), (case, k)
        assert two["ops"][k] == r["ops"], (case, k)


@pytest.mark.parametrize("case", [
    # mode, jump state, l1, l2, pairs, scoring, sites, alphabet, AT_CK_PIECE_PAIRS: what it reaches
    ("local", False, 150, 150, 20000, (2, -2, -5, -2, -10), [], "ACGT", 6000),          # a batch in pieces (as when its checkpoints exceed AT_CK_CAP_MB)
    ("global", False, 1000, 1024, 700, (1, -1, -4, -1, -10), [], "ACGT", 256),          # ... on the 64-lane group (teams of walker lanes)
    ("local", False, 150, 150, 33408, (2, -2, -5, -2, -10), [], "ACGT", 0),             # a sliver of 640 pairs on two 32-lane groups x 5 rows, walked by the first wavefronts
    ("fit", True, 150, 500, 33088, (2, -2, -5, -1, -10), [100, 200, 300, 400], "ACGT", 0),   # ... with the jump state
    ("global", False, 150, 150, 701, (1, -1, -1, -1, -10), [], "ACGTN", 0),             # byte words (reads with N): the _b8 kernels; an odd batch
    ("local", False, 1000, 1024, 257, (1, -1, -1, -1, -10), [], "ACGTN", 0),            # ... on the 64-lane group; the last lane holds one alignment
    ("global", False, 150, 150, 1, (2, -2, -5, -2, -10), [], "ACGT", 0),                # one alignment
    ("local", False, 150, 150, 33408, (2, -2, -5, -2, -10), [], "ACGT", -2),            # AT_WALK_TEAMS=1: teams of 2 lanes on the 8-lane groups, with a sliver
    ("fit", True, 150, 500, 3001, (2, -2, -5, -1, -10), [100, 200, 300, 400], "ACGT", -2),   # ... the jump state, an odd batch
])
def test_two_pass_walk_kernel_batches(al, case, monkeypatch):
    """Pass 2 as a kernel of its own (AT_TP_SPLIT=1, at_walk16.hip.h): batches cut into pieces by the checkpoint cap, batches that end with
    a sliver on 32-lane groups, byte alphabets, odd and tiny batches -- the whole batch against the one-pass kernels, a sample against the
    oracle."""
    mode, uj, l1, l2, n, sc, sites, alpha, cap = case
    rng = random.Random(l1 * 31 + l2 + n)

    def mk(related):
        a = "".join(rng.choice(alpha) for _ in range(l1))
        if not related:
            return a, "".join(rng.choice(alpha) for _ in range(l2))
        t = list(a)
        for _ in range(1 + l1 // 12):
            q = rng.randrange(len(t))
            r = rng.random()
            if r < 0.5:
                t[q] = rng.choice(alpha)
            elif r < 0.75 and len(t) > 1:
                del t[q]
            else:
                t.insert(q, rng.choice(alpha))
        return a, ("".join(rng.choice(alpha) for _ in range(rng.randint(0, max(0, l2 - l1)))) + "".join(t) + "".join(rng.choice(alpha) for _ in range(l2)))[:l2]

    uniq = [mk(k % 3 != 0) for k in range(min(n, 397))]
    pairs = (uniq * (n // len(uniq) + 1))[:n]
    al.set_scoring(*sc, uj, sites)
    monkeypatch.setenv("AT_HOST_CHUNKS", "1")
    monkeypatch.setenv("AT_TWO_PASS", "2")
    monkeypatch.setenv("AT_TP_SPLIT", "1")
    if cap > 0:
        monkeypatch.setenv("AT_CK_PIECE_PAIRS", str(cap))
    if cap == -2:
        monkeypatch.setenv("AT_WALK_TEAMS", "1")
    two = al.align_batch(mode, pairs, traceback=True, render=False)
    cfg = al.last_config
    assert "walk kernel" in cfg, cfg
    if cap > 0:
        assert "in pieces of" in cfg, cfg
    if n > 33000:
        assert "as 32-lane items" in cfg, cfg
    if "N" in alpha:
        assert "bits=8" in cfg, cfg
    monkeypatch.setenv("AT_TWO_PASS", "0")
    one = al.align_batch(mode, pairs, traceback=True, render=False)
    assert "two-pass" not in al.last_config, al.last_config
    for key in ("score", "end_i", "end_j", "state", "nops"):
        assert (np.asarray(two[key]) == np.asarray(one[key])).all(), (case, key)
    assert two["ops"] == one["ops"], case
    for k in list(range(0, min(n, len(uniq)), 9)) + [n - 1]:
        r = O.align(O.MODE_NAMES[mode], pairs[k][0], pairs[k][1], *sc, uj, sites)
        assert (int(two["score"][k]), int(two["end_i"][k]), int(two["end_j"][k]), int(two["state"][k])) == \
               (r["score"], r["end_i"], r["end_j"], r["state"]), (case, k)
        assert two["ops"][k] == r["ops"], (case, k)


def test_two_pass_default_routing(al, monkeypatch):
    """By default the two-pass kernels take the shapes on which they win -- one strip of 64 lanes x 16 rows (reads of 609 .. 1 024
    bases; pass 2 there is the walk kernel with teams of lanes), and fit of 129 .. 152-base reads against a second sequence at least
    one and a half times as long (the walk kernel, one walker per half-lane) -- and leave the others to the one-pass kernels."""
    rng = random.Random(5)
    monkeypatch.delenv("AT_TWO_PASS", raising=False)
    monkeypatch.delenv("AT_TP_SPLIT", raising=False)
    monkeypatch.setenv("AT_HOST_CHUNKS", "1")
    al.set_scoring(1, -1, -4, -1, -10, False, [])
    for l1, want in ((1024, True), (609, True), (150, False), (1025, False)):
        pairs = [("".join(rng.choice("ACGT") for _ in range(l1)), "".join(rng.choice("ACGT") for _ in range(l1 + 20))) for _ in range(40)]
        al.align_batch("global", pairs, traceback=True, render=False)
        assert ("two-pass" in al.last_config) == want and ("walk kernel" in al.last_config) == want, (l1, al.last_config)
        al.align_batch("global", pairs, traceback=False, render=False)
        assert "two-pass" not in al.last_config
    for mode, uj, l1, l2, want in (("fit", True, 150, 500, True), ("fit", False, 140, 300, True), ("fit", True, 150, 200, False), ("fit", True, 150, 226, True),
                                   ("local", False, 150, 500, False), ("fit", False, 120, 500, False)):
        al.set_scoring(2, -2, -5, -1, -10, uj, [100, 150] if uj else [])
        pairs = [("".join(rng.choice("ACGT") for _ in range(l1)), "".join(rng.choice("ACGT") for _ in range(l2))) for _ in range(80)]
        res = al.align_batch(mode, pairs, traceback=True, render=False)
        assert ("walk kernel" in al.last_config) == want, (mode, uj, l1, l2, al.last_config)
        for k in (0, 41, 79):
            r = O.align(O.MODE_NAMES[mode], pairs[k][0], pairs[k][1], 2, -2, -5, -1, -10, uj, [100, 150] if uj else [])
            assert (int(res["score"][k]), res["ops"][k]) == (r["score"], r["ops"]), (mode, l1, l2, k)


@pytest.mark.parametrize("shape", [(1000, 1000, 260), (300, 1000, 300), (150, 160, 500), (40, 64, 400), (500, 512, 200)])
def test_overlap_threshold_filter(al, shape):
    """All-vs-all overlap scores with a threshold (at_set_min_score; at_myers.hip.h, SEMI): pairs whose score is proven below T by the
    bit-parallel bound are not swept.  Against the unthresholded sweep of the same triangle: every pair the filter let through
    (state 2) carries the exact score and end cell; every pair it stopped (state 0) really scores below T, and the bound it reports
    is one; unrelated reads are stopped, planted overlaps are not.  Read sets of mixed lengths, every word class of the filter."""
    import torch
    import aligntools.c_amd as A
    lo, hi, n = shape
    rng = random.Random(lo * 31 + hi)
    reads = ["".join(rng.choice("ACGT") for _ in range(rng.randint(lo, hi))) for _ in range(n)]
    planted = []
    for k in range(0, n - 1, 7):            # read k + 1 starts with (a noisy copy of) the end of read k
        a = reads[k]
        ov = rng.randint(min(20, len(a)), max(min(20, len(a)), min(len(a), hi) * 3 // 4))
        piece = list(a[len(a) - ov:])
        for _ in range(ov // 25):
            q = rng.randrange(len(piece))
            r = rng.random()
            if r < 0.6:
                piece[q] = rng.choice("ACGT")
            elif r < 0.8 and len(piece) > 1:
                del piece[q]
            else:
                piece.insert(q, rng.choice("ACGT"))
        b = ("".join(piece) + reads[k + 1])[:max(lo, min(hi, len(reads[k + 1])))]
        reads[k + 1] = b
        planted.append((k, k + 1))
    words, woff, _w2, lens, _l2, bits = A.pack_pairs([(r.encode(), b"") for r in reads])
    assert bits == 2
    dev = torch.device("cuda", 0)
    d_words = torch.from_numpy(words.view(np.int32)).to(dev)
    d_woff = torch.from_numpy(woff).to(dev)
    d_len = torch.from_numpy(lens).to(dev)
    total = n * (n - 1) // 2
    maxl = int(lens.max())
    stream = torch.cuda.current_stream().cuda_stream

    def sweep():
        res = torch.full((4, total), -7, dtype=torch.int32, device=dev)
        al.align_allpairs_device(A.MODES["overlap"], n, d_words.data_ptr(), bits, d_woff.data_ptr(), d_len.data_ptr(), maxl, 0, total, False,
                                 res[0].data_ptr(), res[1].data_ptr(), res[2].data_ptr(), res[3].data_ptr(), 0, 0, 0, stream)
        torch.cuda.synchronize()
        return res.cpu().numpy()

    for sc in ((1, -2, -5, -1), (2, -3, -4, -1)):
        al.set_scoring(*sc)
        al.set_min_score(None)
        exact = sweep()
        assert "filter" not in al.last_config
        for T in (12, 30, 10 ** 6):
            al.set_min_score(T)
            try:
                got = sweep()
            finally:
                al.set_min_score(None)
            assert "overlap filter" in al.last_config, al.last_config
            swept = got[3] == 2
            assert ((got[3] == 0) | swept).all()
            for row in range(3):
                assert (got[row][swept] == exact[row][swept]).all(), (sc, T, row)
            assert (exact[0][~swept] < T).all(), (sc, T)                   # nothing that reaches T was stopped
            assert (got[0][~swept] >= exact[0][~swept]).all() and (got[0][~swept] < T).all()   # ... and what it reports is a bound
            assert (exact[0][swept] >= T).sum() == (exact[0] >= T).sum()
            if T == 30 and sc == (1, -2, -5, -1) and lo >= 150:
                assert swept.mean() < 0.05, swept.mean()                   # unrelated reads are stopped by the bound
            if T == 10 ** 6:
                assert not swept.any()
        # the planted overlaps score high and are swept
        al.set_min_score(20)
        try:
            got = sweep()
        finally:
            al.set_min_score(None)
        t_of = lambda a, b: a * n - a * (a + 1) // 2 + (b - a - 1)
        hits = [p for p in planted if exact[0][t_of(*p)] >= 20]
        assert len(hits) >= len(planted) // 2
        for p in hits:
            assert got[3][t_of(*p)] == 2
    # the oracle on the planted pairs (default scoring: what `alignTools overlap` can reach)
    al.set_scoring(1, -2, -5, -1)
    exact = sweep()
    for a, b in planted[:12]:
        r = O.align(O.OVERLAP, reads[a], reads[b], 1, -2, -5, -1)
        assert (int(exact[0][t_of(a, b)]), int(exact[2][t_of(a, b)])) == (r["score"], r["end_j"]), (a, b)
