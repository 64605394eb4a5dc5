"""Randomised parity campaign: HIP path (through the C ABI) against the oracle restatement on freshly drawn cases --
modes, scorings (tie-heavy, zero and large penalties), lengths, alphabets, uniform and ragged batches, jump sites,
with and without tracebacks; every fourth batch with tracebacks also through the GPU rendering of the two gapped strings.  `python tests/fuzz_parity.py [cases] [seed]` on a GPU box; tests/test_fuzz.py runs a
short campaign inside the GPU suite."""
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as O   # noqa: E402  (test infrastructure)

SCORINGS = [(1, -2, -5, -1, -10), (2, -2, -5, -2, -10), (1, -1, -1, -1, -1), (1, -1, -4, -1, -10), (3, -1, 0, 0, 0), (2, -3, -4, 0, -3),
            (5, -4, -10, -1, -7), (1, 0, -1, -1, -2), (7, -5, -12, -3, -20), (0, -1, -1, -1, -1), (40, -30, -60, -20, -50), (2, 1, -3, -1, -5)]
ALPHABETS = ["ACGT", "ACGT", "ACGT", "AC", "A", "ACGTN", "ACDEFGHIKLMNPQRSTVWY", "ab"]


def draw_batch(rng):
    only = os.environ.get("AT_FUZZ_MODES")          # e.g. "fitj,overlap": a campaign aimed at those kernels
    mode = rng.choice(only.split(",") if only else ["global", "local", "fit", "fit", "overlap", "edit"])
    sc = rng.choice(SCORINGS)
    if mode == "edit" and os.environ.get("AT_FUZZ_EDIT_UNIT"):   # unit mismatch cost: the bit-parallel kernel on DNA (no extra draw: the seeds keep their streams)
        sc = (sc[0], 1) + sc[2:]
    alpha = rng.choice(ALPHABETS)
    uj = (mode == "fit" and rng.random() < 0.5) or mode == "fitj"
    mode = "fit" if mode == "fitj" else mode
    n = rng.choice([1, 2, 3, 7, 16, 33, 64, 100])
    big = rng.random() < 0.06
    hi = 1500 if big else rng.choice([8, 40, 70, 130, 200, 330, 650])   # (650: the 12- to 19-row classes of the 32-lane groups)
    uniform = rng.random() < 0.5
    if big:
        n = min(n, 7)
    elif hi == 650:
        n = min(n, 16) if uniform or rng.random() < 0.5 else 64 + n % 8   # (64 pairs and more: ragged batches of long reads go to the 32-lane frames / the ragged packed overlap)
    if uniform:
        l1 = rng.randint(1, hi)
        l2 = rng.randint(max(l1, 2) if mode == "fit" else 1, max(l1, 2) + hi if mode == "fit" else hi)
        shapes = [(l1, l2)] * n
    else:
        shapes = []
        for _ in range(n):
            l1 = rng.randint(1, hi)
            l2 = rng.randint(max(l1, 2) if mode == "fit" else 1, max(l1, 2) + hi if mode == "fit" else hi)
            shapes.append((l1, l2))
    pairs = []
    for l1, l2 in shapes:
        a = "".join(rng.choice(alpha) for _ in range(l1))
        if rng.random() < 0.5:      # related: long tracebacks, gaps
            t = list(a)
            for _ in range(max(1, len(t) // 12)):
                q = rng.randrange(len(t))
                r = rng.random()
                if r < 0.4:
                    t[q] = rng.choice(alpha)
                elif r < 0.7 and len(t) > 1:
                    del t[q]
                else:
                    t.insert(q, rng.choice(alpha))
            flank = "".join(rng.choice(alpha) for _ in range(l2))
            cut = rng.randrange(0, max(1, l2 - len(t) + 1)) if l2 > len(t) else 0
            b = (flank[:cut] + "".join(t) + flank)[:l2]
        else:
            b = "".join(rng.choice(alpha) for _ in range(l2))
        pairs.append((a, b))
    max_l2 = max(s[1] for s in shapes)
    sites = sorted(rng.sample(range(max_l2 + 3), min(max_l2, rng.choice([0, 1, 3, 8])))) if uj else []
    tb = rng.random() < 0.8
    if os.environ.get("AT_FUZZ_TB") in ("0", "1"):          # a campaign of scores-only (0) or traceback (1) batches; the draw above keeps the seeds' streams
        tb = os.environ["AT_FUZZ_TB"] == "1"
    return mode, sc, uj, sites, pairs, tb


def kernel_class(cfg):
    """The kernel class a batch ran on, from at_last_config: e.g. "packed16x16 8x8 K19 tb", "packed16x4 1x64 K16 ragged tb" (overlap),
    "int32", "myers" -- what a campaign has covered."""
    import re
    if "myers" in cfg:
        m = re.search(r"words/lane=(\d+) (\d+)x(\d+)-lane", cfg)
        return "myers W%s %sx%s" % m.groups() if m else "myers"
    m = re.search(r"packed16 x(\d+) bits=(\d) (\d+)x(\d+)-lane groups.*?rows/lane=(\d+)", cfg)
    if not m:
        return "int32"
    return "packed16x%s %sx%s K%s%s%s%s" % (m.group(1), m.group(3), m.group(4), m.group(5), " ragged" if "ragged frames" in cfg else "",
                                          " hbm-ptr" if "hbm-pointers" in cfg else "", " +sliver" if "32-lane items" in cfg else "")


def run(cases, seed, al=None, verbose=True, classes=None):
    import aligntools.c_amd as A
    own = al is None
    if own:
        al = A.Aligner()
    rng = random.Random(seed)
    done = batches = 0
    while done < cases:
        mode, sc, uj, sites, pairs, tb = draw_batch(rng)
        al.set_scoring(*sc, uj, sites)
        try:
            res = al.align_batch(mode, pairs, traceback=tb, render=False)
        except A.AlignToolsError as ex:
            # the only legitimate refusals: scores leaving the exact range (-3), inputs outside the reference's domain (-4)
            assert ex.code in (-3, -4), (mode, sc, uj, sites, ex)
            if ex.code == -4:
                assert any(O.align(O.MODE_NAMES[mode], a, b, *sc, uj, sites)["rc"] != 0 for a, b in pairs), (mode, sc, pairs[:2])
            batches += 1
            continue
        if classes is not None:
            key = mode + ("j" if uj else "") + ("" if tb or mode == "edit" else " scores-only") + " " + kernel_class(al.last_config)
            classes[key] = classes.get(key, 0) + 1
        for k, (a, b) in enumerate(pairs):
            r = O.align(O.MODE_NAMES[mode], a, b, *sc, uj, sites)
            ctx = (seed, batches, mode, sc, uj, sites, k, a, b, al.last_config)
            assert r["rc"] == 0, ctx
            assert int(res["score"][k]) == r["score"], ctx
            if mode != "edit":
                assert (int(res["end_i"][k]), int(res["end_j"][k]), int(res["state"][k])) == (r["end_i"], r["end_j"], r["state"]), ctx
                if tb:
                    assert res["ops"][k] == r["ops"], ctx
        if tb and mode != "edit" and batches % 4 == 0:
            # the same batch through at_align_batch_strings: the two gapped strings rendered on the GPU (at_render_k) against the
            # oracle's -- the reference's r1 / r2
            st = al.align_batch_strings(mode, pairs)
            for k, (a, b) in enumerate(pairs):
                r = O.align(O.MODE_NAMES[mode], a, b, *sc, uj, sites)
                assert (int(st["score"][k]), st["r1"][k], st["r2"][k]) == (r["score"], r["r1"], r["r2"]), (seed, batches, mode, sc, uj, sites, k, a, b, "strings")
        done += len(pairs)
        batches += 1
    if own:
        al.close()
    if verbose:
        print("fuzz parity: %d cases in %d batches, seed %d: all equal" % (done, batches, seed))
    return done


if __name__ == "__main__":
    os.environ.setdefault("AT_PACKED_MIN_ROUNDS", "0")   # small batches must still reach the 64-lane packed kernels
    seen = {}
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 20000, int(sys.argv[2]) if len(sys.argv) > 2 else 1, classes=seen)
    print("kernel classes:", "; ".join("%s (%d)" % kv for kv in sorted(seen.items())))
