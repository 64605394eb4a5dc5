#!/usr/bin/env python3
"""fuzz_walk_kernel.py -- a randomised campaign aimed at the two-pass tracebacks with the walk kernel (at_walk16.hip.h): uniform batches
whose shape lies in the two-pass classes (reads of 129 .. 152 bases on the 8-lane groups, 609 .. 1 024 on the 64-lane group), random
scoring (tie-heavy ones among them), related and unrelated pairs, site lists, odd batch sizes, batches cut into pieces, teams of lanes
on and off -- every batch whole against the one-pass kernels (score, end cell, start state, ops) and a sample against the oracle.

    python3 tests/fuzz_walk_kernel.py [batches] [seed]

Test infrastructure: the oracle is the checker.  (tests/fuzz_parity.py draws its shapes at random and meets these classes rarely.)
"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def run(batches, seed, verbose=True):
    import aligntools.c_amd as A
    import oracle as O
    os.environ.setdefault("AT_PACKED_MIN_ROUNDS", "0")
    os.environ["AT_HOST_CHUNKS"] = "1"
    rng = random.Random(seed)
    al = A.Aligner(0)
    seen = {}
    done = 0
    for b in range(batches):
        wide = rng.random() < 0.4
        mode = rng.choice(["global", "local", "fit", "fit"])
        uj = mode == "fit" and rng.random() < 0.5
        l1 = rng.randint(609, 1024) if wide else rng.randint(129, 152)
        if mode == "fit":
            l2 = l1 + rng.randint(0, 700 if not wide else 400)
        else:
            l2 = rng.randint(max(20, l1 - 100), l1 + (300 if not wide else 200))
        n = rng.choice([1, 2, 3, 17, 129, 255, 257, 700]) if rng.random() < 0.5 else rng.randint(1, 900)
        if rng.random() < 0.3:          # tie-heavy
            sc = (1, -1, -1, -1, rng.choice([-1, -2, -6]))
        else:
            m = rng.randint(1, 3)
            sc = (m, -rng.randint(1, 3), -rng.randint(1, 6), -rng.randint(1, 2), -rng.randint(1, 12))
        sites = sorted(rng.sample(range(1, l2), min(l2 - 1, rng.randint(0, 6)))) if uj else []
        alpha = "ACGTN" if rng.random() < 0.15 else "ACGT"

        def mk(related):
            a = "".join(rng.choice(alpha) for _ in range(l1))
            if not related:
                return a, "".join(rng.choice(alpha) for _ in range(l2))
            t = list(a)
            for _ in range(1 + l1 // rng.choice([8, 15, 40])):
                q = rng.randrange(len(t))
                r = rng.random()
                if r < 0.5:
                    t[q] = rng.choice(alpha)
                elif r < 0.75 and len(t) > 1:
                    del t[q]
                else:
                    t.insert(q, rng.choice(alpha))
            pre = "".join(rng.choice(alpha) for _ in range(rng.randint(0, max(0, l2 - l1))))
            return a, (pre + "".join(t) + "".join(rng.choice(alpha) for _ in range(l2)))[:l2]

        uniq = [mk(rng.random() < 0.6) for _ in range(min(n, 40))]
        pairs = [uniq[k % len(uniq)] for k in range(n)]
        al.set_scoring(*sc, uj, sites)
        os.environ["AT_TWO_PASS"] = "2"
        os.environ["AT_TP_SPLIT"] = "1"
        os.environ["AT_WALK_TEAMS"] = rng.choice(["0", "1"]) if not wide else rng.choice(["1", "1", "0"])
        if rng.random() < 0.25 and n > 64:
            os.environ["AT_CK_PIECE_PAIRS"] = str(rng.choice([32, 64, 128, 300]))
        else:
            os.environ.pop("AT_CK_PIECE_PAIRS", None)
        two = al.align_batch(mode, pairs, traceback=True, render=False)
        cfg = al.last_config
        key = ("walk kernel" if "walk kernel" in cfg else "rounds" if "two-pass" in cfg else cfg.split(" ")[0]) + (" teams" if os.environ["AT_WALK_TEAMS"] == "1" and "walk kernel" in cfg else "") + \
              (" pieces" if "in pieces" in cfg else "") + (" bits=8" if "bits=8" in cfg else "") + (" 64-lane" if "1x64" in cfg else " 8-lane" if "8x8" in cfg else "")
        seen[key] = seen.get(key, 0) + 1
        os.environ["AT_TWO_PASS"] = "0"
        one = al.align_batch(mode, pairs, traceback=True, render=False)
        for f in ("score", "end_i", "end_j", "state", "nops"):
            assert (np.asarray(two[f]) == np.asarray(one[f])).all(), (seed, b, mode, uj, l1, l2, n, sc, sites, f, cfg)
        assert two["ops"] == one["ops"], (seed, b, mode, uj, l1, l2, n, sc, sites, "ops", cfg)
        for k in sorted(set([0, n - 1] + [rng.randrange(n) for _ in range(3 if wide else 8)])):
            r = O.align(O.MODE_NAMES[mode], pairs[k][0], pairs[k][1], *sc, uj, sites)
            assert (int(two["score"][k]), int(two["end_i"][k]), int(two["end_j"][k]), int(two["state"][k]), two["ops"][k]) == \
                   (r["score"], r["end_i"], r["end_j"], r["state"], r["ops"]), (seed, b, mode, uj, l1, l2, n, sc, sites, k, cfg)
        done += n
    for v in ("AT_TWO_PASS", "AT_TP_SPLIT", "AT_WALK_TEAMS", "AT_CK_PIECE_PAIRS"):
        os.environ.pop(v, None)
    al.close()
    if verbose:
        print("fuzz walk kernel: %d alignments in %d batches, seed %d: all equal" % (done, batches, seed))
        print("kernel classes: " + "; ".join("%s (%d)" % kv for kv in sorted(seen.items())))
    return done


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 300, int(sys.argv[2]) if len(sys.argv) > 2 else 6101)
