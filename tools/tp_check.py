#!/usr/bin/env python3
"""tp_check.py -- two-pass tracebacks (AT_TWO_PASS=1, CK kernels) against the one-pass kernels (AT_TWO_PASS=0) and the oracle on
uniform batches of the BASELINE shapes: every score / end cell / state / ops string of the batch equal, a sample against the oracle.

    python3 tools/tp_check.py [pairs-scale]
"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import aligntools.c_amd as A
import oracle as O

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
rng = random.Random(int(os.environ.get("TP_SEED", "77")))
al = A.Aligner(0)


def mk(l1, l2, related, alpha="ACGT"):
    a = "".join(rng.choice(alpha) for _ in range(l1))
    if related:
        t = list(a)
        for _ in range(1 + l1 // 15):
            q = rng.randrange(len(t))
            r = rng.random()
            if r < 0.5:
                t[q] = rng.choice(alpha)
            elif r < 0.75 and len(t) > 1:
                del t[q]
            else:
                t.insert(q, rng.choice(alpha))
        b = ("".join(rng.choice(alpha) for _ in range(rng.randint(0, max(0, l2 - l1)))) + "".join(t) + "".join(rng.choice(alpha) for _ in range(l2)))[:l2]
    else:
        b = "".join(rng.choice(alpha) for _ in range(l2))
    return a, b


CASES = [
    # mode, use_jump, l1, l2, pairs, scoring, sites
    ("local", False, 150, 150, 40000, (2, -2, -5, -2, -10), []),
    ("global", False, 150, 150, 40000, (1, -1, -4, -1, -10), []),
    ("fit", False, 150, 500, 20000, (2, -2, -5, -1, -10), []),
    ("fit", True, 150, 500, 20000, (2, -2, -5, -1, -10), [100, 200, 300, 400]),
    ("global", False, 1024, 1024, 7000, (1, -1, -4, -1, -10), []),
    ("local", False, 1000, 1024, 7000, (1, -1, -4, -1, -10), []),
    ("fit", False, 700, 1024, 7000, (1, -1, -4, -1, -10), []),
    ("local", False, 1000, 1024, 7000, (2, -2, -5, -2, -10), []),          # scores x4
    ("fit", True, 620, 660, 7000, (1, -1, -4, -1, -6), [100, 450, 451, 600]),   # scores x16, jump state, 64-lane group
    ("local", False, 129, 140, 30000, (1, -1, -1, -1, -10), []),      # tie-heavy
    ("fit", True, 140, 300, 30000, (1, -1, -1, -1, -2), [7, 50, 51, 120]),
]
bad = 0
for mode, uj, l1, l2, n, sc, sites in CASES:
    n = max(64, int(n * scale))
    uniq = [mk(l1, l2, k % 2 == 1) for k in range(min(n, 600))]
    pairs = (uniq * (n // len(uniq) + 1))[:n]
    al.set_scoring(*sc, uj, sites)
    out = {}
    for tp in (os.environ.get("TP_MODE", "2"), "0"):
        os.environ["AT_TWO_PASS"] = tp
        t0 = time.time()
        out[tp] = al.align_batch(mode, pairs, traceback=True, render=False)
        dt = time.time() - t0
        print("%-6s%s %4dx%-4d n=%-6d two_pass=%s %.2fs  %s" % (mode, " -s" if uj else "", l1, l2, n, tp, dt, al.last_config[:150]), flush=True)
    a, b = out[os.environ.get("TP_MODE", "2")], out["0"]
    for key in ("score", "end_i", "end_j", "state"):
        d = np.nonzero(np.asarray(a[key]) != np.asarray(b[key]))[0]
        if len(d):
            bad += 1
            print("  MISMATCH %s at %d pairs, first %s: %s vs %s" % (key, len(d), d[:5], np.asarray(a[key])[d[:5]], np.asarray(b[key])[d[:5]]))
    dops = [k for k in range(n) if a["ops"][k] != b["ops"][k]]
    if dops:
        bad += 1
        k = dops[0]
        print("  MISMATCH ops at %d pairs, first %d (uniq %d): len %d vs %d" % (len(dops), k, k % len(uniq), len(a["ops"][k]), len(b["ops"][k])))
        x, y = a["ops"][k], b["ops"][k]
        q = next((i for i in range(min(len(x), len(y))) if x[i] != y[i]), min(len(x), len(y)))
        print("   first differing op %d: %r vs %r ; end cell %d,%d" % (q, list(x[max(0, q - 4):q + 6]), list(y[max(0, q - 4):q + 6]), a["end_i"][k], a["end_j"][k]))
    for k in range(0, min(n, len(uniq)), 7):
        r = O.align(O.MODE_NAMES[mode], pairs[k][0], pairs[k][1], *sc, uj, sites)
        if (int(a["score"][k]), a["ops"][k]) != (r["score"], r["ops"]):
            bad += 1
            print("  ORACLE MISMATCH pair %d: score %d vs %d, ops equal %s" % (k, a["score"][k], r["score"], a["ops"][k] == r["ops"]))
            break
os.environ.pop("AT_TWO_PASS", None)
print("tp_check: %s" % ("OK" if not bad else "%d FAILURES" % bad))
sys.exit(1 if bad else 0)
