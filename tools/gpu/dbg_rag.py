import random, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import aligntools.c_amd as A
import oracle as O
al = A.Aligner()
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
        try:
            res = al.align_batch(mode, pairs, render=False)
        except Exception as ex:
            print("FAIL", mode, sc, uj, str(ex)[:200], al.last_config)
            import numpy as np
            bp = [(A._b(a), A._b(b)) for a, b in pairs]
            n = len(bp)
            blob, off1, len1, off2, len2 = A._flatten(bp)
            score, ei, ej, st, nops = (np.zeros(n, dtype=np.int32) for _ in range(5))
            ops = np.zeros(len(blob) + 64, dtype=np.uint8)
            rc = al._lib.at_align_batch(al._h, A.MODES[mode], n, A._ptr(blob), A._ptr(off1), A._ptr(len1), A._ptr(off2), A._ptr(len2), 1,
                                        A._ptr(score), A._ptr(ei), A._ptr(ej), A._ptr(st), A._ptr(ops), A._ptr(off1.copy()), A._ptr(nops))
            nbad = 0
            for k in range(n):
                if score[k] == -2**31 or nops[k] < 0:
                    nbad += 1
                    r = O.align(O.MODE_NAMES[mode], pairs[k][0], pairs[k][1], *sc, -7, uj, sites)
                    if nbad < 8:
                        print("   bad pair", k, "l1,l2", len1[k], len2[k], "score", score[k], "nops", nops[k], "end", ei[k], ej[k], "st", st[k], "| oracle", r["score"], r["end_i"], r["end_j"], r["state"])
            print("   nbad", nbad, "of", n)
            continue
        bad = 0
        for k, (s1, s2) in enumerate(pairs):
            r = O.align(O.MODE_NAMES[mode], s1, s2, *sc, -7, uj, sites)
            if (int(res["score"][k]), int(res["end_i"][k]), int(res["end_j"][k]), int(res["state"][k]), res["ops"][k]) != (r["score"], r["end_i"], r["end_j"], r["state"], r["ops"]):
                bad += 1
                if bad < 4:
                    print("  MISMATCH", mode, sc, k, len(s1), len(s2), int(res["score"][k]), r["score"], (int(res["end_i"][k]), int(res["end_j"][k])), (r["end_i"], r["end_j"]))
        print("ok" if not bad else "BAD %d" % bad, mode, sc, uj, al.last_config[:150])
