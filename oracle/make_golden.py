#!/usr/bin/env python3
"""Generate tests/golden/*.jsonl from the REAL reference (oracle/_ref).

TEST INFRASTRUCTURE.  Runs only in the build container (needs /root/reference
to have been compiled by oracle/Makefile into oracle/_ref/libat_ref.so).  The
fixtures are DATA: inputs + the reference's outputs (score and the two gapped
strings, or their md5 when long).  No reference source text is stored.

    python oracle/make_golden.py            # rewrites tests/golden/

Files written
    known_answers.jsonl   the reference's own test/*.fa inputs through every
                          sub-command
    random_small.jsonl    randomized cases, tie-heavy scoring, alphabets of 1-5
                          symbols, ragged lengths incl. the 63/64/65/127/128/129
                          wave-tile edges
    random_dna.jsonl      ACGT-only cases at the BASELINE shapes (150x150,
                          150x500, a few 1024x1024) -- long outputs as md5
    cli.jsonl             stdout/stderr/rc of the stock reference CLI
    known_answers_big.jsonl   the two runs of the reference's own fixtures that need gigabytes in the reference
                          (`fit -s test/tmp.fa`, 1 327 x 114 491: 7.3 GB; `global test/test_fit.fa`,
                          257 x 33 733); written by `python oracle/make_golden.py --only big` (the default run leaves it alone)
"""
import hashlib
import json
import os
import random
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402

REF_TEST = "/root/reference/test"
OUT = os.path.join(ROOT, "tests", "golden")
LONG = 400  # store md5 instead of strings beyond this length


def read_fasta(path):
    recs = []
    with open(path) as fh:
        for line in fh:
            line = line.rstrip("\n").rstrip("\r")
            if line.startswith(">"):
                name, _, comment = line[1:].partition(" ")
                recs.append([name, comment, ""])
            elif recs:
                recs[-1][2] += line
    return recs


def md5(s):
    return hashlib.md5(s.encode("latin1")).hexdigest()


def case(mode, s1, s2, m, u, o, e, j=-10, use_jump=False, sites=None, tag="", keep_inputs=True):
    r = O.ref_align(O.MODE_NAMES[mode], s1, s2, m, u, o, e, j, use_jump, sites)
    assert r["rc"] == 0, (mode, len(s1), len(s2))
    d = dict(mode=mode, m=m, u=u, o=o, e=e, j=j, use_jump=bool(use_jump), sites=list(sites or []),
             s1=s1, s2=s2, score=r["score"], tag=tag)
    if mode != "edit":
        if len(r["r1"]) > LONG:
            d["rlen"] = len(r["r1"])
            d["r1_md5"] = md5(r["r1"])
            d["r2_md5"] = md5(r["r2"])
        else:
            d["r1"] = r["r1"]
            d["r2"] = r["r2"]
    return d


def mutate(rng, s, alpha, sub=0.05, ins=0.02, dele=0.02):
    out = []
    for ch in s:
        x = rng.random()
        if x < dele:
            continue
        if x < dele + sub:
            out.append(rng.choice(alpha))
        else:
            out.append(ch)
        if rng.random() < ins:
            out.append(rng.choice(alpha))
    return "".join(out)


SCORINGS = [(1, -1, -1, -1), (2, -2, -5, -2), (1, -2, -5, -1), (1, -1, -4, -1), (3, -1, -2, -2),
            (2, -3, 0, -1), (1, -1, 1, -1), (0, 0, 0, 0), (2, -2, -3, 1), (5, -4, -10, -1)]


def known_answers():
    loc = read_fasta(f"{REF_TEST}/test_local.fa")
    glo = read_fasta(f"{REF_TEST}/test_global.fa")
    edi = read_fasta(f"{REF_TEST}/test_edit.fa")
    fit = read_fasta(f"{REF_TEST}/test_fit.fa")
    sites = [int(x) for x in fit[1][1].split("|")]
    out = []
    for name, recs in (("test_local", loc), ("test_global", glo), ("test_edit", edi), ("test_fit", fit)):
        s1, s2 = recs[0][2], recs[1][2]
        for mode, args in (("local", (2, -2, -5, -2)), ("local", (1, -2, -5, -1)),
                           ("global", (1, -1, -4, -1)), ("global", (1, -2, -5, -1)),
                           ("overlap", (1, -2, -5, -1)),
                           ("edit", (1, 1, 2, -1)), ("edit", (1, -2, -5, -1))):
            if name == "test_fit" and mode in ("global", "overlap", "edit"):
                if mode == "global":
                    continue  # 257 x 33733 global: 70 MB of doubles per matrix, skip
            out.append(case(mode, s1, s2, *args, tag=name))
        if len(s1) <= len(s2):
            out.append(case("fit", s1, s2, 2, -2, -5, -1, tag=name))
            out.append(case("fit", s1, s2, 1, -2, -5, -1, tag=name))
    s1, s2 = fit[0][2], fit[1][2]
    out.append(case("fit", s1, s2, 2, -2, -5, -1, -10, True, sites, tag="test_fit -s (README.md:82)"))
    out.append(case("fit", s1, s2, 1, -2, -5, -1, -10, True, sites, tag="test_fit -s defaults"))
    out.append(case("fit", s1, s2, 2, -2, -5, -1, -10, True, [100], tag="test_fit -s, site list [100]"))
    return out


def known_answers_big():
    """The runs of the reference's own inputs that take gigabytes there (`fit -s test/tmp.fa` is
    1 327 x 114 491 = 7.3 GB of matrices, score 1327 and two 1 327-character strings)."""
    out = []
    tmp = read_fasta(f"{REF_TEST}/tmp.fa")
    sites = [int(x) for x in tmp[1][1].split("|")]
    s1, s2 = tmp[0][2], tmp[1][2]
    out.append(case("fit", s1, s2, 1, -2, -5, -1, -10, True, sites, tag="tmp.fa fit -s (defaults)"))
    assert out[-1]["score"] == 1327 and out[-1]["rlen"] == 1327, (out[-1]["score"], out[-1].get("rlen"))
    out.append(case("fit", s1, s2, 2, -2, -5, -1, -10, True, sites, tag="tmp.fa fit -s -m 2"))
    fit = read_fasta(f"{REF_TEST}/test_fit.fa")
    out.append(case("global", fit[0][2], fit[1][2], 1, -2, -5, -1, tag="test_fit global (defaults)"))
    out.append(case("global", fit[0][2], fit[1][2], 1, -1, -4, -1, tag="test_fit global -m 1 -u -1 -o -4 -e -1"))
    return out


def random_small(rng, n):
    out = []
    edges = [1, 2, 3, 7, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 150]
    modes = ["global", "local", "fit", "overlap", "edit"]
    for it in range(n):
        mode = modes[it % 5]
        k = rng.choice([1, 2, 3, 4, 4, 4, 5])
        alpha = "ACGTN"[:k]
        if rng.random() < 0.1:
            alpha = "acgtXYZ*"
        if rng.random() < 0.35:
            l1 = rng.choice(edges)
            l2 = rng.choice(edges + [200, 257, 300, 500])
        else:
            l1 = rng.randint(1, 90)
            l2 = rng.randint(1, 140)
        s1 = "".join(rng.choice(alpha) for _ in range(l1))
        style = rng.random()
        if style < 0.45:
            s2 = "".join(rng.choice(alpha) for _ in range(l2))
        elif style < 0.8:
            s2 = ("".join(rng.choice(alpha) for _ in range(rng.randint(0, 12))) + mutate(rng, s1, alpha, 0.08, 0.04, 0.04) +
                  "".join(rng.choice(alpha) for _ in range(rng.randint(0, 12))))
        else:  # overlap-shaped: suffix of s1 = prefix of s2
            cut = rng.randint(0, len(s1))
            s2 = mutate(rng, s1[cut:], alpha, 0.03, 0.01, 0.01) + "".join(rng.choice(alpha) for _ in range(rng.randint(1, 40)))
        if not s2:
            s2 = rng.choice(alpha)
        if mode == "fit":
            if len(s2) < 2:
                s2 += rng.choice(alpha) * 2
            if len(s1) > len(s2):
                s1, s2 = s2, s1
        sc = rng.choice(SCORINGS)
        j = rng.choice([-10, -10, -3, -1, 0])
        use_jump = mode == "fit" and rng.random() < 0.6
        sites = sorted(set(rng.randint(0, len(s2)) for _ in range(rng.randint(0, 6)))) if use_jump else []
        try:
            out.append(case(mode, s1, s2, *sc, j, use_jump, sites, tag="rand"))
        except AssertionError:
            pass
    # fit with a real intron: read = exon1 + exon2, contig = exon1 + intron + exon2
    for it in range(60):
        alpha = "ACGT"
        e1 = "".join(rng.choice(alpha) for _ in range(rng.randint(10, 60)))
        e2 = "".join(rng.choice(alpha) for _ in range(rng.randint(10, 60)))
        intron = "".join(rng.choice(alpha) for _ in range(rng.randint(20, 200)))
        pre = "".join(rng.choice(alpha) for _ in range(rng.randint(0, 30)))
        post = "".join(rng.choice(alpha) for _ in range(rng.randint(1, 30)))
        read = mutate(rng, e1 + e2, alpha, 0.02, 0.01, 0.01) or "A"
        contig = pre + e1 + intron + e2 + post
        jcol = len(pre) + len(e1)
        for sites in ([], [jcol], [jcol - 1], [jcol + 1], [jcol - 1, jcol, jcol + 1], [len(contig) + 5]):
            out.append(case("fit", read, contig, 2, -2, -5, -1, rng.choice([-10, -4]), True, sites, tag="intron"))
    return out


def random_dna(rng):
    out = []
    alpha = "ACGT"

    def dna(n):
        return "".join(rng.choice(alpha) for _ in range(n))
    for it in range(60):   # C2 shape
        s1 = dna(150)
        s2 = dna(150) if it % 2 == 0 else (dna(rng.randint(0, 20)) + mutate(rng, s1, alpha))[:150].ljust(150, "A")
        out.append(case("local", s1, s2, 2, -2, -5, -2, tag="C2"))
        out.append(case("global", s1, s2, 1, -1, -4, -1, tag="C2g"))
        out.append(case("overlap", s1, s2, 1, -2, -5, -1, tag="C2o"))
        out.append(case("edit", s1, s2, 1, 1, -5, -1, tag="C2e"))
    for it in range(40):   # C4 shape: 150 read vs 500 contig, sites 100|200|300|400
        contig = dna(500)
        if it % 2 == 0:
            a = rng.randint(0, 340)
            read = mutate(rng, contig[a:a + 150], alpha)[:150]
            if it % 4 == 0:   # spliced read across a listed junction
                read = (contig[40:100] + contig[200:290])[:150]
                read = mutate(rng, read, alpha, 0.02, 0.0, 0.0)
        else:
            read = dna(150)
        out.append(case("fit", read, contig, 2, -2, -5, -1, -10, True, [100, 200, 300, 400], tag="C4"))
        out.append(case("fit", read, contig, 2, -2, -5, -1, -10, False, [], tag="C4 no jump"))
    for it in range(3):    # C3 shape, md5 only
        s1 = dna(1024)
        s2 = dna(1024) if it == 0 else (mutate(rng, s1, alpha) + dna(64))[:1024]
        out.append(case("global", s1, s2, 1, -1, -4, -1, tag="C3"))
        out.append(case("local", s1, s2, 2, -2, -5, -2, tag="C3l"))
    for it in range(3):    # C5 shape
        a = dna(1000)
        b = (a[rng.randint(300, 900):] + dna(1000))[:1000]
        out.append(case("overlap", a, b, 1, -2, -5, -1, tag="C5"))
        out.append(case("overlap", dna(1000), dna(1000), 1, -2, -5, -1, tag="C5r"))
    return out


def cli_cases():
    """Byte-exact behaviour of the stock reference CLI (stdout, stderr, rc)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "alignTools")
    out = []
    runs = [
        ["local", "-m", "2", "-u", "-2", "-o", "-5", "-e", "-2", "test/test_local.fa"],
        ["local", "test/test_local.fa"],
        ["global", "-m", "1", "-u", "-1", "-o", "-4", "-e", "-1", "test/test_global.fa"],
        ["global", "test/test_global.fa"],
        ["global", "test/test_local.fa"],
        ["local", "test/test_edit.fa"],
        ["fit", "-m", "2", "-u", "-2", "-s", "test/test_fit.fa"],
        ["overlap", "test/test_global.fa"],
        ["overlap", "test/test_local.fa"],
        ["overlap", "-m", "2", "test/test_global.fa"],
        ["edit", "-u", "1", "-o", "2", "test/test_edit.fa"],
        ["edit", "test/test_edit.fa"],
        ["edit", "-u", "1", "test/test_global.fa"],
        [], ["foo"], ["local"], ["global"], ["fit"], ["overlap"], ["edit"],
        ["local", "-x", "test/test_local.fa"], ["local", "-j", "3", "test/test_local.fa"],
        ["local", "nofile.fa"], ["fit", "test/test_global.fa", ],
        ["fit", "-s", "test/test_local.fa"],
    ]
    for argv in runs:
        p = subprocess.run([exe] + argv, cwd="/root/reference", capture_output=True)
        so = p.stdout.decode("latin1")
        d = dict(argv=argv, rc=p.returncode, stderr=p.stderr.decode("latin1").replace(exe, "alignTools"))
        if len(so) > 2000:
            d["stdout_md5"] = md5(so)
            d["stdout_len"] = len(so)
        else:
            d["stdout"] = so
        out.append(d)
    return out


def dump(name, rows):
    path = os.path.join(OUT, name)
    with open(path, "w") as fh:
        for r in rows:
            fh.write(json.dumps(r, sort_keys=True) + "\n")
    print(f"{name}: {len(rows)} cases, {os.path.getsize(path)} bytes")


def main():
    assert O.have_ref(), "build oracle/_ref first (make -C oracle)"
    os.makedirs(OUT, exist_ok=True)
    if sys.argv[1:3] == ["--only", "big"]:
        dump("known_answers_big.jsonl", known_answers_big())
        return
    dump("known_answers.jsonl", known_answers())
    dump("random_small.jsonl", random_small(random.Random(20261003), 1500))
    dump("random_dna.jsonl", random_dna(random.Random(7)))
    dump("cli.jsonl", cli_cases())


if __name__ == "__main__":
    main()
