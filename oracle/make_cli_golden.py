#!/usr/bin/env python3
"""Record the stock reference CLI (oracle/_ref/alignTools, built by oracle/Makefile from /root/reference) on
synthetic INPUT FILES that exercise its reader -- multi-line and CRLF FASTA, FASTQ, gzip, lower case, comments,
site lists -- and write tests/golden/cli_files.jsonl: file bytes (base64), argv, stdout, stderr, rc.
Not recorded: cases on which the reference dies of a signal, cases whose output changes with the heap fill pattern
(MALLOC_PERTURB_) or whose two strings differ in length (the one-byte overrun of strrev, alignment.h:176-183, shows up
as a stray byte behind r1 -- here the size field of the next heap block), and
local on an empty sequence (uninitialised end cell, SURVEY.md 8a "input domain").  Test infrastructure."""
import base64
import gzip
import json
import os
import random
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "alignTools")
OUT = os.path.join(ROOT, "tests", "golden", "cli_files.jsonl")


def wrap(s, w):
    return "\n".join(s[k:k + w] for k in range(0, len(s), w))


def files(rng):
    dna = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    a, b = dna(150), dna(150)
    rel = a[20:90] + dna(40) + a[95:]
    prot1, prot2 = "MKVLAAGIVGLLLAQWEE", "KVLAGIVGLALAQW"
    contig = dna(400)
    read = contig[120:180] + contig[260:330]          # a spliced read: jump between two blocks
    out = []
    out.append(("plain.fa", (">s1\n%s\n>s2\n%s\n" % (a, rel)).encode()))
    out.append(("multiline.fa", (">s1 some comment\n%s\n>s2\n%s\n" % (wrap(a, 60), wrap(rel, 37))).encode()))
    out.append(("crlf.fa", (">s1\r\n%s\r\n>s2\r\n%s\r\n" % (wrap(a, 50).replace("\n", "\r\n"), rel)).encode()))
    out.append(("noeol.fa", (">s1\n%s\n>s2\n%s" % (a, rel)).encode()))
    out.append(("blank_lines.fa", ("\n\n>s1\n%s\n\n%s\n\n>s2\n\n%s\n\n" % (a[:70], a[70:], rel)).encode()))
    out.append(("lower.fa", (">s1\n%s\n>s2\n%s\n" % (a.lower(), rel)).encode()))
    out.append(("mixedcase.fa", (">s1\n%s\n>s2\n%s\n" % (a[:75] + a[75:].lower(), rel[:60].lower() + rel[60:])).encode()))
    out.append(("withN.fa", (">s1\n%s\n>s2\n%s\n" % (a[:40] + "NNNNN" + a[45:], rel[:30] + "N" + rel[31:])).encode()))
    out.append(("protein.fa", (">p1\n%s\n>p2\n%s\n" % (prot1, prot2)).encode()))
    out.append(("tiny.fa", b">x\nA\n>y\nAC\n"))
    out.append(("spaces.fa", (">s1\n%s  \n %s\n>s2\n%s\t\n" % (a[:70], a[70:], rel)).encode()))
    out.append(("reads.fq", ("@r1 c\n%s\n+\n%s\n@r2\n%s\n+r2\n%s\n" % (a, "I" * len(a), rel, "#" * len(rel))).encode()))
    out.append(("reads_ml.fq", ("@r1\n%s\n%s\n+\n%s\n%s\n@r2\n%s\n+\n%s\n" % (a[:80], a[80:], "I" * 80, "I" * 70, rel, "5" * len(rel))).encode()))
    out.append(("plain.fa.gz", gzip.compress((">s1\n%s\n>s2\n%s\n" % (a, rel)).encode(), mtime=0)))
    out.append(("reads.fq.gz", gzip.compress(("@r1\n%s\n+\n%s\n@r2\n%s\n+\n%s\n" % (a, "I" * len(a), rel, "I" * len(rel))).encode(), mtime=0)))
    out.append(("fit.fa", (">read\n%s\n>contig\n%s\n" % (read, wrap(contig, 80))).encode()))
    out.append(("fit_sites.fa", (">read\n%s\n>contig 180|260\n%s\n" % (read, wrap(contig, 80))).encode()))
    out.append(("fit_sites_tab.fa", (">read\n%s\n>contig\t179|259|300\n%s\n" % (read, contig)).encode()))
    out.append(("fit_sites_one.fa", (">read\n%s\n>contig 180\n%s\n" % (read, contig)).encode()))
    out.append(("fit_sites_odd.fa", (">read\n%s\n>contig 180|abc|260|\n%s\n" % (read, contig)).encode()))
    out.append(("fit_nosites.fa", (">read\n%s\n>contig\n%s\n" % (read, contig)).encode()))
    out.append(("long_names.fa", (">%s desc %s\n%s\n>%s\n%s\n" % ("n" * 300, "d" * 500, a, "m" * 1000, rel)).encode()))
    out.append(("three.fa", (">a\n%s\n>b\n%s\n>c\n%s\n" % (a[:30], a[30:60], a[60:90])).encode()))
    out.append(("one.fa", (">a\n%s\n" % a).encode()))
    out.append(("empty.fa", b""))
    out.append(("empty_second.fa", (">a\n%s\n>b\n" % a[:20]).encode()))
    return out


RUNS = {
    "default": [["local"], ["global"], ["overlap"], ["edit"], ["local", "-m", "2", "-u", "-2", "-o", "-5", "-e", "-2"],
                ["global", "-m", "1", "-u", "-1", "-o", "-4", "-e", "-1"], ["edit", "-u", "1"]],
    "fit": [["fit"], ["fit", "-m", "2", "-u", "-2"], ["fit", "-s"], ["fit", "-s", "-j", "-4"], ["fit", "-m", "2", "-u", "-2", "-o", "-5", "-e", "-1", "-j", "-10", "-s"]],
    "few": [["local"], ["edit"]],
}


def main():
    assert os.path.exists(EXE), "build oracle/_ref first (make -C oracle)"
    rng = random.Random(20261004)
    rows, skipped = [], 0
    with tempfile.TemporaryDirectory() as d:
        for name, data in files(rng):
            with open(os.path.join(d, name), "wb") as fh:
                fh.write(data)
            kind = "fit" if name.startswith("fit") else ("few" if name in ("three.fa", "one.fa", "empty.fa", "empty_second.fa", "long_names.fa") else "default")
            for argv in RUNS[kind]:
                if argv[0] == "overlap":
                    full = argv[:1] + [name] + argv[1:]      # overlap reads argv[1] (alignment.h:994)
                else:
                    full = argv + [name]
                if name == "empty_second.fa" and argv[0] == "local":
                    skipped += 1                              # outside the domain on which the reference is defined
                    continue
                p = subprocess.run([EXE] + full, cwd=d, capture_output=True)
                if p.returncode < 0:
                    skipped += 1                              # the reference crashed: undefined behaviour, no golden
                    continue
                stable = True
                for fill in ("85", "170"):                    # does the output depend on what lies behind a heap block?
                    q = subprocess.run([EXE] + full, cwd=d, capture_output=True, env=dict(os.environ, MALLOC_PERTURB_=fill))
                    stable = stable and (q.returncode, q.stdout, q.stderr) == (p.returncode, p.stdout, p.stderr)
                lines = p.stdout.split(b"\n")
                if p.returncode == 0 and argv[0] != "edit" and len(lines) >= 3 and len(lines[-3]) != len(lines[-2]):
                    stable = False                            # a stray byte behind r1: strrev read past its buffer
                if not stable:
                    skipped += 1
                    continue
                rows.append(dict(file=name, data=base64.b64encode(data).decode(), argv=full, rc=p.returncode,
                                 stdout=p.stdout.decode("latin1"), stderr=p.stderr.decode("latin1").replace(EXE, "alignTools")))
    with open(OUT, "w") as fh:
        for r in rows:
            fh.write(json.dumps(r, sort_keys=True) + "\n")
    print("%s: %d cases (%d undefined-behaviour cases skipped), %d bytes" % (OUT, len(rows), skipped, os.path.getsize(OUT)))


if __name__ == "__main__":
    sys.exit(main())
