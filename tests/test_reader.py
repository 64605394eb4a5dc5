"""CPU suite: the product's block-wise gz FASTA / FASTQ reader (aligntools/c_amd/host/fasta.c) against the character-at-a-time
restatement of the reference's record semantics (tests/c/ref_reader.c; kstring_read alignment.h:217-262 over kseq_read
kseq.h:189-229) -- the same records, byte for byte, on hostile, randomised and window-sized inputs."""
import gzip
import os
import random
import subprocess

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def readers(tmp_path_factory):
    d = tmp_path_factory.mktemp("readers")
    inc = os.path.join(ROOT, "include")
    src = os.path.join(ROOT, "tests", "c", "ref_reader.c")
    ref, new = str(d / "ref_reader"), str(d / "new_reader")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-O1", "-I" + inc, src, "-o", ref, "-lz"], check=True)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-O1", "-I" + inc, "-DUSE_PRODUCT_READER", src,
                    os.path.join(ROOT, "aligntools", "c_amd", "host", "fasta.c"), "-o", new, "-lz"], check=True)
    return ref, new


def _both(readers, files, cwd):
    outs = []
    for exe in readers:
        p = subprocess.run([exe] + files, cwd=cwd, capture_output=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append(p.stdout)
    return outs


def test_block_reader_equals_character_reader_on_random_bytes(readers, tmp_path):
    rng = random.Random(20261004)
    alpha = [b">", b"@", b"+", b"\n", b"\n", b"\n", b"\r", b" ", b"\t", b"A", b"C", b"G", b"T", b"ACGT", b"|", b"12", b"name", b"\r\n", b"x",
             b"\x0b", b"\x0c", b"IIII", b"", b"N"]
    files = []
    for i in range(1500):
        data = b"".join(rng.choice(alpha) for _ in range(rng.randint(0, 40)))
        fn = "f%d.fa" % i
        if i % 5 == 0:
            fn += ".gz"
            with gzip.open(tmp_path / fn, "wb") as f:
                f.write(data)
        else:
            (tmp_path / fn).write_bytes(data)
        files.append(fn)
    a, b = _both(readers, files, tmp_path)
    assert a == b and a.count(b" records\n") == len(files)


def test_block_reader_equals_character_reader_on_structured_files(readers, tmp_path):
    """FASTA / FASTQ records with CRLF, blank lines, comments that are inherited, quality blocks that are short, long or
    missing, a missing final newline."""
    rng = random.Random(7)
    files = []
    for i in range(600):
        recs = []
        for _r in range(rng.randint(0, 6)):
            fq = rng.random() < 0.3
            name = b"".join(rng.choice([b"a", b"b", b"1", b"_"]) for _ in range(rng.randint(0, 5)))
            com = rng.choice([b"", b" c|1|2", b"\tx y", b" ", b"\r", b" 100|200\r"])
            seq = b"".join(rng.choice([b"A", b"C", b"G", b"T"]) for _ in range(rng.randint(0, 30)))
            lines, q = [], seq
            while q:
                k = rng.randint(1, 12)
                lines.append(q[:k])
                q = q[k:]
            eol = rng.choice([b"\n", b"\r\n"])
            body = eol.join(lines) + (eol if rng.random() < 0.9 else b"")
            if rng.random() < 0.1:
                body = b"\n\n" + body
            rec = (b"@" if fq else b">") + name + com + eol + body
            if fq:
                ql = max(0, len(seq) + rng.choice([0, 0, 0, 0, -1, 1, 3]))
                qual = b"".join(rng.choice([b"I", b"@", b">", b"+", b"#"]) for _ in range(ql))
                rec += b"+" + rng.choice([b"", name]) + eol + qual + (eol if rng.random() < 0.9 else b"")
            recs.append(rec)
        fn = "s%d.fa" % i
        (tmp_path / fn).write_bytes(b"".join(recs))
        files.append(fn)
    a, b = _both(readers, files, tmp_path)
    assert a == b


def test_block_reader_lines_longer_than_its_window(readers, tmp_path):
    """One line of 12 MB and one of 5 MB (the window is 4 MB and doubles), records that straddle window refills."""
    (tmp_path / "big.fa").write_bytes(b">big c\n" + b"ACGT" * 3000000 + b"\n>two\n" + b"A" * 5000000 + b"\n")
    rng = random.Random(5)
    with open(tmp_path / "many.fa", "wb") as f:      # 9 MB of short records: every refill cuts a record somewhere
        for k in range(60000):
            f.write(b">r%d c%d\n" % (k, k) + bytes(rng.choice(b"ACGT") for _ in range(rng.randint(100, 160))) + b"\n")
    a, b = _both(readers, ["big.fa", "many.fa"], tmp_path)
    assert a == b and b"big.fa: 2 records" in a and b"many.fa: 60000 records" in a


@pytest.mark.skipif(not os.path.isfile("/root/reference/src/kseq.h"), reason="needs /root/reference (build container only)")
def test_both_readers_equal_the_reference_kseq(readers, tmp_path):
    """The ground truth itself: klib's kseq.h as the reference vendors it, compiled in place (tests/c/kseq_harness.c includes it through
    -I/root/reference/src), against the product's block-wise reader and the restatement, on hostile files -- among them the two classes
    the round-3 advisor found: a header that ends at EOF right behind its first whitespace (the comment keeps the previous record's
    text), and a lone '\r' as the file's last byte (kept)."""
    exe = str(tmp_path / "kseq_harness")
    subprocess.run(["gcc", "-std=gnu99", "-O1", "-w", "-I/root/reference/src", os.path.join(ROOT, "tests", "c", "kseq_harness.c"), "-o", exe, "-lz"], check=True)
    rng = random.Random(4711)
    alpha = [b">", b"@", b"+", b"\n", b"\n", b"\n", b"\r", b" ", b"\t", b"A", b"C", b"G", b"T", b"ACGT", b"|", b"12", b"name", b"\r\n", b"x", b"IIII", b"", b"N"]
    files = []
    fixed = [b">a x|y\nACGT\n>b ", b">a x|y\nACGT\n>b\t", b">a\nAC\n\r", b">a c1\nAC\n>b c2\nGT\r", b">a c1\nAC\n>b\nGT\n", b"> ", b">", b">x\n", b"@r c\nACGT\n+\nIIII\n@s \n"]
    for i in range(2500):
        data = fixed[i] if i < len(fixed) else b"".join(rng.choice(alpha) for _ in range(rng.randint(0, 40)))
        fn = "k%d.fa" % i
        (tmp_path / fn).write_bytes(data)
        files.append(fn)
    truth = subprocess.run([exe] + files, cwd=tmp_path, capture_output=True, timeout=600)
    assert truth.returncode == 0, truth.stderr[-2000:]
    a, b = _both(readers, files, tmp_path)
    assert a == truth.stdout, "the restatement (tests/c/ref_reader.c) differs from kseq"
    assert b == truth.stdout, "the product reader (host/fasta.c) differs from kseq"
