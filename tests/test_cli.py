"""The C host: `alignTools <cmd> [opts] <target.fa>` byte-for-byte against what the stock
reference CLI printed (tests/golden/cli.jsonl: stdout, stderr, return code), and the
gz-FASTA reader against the record semantics of the reference's kstring_read/kseq_read.
Error paths run anywhere; commands that align need the GPU (-m gpu)."""
import base64
import ctypes as C
import gzip
import hashlib
import os
import random
import subprocess

import pytest

from conftest import ROOT, load_golden

EXE = os.path.join(ROOT, "aligntools", "c_amd", "bin", "alignTools")
HOSTLIB = os.path.join(ROOT, "aligntools", "c_amd", "libaligntools.so")


@pytest.fixture(scope="module")
def built():
    from aligntools.c_amd import build
    build.build()
    assert os.path.exists(EXE)
    return EXE


@pytest.fixture(scope="module")
def ref_inputs(tmp_path_factory):
    """The reference's test/*.fa inputs, re-created from the golden fixtures (sequences + site comment)."""
    d = tmp_path_factory.mktemp("cli")
    os.makedirs(d / "test")
    seen = {}
    for c in load_golden("known_answers.jsonl"):
        tag = c["tag"].split(" ")[0]
        if tag in seen:
            continue
        seen[tag] = (c["s1"], c["s2"])
    comment = {"test_fit": " 1036|3395|23045|24611"}
    for tag, (s1, s2) in seen.items():
        with open(d / "test" / (tag + ".fa"), "w") as fh:
            fh.write(">first\n")
            for k in range(0, len(s1), 70):
                fh.write(s1[k:k + 70] + "\n")
            fh.write(">second%s\n%s\n" % (comment.get(tag, ""), s2))
    return d


def _run(argv, cwd):
    p = subprocess.run([EXE] + argv, cwd=cwd, capture_output=True)
    return p.returncode, p.stdout.decode("latin1"), p.stderr.decode("latin1").replace(EXE, "alignTools")


def _cases(need_gpu):
    out = []
    for c in load_golden("cli.jsonl"):
        aligns = c["rc"] == 0 or c["argv"][:1] == ["fit"] and c["argv"][-1].endswith("test_global.fa")
        if aligns == need_gpu:
            out.append(c)
    return out


@pytest.mark.parametrize("case", _cases(False), ids=lambda c: " ".join(c["argv"]) or "noargs")
def test_cli_error_paths_match_reference(built, ref_inputs, case):
    rc, so, se = _run(case["argv"], ref_inputs)
    assert rc == case["rc"]
    assert so == case.get("stdout", "")
    assert se == case["stderr"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", _cases(True), ids=lambda c: " ".join(c["argv"]))
def test_cli_outputs_match_reference(built, ref_inputs, case):
    rc, so, se = _run(case["argv"], ref_inputs)
    assert rc == case["rc"], se
    if "stdout" in case:
        assert so == case["stdout"]
    else:
        assert len(so) == case["stdout_len"] and hashlib.md5(so.encode("latin1")).hexdigest() == case["stdout_md5"]
    assert se == case["stderr"]


@pytest.mark.gpu
def test_cli_edit_rejects_dash_e_instead_of_crashing(built, ref_inputs):
    """The reference declares `-e` without an argument for `edit` and calls atoi(NULL) (SIGSEGV); we return 1."""
    rc, so, se = _run(["edit", "-e", "test/test_edit.fa"], ref_inputs)
    assert rc == 1 and so == ""


@pytest.mark.gpu
def test_cli_batch_extension(built, tmp_path):
    with open(tmp_path / "pairs.fa", "w") as fh:
        fh.write(">a1\nPLEASANTLY\n>a2\nMEANLY\n>b1\nACGTACGT\n>b2\nACGTTACGT\n")
    p = subprocess.run([EXE, "batch", "local", "-m", "2", "-u", "-2", "-o", "-5", "-e", "-2", "pairs.fa"], cwd=tmp_path,
                       capture_output=True)
    assert p.returncode == 0, p.stderr
    lines = p.stdout.decode().splitlines()
    assert lines[0] == "a1\ta2\tscore=4.000000" and lines[1:3] == ["LEA", "MEA"]
    assert lines[3].startswith("b1\tb2\tscore=")


def _reads_file(path, n, seed):
    rng = random.Random(seed)
    base = "".join(rng.choice("ACGT") for _ in range(400))
    with open(path, "w") as fh:
        for k in range(n):
            a = rng.randint(0, 200)
            s = list(base[a:a + rng.randint(60, 150)])
            for _ in range(4):
                s[rng.randrange(len(s))] = rng.choice("ACGT")
            fh.write(">r%d some comment\n%s\n" % (k, "".join(s)))


@pytest.mark.gpu
def test_cli_batch_score_only_and_all_vs_all(built, tmp_path):
    """`--score-only` drops the strings, `--all-vs-all` aligns every ordered pair a < b of the records (enumerated on the GPU
    when only scores are wanted): lines, order and scores equal the oracle's."""
    import oracle as O
    _reads_file(tmp_path / "reads.fa", 12, 3)
    recs = []
    for line in open(tmp_path / "reads.fa"):
        if line.startswith(">"):
            recs.append([line[1:].split()[0], ""])
        else:
            recs[-1][1] += line.strip()
    for mode, extra in (("overlap", []), ("local", ["-m", "2", "-u", "-2", "-o", "-5", "-e", "-2"]), ("global", []), ("edit", ["-u", "1"])):
        sc = (2, -2, -5, -2) if mode == "local" else (1, 1, -5, -1) if mode == "edit" else (1, -2, -5, -1)
        for tb in (False, True):
            if mode == "edit" and tb:
                continue
            argv = [EXE, "batch", mode] + extra + ["--all-vs-all"] + ([] if tb else ["--score-only"]) + ["reads.fa"]
            p = subprocess.run(argv, cwd=tmp_path, capture_output=True)
            assert p.returncode == 0, p.stderr
            lines = p.stdout.decode().splitlines()
            q = 0
            for a in range(len(recs)):
                for b in range(a + 1, len(recs)):
                    r = O.align(O.MODE_NAMES[mode], recs[a][1], recs[b][1], *sc)
                    want = "%s\t%s\t%s" % (recs[a][0], recs[b][0], ("edit_distance=%d" % r["score"]) if mode == "edit" else "score=%f" % r["score"])
                    assert lines[q] == want, (mode, tb, a, b, lines[q])
                    if tb:
                        assert lines[q + 1:q + 3] == [r["r1"], r["r2"]], (mode, a, b)
                    q += 3 if tb else 1
            assert q == len(lines)
    # pair lists: --score-only
    p = subprocess.run([EXE, "batch", "overlap", "--score-only", "reads.fa"], cwd=tmp_path, capture_output=True)
    assert p.returncode == 0 and len(p.stdout.decode().splitlines()) == 6
    p = subprocess.run([EXE, "batch", "fit", "--all-vs-all", "reads.fa"], cwd=tmp_path, capture_output=True)
    assert p.returncode == 255 and b"fit needs ordered pairs" in p.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [[], ["--score-only"], ["--all-vs-all", "--score-only"]])
def test_cli_batch_gpus_n_matches_one_gpu(built, tmp_path, flags, fake_rccl):
    """`--gpus N`: one process per GPU, rank 0's options broadcast, contiguous shares of the pairs, results gathered and
    printed by rank 0.  The test box has one card, so N = 2 and N = 3 are rehearsed with the ranks sharing it and the
    collectives going through a stand-in for librccl.so (AT_RCCL_LIB = tests/c/fake_rccl_files.c, files in the rendezvous
    directory: RCCL itself refuses two ranks on one device -- the product library holds no such transport); N = 1
    needs no collective.  Output and return code must equal the single-process run byte for byte."""
    _reads_file(tmp_path / "reads.fa", 14, 5)
    base = [EXE, "batch", "local", "-m", "2", "-u", "-2", "-o", "-5", "-e", "-2"] + flags
    one = subprocess.run(base + ["reads.fa"], cwd=tmp_path, capture_output=True)
    assert one.returncode == 0, one.stderr
    env = dict(os.environ, AT_RCCL_LIB=fake_rccl, AT_ONE_DEVICE="1")
    for n in (1, 2, 3):
        p = subprocess.run(base + ["--gpus", str(n), "reads.fa"], cwd=tmp_path, capture_output=True, env=env, timeout=600)
        assert p.returncode == 0, (n, p.stderr[-2000:])
        assert p.stdout == one.stdout, (n, flags)
        assert p.stderr.decode().count("[main] CMD:") == 1      # only rank 0 signs off
    # the RCCL branch itself (dlopen, ncclCommInitRank, ncclBroadcast, ncclAllGather) with a world of one rank on this card
    rccl = subprocess.run(base + ["--gpus", "1", "reads.fa"], cwd=tmp_path, capture_output=True, timeout=600,
                          env=dict(os.environ, AT_COMM_FORCE_RCCL="1", HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert rccl.returncode == 0, rccl.stderr[-2000:]
    assert rccl.stdout == one.stdout, flags
    # a failing rank ends the whole job with its code: fit with a read longer than its contig
    with open(tmp_path / "bad.fa", "w") as fh:
        fh.write(">a\nACGTACGTAC\n>b\nACGT\n>c\nAC\n>d\nACGT\n")
    p = subprocess.run([EXE, "batch", "fit", "--gpus", "2", "bad.fa"], cwd=tmp_path, capture_output=True, env=env, timeout=600)
    assert p.returncode == 255 and b"first sequence must be shorter" in p.stderr


@pytest.mark.gpu
def test_cli_batch_streams_chunks_and_scales_all_vs_all(built, tmp_path):
    """The batch driver parses, aligns and prints chunk by chunk: a run cut into chunks of 64 pairs prints the bytes of a run in
    one chunk, with and without tracebacks.  And --all-vs-all --score-only over 3 000 reads (4.5 M pairs, slices streamed from
    at_align_allpairs_stream, the triangle walked incrementally on the host) finishes in seconds -- its per-pair host work used
    to be O(reads) -- with the oracle's scores on sampled lines."""
    import time
    import oracle as O
    rng = random.Random(11)
    with open(tmp_path / "pairs.fa", "w") as fh:
        for k in range(1000):
            a = "".join(rng.choice("ACGT") for _ in range(rng.randint(30, 120)))
            b = "".join(rng.choice("ACGT") for _ in range(rng.randint(30, 120)))
            fh.write(">p%da c\n%s\n>p%db\n%s\n" % (k, a, k, b))
    for flags in ([], ["--score-only"]):
        base = [EXE, "batch", "local", "-m", "2", "-u", "-2", "-o", "-5", "-e", "-2"] + flags + ["pairs.fa"]
        one = subprocess.run(base, cwd=tmp_path, capture_output=True, timeout=600)
        cut = subprocess.run(base, cwd=tmp_path, capture_output=True, timeout=600, env=dict(os.environ, AT_CLI_FIRST_CHUNK="64", AT_CLI_CHUNK="64"))
        assert one.returncode == 0 and cut.returncode == 0, (one.stderr, cut.stderr)
        assert one.stdout == cut.stdout and len(one.stdout.splitlines()) == (1000 if flags else 3000)
    # an odd record at the end of a file of several chunks: the error still ends the run with the reference's rc
    with open(tmp_path / "pairs.fa", "a") as fh:
        fh.write(">odd\nACGT\n")
    p = subprocess.run([EXE, "batch", "local", "pairs.fa"], cwd=tmp_path, capture_output=True, timeout=600, env=dict(os.environ, AT_CLI_FIRST_CHUNK="64", AT_CLI_CHUNK="64"))
    assert p.returncode == 255 and b"even number of records (got 2001)" in p.stderr
    p = subprocess.run([EXE, "batch", "local", "pairs.fa"], cwd=tmp_path, capture_output=True, timeout=600)
    assert p.returncode == 255 and b"even number of records (got 2001)" in p.stderr and p.stdout == b""
    # all-vs-all over 3 000 reads
    n = 3000
    reads = ["".join(rng.choice("ACGT") for _ in range(rng.randint(40, 60))) for _ in range(n)]
    with open(tmp_path / "reads.fa", "w") as fh:
        for k, r in enumerate(reads):
            fh.write(">r%d\n%s\n" % (k, r))
    t0 = time.time()
    p = subprocess.run([EXE, "batch", "overlap", "--all-vs-all", "--score-only", "reads.fa"], cwd=tmp_path, capture_output=True, timeout=600,
                       env=dict(os.environ, AT_CLI_CHUNK="1000000"))
    dt = time.time() - t0
    assert p.returncode == 0, p.stderr[-2000:]
    lines = p.stdout.decode().splitlines()
    assert len(lines) == n * (n - 1) // 2
    assert dt < 60, dt
    for q in [0, 1, n - 2, n - 1, 2 * n - 4, 1234567, len(lines) - 2, len(lines) - 1] + [rng.randrange(len(lines)) for _ in range(40)]:
        a, b, sc = lines[q].split("\t")
        ia, ib = int(a[1:]), int(b[1:])
        assert ia < ib and ia * (2 * n - ia - 1) // 2 + (ib - ia - 1) == q, (q, a, b)
        assert sc == "score=%f" % O.align(O.OVERLAP, reads[ia], reads[ib], 1, -2, -5, -1)["score"], (q, a, b)


# ---------------------------------------------------------------- FASTA reader (CPU)
class _Records(C.Structure):
    _fields_ = [("n", C.c_size_t), ("name", C.POINTER(C.c_char_p)), ("comment", C.POINTER(C.c_char_p)),
                ("seq", C.POINTER(C.c_char_p)), ("len", C.POINTER(C.c_size_t))]


def _read(path):
    lib = C.CDLL(HOSTLIB)
    lib.at_read_records.argtypes = [C.c_char_p, C.POINTER(_Records)]
    lib.at_free_records.argtypes = [C.POINTER(_Records)]
    r = _Records()
    rc = lib.at_read_records(str(path).encode(), C.byref(r))
    if rc:
        return None
    out = [(r.name[k].decode(), None if not r.comment[k] else r.comment[k].decode(), r.seq[k].decode()) for k in range(r.n)]
    lib.at_free_records(C.byref(r))
    return out


def test_reader_semantics(built, tmp_path):
    p = tmp_path / "a.fa"
    p.write_bytes(b"junk before\n>r1 first comment\r\nACGT\r\nAC\r\n\n>r2\nTTTT\nGG\n")
    assert _read(p) == [("r1", "first comment", "ACGTAC"), ("r2", "first comment", "TTTTGG")]   # comment buffer is shared
    q = tmp_path / "b.fq.gz"
    with gzip.open(q, "wb") as fh:
        fh.write(b"@q1 c1\nACGT\n+\nIIII\n@q2\tx|y\nGG\nTT\n+anything\nII\nII\n")
    assert _read(q) == [("q1", "c1", "ACGT"), ("q2", "x|y", "GGTT")]
    assert _read(tmp_path / "missing.fa") is None
    e = tmp_path / "c.fa"
    e.write_bytes(b">only\n")
    assert _read(e) == [("only", None, "")]


def test_reader_errors_like_reference(built, tmp_path):
    (tmp_path / "three.fa").write_text(">a\nAC\n>b\nAC\n>c\nAC\n")
    (tmp_path / "one.fa").write_text(">a\nAC\n")
    rc, so, se = _run(["edit", "three.fa"], tmp_path)
    assert (rc, so, se) == (255, "", "FATAL ERROR: input fasta file has more than 2 sequences\n")
    rc, so, se = _run(["edit", "one.fa"], tmp_path)
    assert (rc, so, se) == (255, "", "FATAL ERROR: read_kstring: fail to read sequence\n")
    (tmp_path / "big.fa").write_text(">a\nACGTACGT\n>b\nACG\n")
    rc, so, se = _run(["fit", "big.fa"], tmp_path)
    assert (rc, so, se) == (255, "", "FATAL ERROR: first sequence must be shorter than the second\n\n")


def _file_cases(need_gpu):
    return [c for c in load_golden("cli_files.jsonl") if (c["rc"] == 0) == need_gpu]


def _replay(case, tmp_path):
    (tmp_path / case["file"]).write_bytes(base64.b64decode(case["data"]))
    rc, so, se = _run(case["argv"], tmp_path)
    assert rc == case["rc"], (case["argv"], rc, se)
    assert se == case["stderr"], case["argv"]
    assert so == case["stdout"], case["argv"]


@pytest.mark.parametrize("case", _file_cases(False), ids=lambda c: " ".join(c["argv"]))
def test_cli_input_files_error_paths(built, tmp_path, case):
    """Reader errors on synthetic input files, recorded from the stock binary (oracle/make_cli_golden.py)."""
    _replay(case, tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("case", _file_cases(True), ids=lambda c: " ".join(c["argv"]))
def test_cli_input_files_match_reference(built, tmp_path, case):
    """Multi-line / CRLF / FASTQ / gzip / lower-case / commented inputs through every sub-command: stdout, stderr and
    return code equal what the stock binary printed on the same files."""
    _replay(case, tmp_path)


@pytest.mark.gpu
def test_cli_batch_all_vs_all_min_score(built, tmp_path, fake_rccl):
    """`batch overlap --all-vs-all --score-only --min-score T` prints the lines of the unthresholded run whose score reaches T, in the
    same order (pairs the bit-parallel bound proves below T are not swept at all); with --gpus N too; and the flag goes with
    overlap --all-vs-all --score-only only."""
    rng = random.Random(23)
    n = 400
    reads = ["".join(rng.choice("ACGT") for _ in range(rng.randint(300, 400))) for _ in range(n)]
    for k in range(0, n - 1, 5):       # real overlaps: read k + 1 starts with the end of read k
        ov = rng.randint(40, 250)
        reads[k + 1] = (reads[k][-ov:] + reads[k + 1])[:400]
    with open(tmp_path / "reads.fa", "w") as fh:
        for k, r in enumerate(reads):
            fh.write(">r%d\n%s\n" % (k, r))
    base = [EXE, "batch", "overlap", "--all-vs-all", "--score-only"]
    full = subprocess.run(base + ["reads.fa"], cwd=tmp_path, capture_output=True, timeout=600)
    assert full.returncode == 0, full.stderr[-2000:]
    lines = full.stdout.decode().splitlines()
    assert len(lines) == n * (n - 1) // 2
    for T in (25, 100):
        want = [ln for ln in lines if float(ln.rsplit("score=", 1)[1]) >= T]
        assert len(want) >= (40 if T == 25 else 20)
        p = subprocess.run(base + ["--min-score", str(T), "reads.fa"], cwd=tmp_path, capture_output=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        assert p.stdout.decode().splitlines() == want, T
        env = dict(os.environ, AT_RCCL_LIB=fake_rccl, AT_ONE_DEVICE="1")
        p = subprocess.run(base + ["--min-score", str(T), "--gpus", "2", "reads.fa"], cwd=tmp_path, capture_output=True, timeout=600, env=env)
        assert p.returncode == 0, p.stderr[-2000:]
        assert p.stdout.decode().splitlines() == want, T
    p = subprocess.run([EXE, "batch", "local", "--all-vs-all", "--score-only", "--min-score", "5", "reads.fa"], cwd=tmp_path, capture_output=True)
    assert p.returncode == 1 and b"--min-score goes with" in p.stderr
