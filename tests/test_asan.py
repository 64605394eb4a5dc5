"""CPU sanitizer build ("build's CPU code should be ASan/UBSan-clean"): `make -C aligntools/c_amd asan`
compiles the C host, the oracle restatement and the C consumer with -fsanitize=address,undefined; this module runs them
on the inputs the reference's own CLI was recorded on.  CPU only -- GPU sanitizers are not available on this pool."""
import base64
import json
import os
import subprocess

import pytest

from conftest import ROOT, load_golden

PKG = os.path.join(ROOT, "aligntools", "c_amd")
ENV = dict(os.environ, ASAN_OPTIONS="abort_on_error=0:exitcode=97:detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")


@pytest.fixture(scope="module")
def asan():
    p = subprocess.run(["make", "-C", PKG, "asan"], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    return os.path.join(PKG, "build_asan")


def _run(cmd, **kw):
    p = subprocess.run(cmd, capture_output=True, env=kw.pop("env", ENV), timeout=300, **kw)
    out, err = p.stdout.decode("latin1"), p.stderr.decode("latin1")
    assert "ERROR: AddressSanitizer" not in err and "runtime error:" not in err and "LeakSanitizer" not in err, err[-3000:]
    assert p.returncode != 97, err[-3000:]
    return p.returncode, out, err


def test_parser_on_every_recorded_input(asan, tmp_path):
    """The gz FASTA / FASTQ reader and the site parser on every input file of the CLI recordings (multi-line, CRLF, FASTQ
    incl. truncated quality blocks, gzip, comments, empty records) plus hostile ones: leak-checked, bounds-checked."""
    files = []
    for k, c in enumerate(load_golden("cli_files.jsonl")):
        f = tmp_path / ("%03d_%s" % (k, c["file"]))
        f.write_bytes(base64.b64decode(c["data"]))
        files.append(str(f))
    hostile = {"empty.fa": b"", "only_header.fa": b">x", "no_newline.fa": b">a c|1|2\nACGT", "pipes.fa": b">a\nAC\n>b |||9||x|-3|\nACGT\n",
               "plus.fq": b"@r\nACGT\n+\n", "shortq.fq": b"@r\nACGT\n+\nII", "crlf.fa": b">a x\r\nAC\r\nGT\r\n>b 1|2\r\nA\r\n",
               "binary.fa": bytes(range(256)) * 40, "long_line.fa": b">a\n" + b"ACGT" * 50000 + b"\n>b " + b"1|" * 3000 + b"\nAC\n"}
    for name, data in hostile.items():
        f = tmp_path / name
        f.write_bytes(data)
        files.append(str(f))
    files.append(str(tmp_path / "missing.fa"))
    rc, out, err = _run([os.path.join(asan, "asan_parser")] + files)
    assert rc == 0, err
    assert out.count("\n") == len(files)
    assert "long_line.fa: 2 records, 200002 bases, 3000 sites" in out


def test_oracle_under_sanitizers(asan):
    rc, out, err = _run([os.path.join(asan, "asan_oracle")])
    assert rc == 0 and "asan_oracle ok" in out, out + err


def test_cli_error_paths_under_sanitizers(asan):
    """Every recorded invocation of the stock CLI that ends before the first DP cell (usage texts, unknown command, bad
    options, missing file, wrong record count, fit with l1 > l2, -s without a comment) replayed on the instrumented CLI:
    same stdout / stderr / return code, nothing for the sanitizers to report.  (die() exits without freeing, like the
    reference's: leak detection is off for these.)"""
    env = dict(ENV, ASAN_OPTIONS="exitcode=97:detect_leaks=0")
    exe = os.path.join(asan, "alignTools")
    n = 0
    for c in load_golden("cli.jsonl"):
        if c["rc"] == 0:
            continue   # needs the GPU
        p = subprocess.run([exe] + c["argv"], cwd=os.path.join(ROOT, "tests", "golden"), capture_output=True, env=env, timeout=120)
        err = p.stderr.decode("latin1")
        assert "AddressSanitizer" not in err and "runtime error:" not in err, err[-2000:]
        if any(a.startswith("test/") for a in c["argv"]):
            continue   # (recorded in the reference's own directory: its test/*.fa are not in this repo)
        assert p.returncode == c["rc"], (c["argv"], p.returncode, err)
        assert err.replace(exe, "alignTools") == c["stderr"], c["argv"]
        n += 1
    assert n >= 8


def test_c_consumer_host_half_under_sanitizers(asan):
    env = dict(ENV, ASAN_OPTIONS="exitcode=97:detect_leaks=0")   # (the HIP runtime's own start-up allocations are not ours to free)
    rc, out, err = _run([os.path.join(asan, "abi_consumer"), "nogpu"], env=env)
    assert rc == 0 and "host-only ok" in out, out + err
