"""End-to-end rate of the batch CLI (`alignTools batch local pairs.fa`): parse 2N FASTA records (plain and gzip), one GPU
batch with the strings rendered on the GPU, print.  C2 workload (100k pairs of 150 x 150).  Quoted in DESIGN.md section 7."""
import gzip
import os
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aligntools.c_amd.synth import synth_pairs_blob   # noqa: E402

EXE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "aligntools", "c_amd", "bin", "alignTools")
n, l1, l2 = 100000, 150, 150
blob = synth_pairs_blob(0x5EED0002, n, l1, l2)
with tempfile.TemporaryDirectory() as d:
    plain = os.path.join(d, "pairs.fa")
    with open(plain, "wb") as fh:
        for k, row in enumerate(blob):
            fh.write(b">a%d\n" % k + row[:l1].tobytes() + b"\n>b%d\n" % k + row[l1:].tobytes() + b"\n")
    gz = plain + ".gz"
    with open(plain, "rb") as f, gzip.open(gz, "wb", compresslevel=4) as g:
        g.write(f.read())
    for path in (plain, gz):
        for it in range(2):
            t0 = time.perf_counter()
            p = subprocess.run([EXE, "batch", "local", "-m", "2", "-u", "-2", "-o", "-5", "-e", "-2", path], stdout=subprocess.DEVNULL,
                               stderr=subprocess.PIPE)
            t = time.perf_counter() - t0
            assert p.returncode == 0, p.stderr.decode()[-500:]
        print("%s (%.1f MB): %.3f s for %d pairs = %.1f GCUPS end to end" % (os.path.basename(path), os.path.getsize(path) / 1e6, t, n,
                                                                          n * l1 * l2 / t / 1e9))
