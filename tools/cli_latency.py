"""Wall time of the drop-in CLI next to the stock reference binary (oracle/_ref/alignTools, built in the
container and shipped as a binary) on the reference's own example inputs, re-created from tests/golden."""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ours = os.path.join(ROOT, "aligntools", "c_amd", "bin", "alignTools")
ref = os.path.join(ROOT, "oracle", "_ref", "alignTools")
seen = {}
for line in open(os.path.join(ROOT, "tests", "golden", "known_answers.jsonl")):
    c = json.loads(line)
    seen.setdefault(c["tag"].split(" ")[0], (c["s1"], c["s2"]))
d = tempfile.mkdtemp()
os.makedirs(os.path.join(d, "test"))
for tag, (s1, s2) in seen.items():
    with open(os.path.join(d, "test", tag + ".fa"), "w") as fh:
        fh.write(">a\n%s\n>b%s\n%s\n" % (s1, " 1036|3395|23045|24611" if tag == "test_fit" else "", s2))
for argv in (["local", "-m", "2", "-u", "-2", "-o", "-5", "-e", "-2", "test/test_local.fa"],
             ["global", "test/test_global.fa"], ["edit", "test/test_edit.fa"], ["fit", "-m", "2", "-u", "-2", "-s", "test/test_fit.fa"]):
    row = []
    for exe in (ref, ours):
        if not os.path.exists(exe):
            row.append("n/a"); continue
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            p = subprocess.run([exe] + argv, cwd=d, capture_output=True)
            best = min(best, time.perf_counter() - t0)
        row.append("%.3f s (rc %d)" % (best, p.returncode))
    print(" ".join(argv), "| reference", row[0], "| MI355X", row[1])
