#!/bin/bash
# GPU-side timeline of one at_align_batch call (kernels and copies with their timestamps)
set -e
export TMPDIR=/tmp
R=$PWD
O=gpurun_out/r03l
rm -rf $O; mkdir -p $O
(cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace -d $R/$O/tl -o tl --output-format csv -- python3 $R/tools/host_path_rate.py > $R/$O/rate.txt 2> $R/$O/err.txt) || true
cat $O/rate.txt | cut -c1-100
python3 - <<'PY'
import csv, glob
kern = []; cp = []
for f in glob.glob("gpurun_out/r03l/tl/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kern.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]))
for f in glob.glob("gpurun_out/r03l/tl/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        cp.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "?") + " " + str(r.get("Bytes", r.get("Size", "?")))))
ev = sorted([(a, b, "K " + n) for a, b, n in kern] + [(a, b, "C " + n) for a, b, n in cp])
# the traceback=1 calls come first (6 calls); find bursts separated by > 1 ms of silence
bursts = []; cur = []
for e in ev:
    if cur and e[0] - max(x[1] for x in cur) > 1_000_000:
        bursts.append(cur); cur = []
    cur.append(e)
if cur: bursts.append(cur)
big = [b for b in bursts if sum(1 for e in b if "at_sweep16" in e[2]) >= 6]
print(len(bursts), "bursts,", len(big), "with 6 sweeps")
b = big[3] if len(big) > 3 else big[-1]
t0 = b[0][0]
with open("gpurun_out/r03l/timeline_one_call.txt", "w") as fh:
    for a, e, n in b:
        line = "%9.1f us .. %9.1f us  (%7.1f)  %s" % ((a - t0) / 1e3, (e - t0) / 1e3, (e - a) / 1e3, n)
        fh.write(line + "\n")
print(open("gpurun_out/r03l/timeline_one_call.txt").read()[:6000])
PY
rm -rf $O/tl
