#!/bin/bash
set -e
export TMPDIR=/tmp
O=gpurun_out/r02k
mkdir -p $O
run() { local label=$1; shift
  timeout -k 10 300 env "$@" > $O/bench_$label.json 2> $O/bench_$label.err || true
  python3 tools/bl.py $label < $O/bench_$label.json || true
}
python3 - <<'PY'
import random, sys
sys.path.insert(0, ".")
import aligntools.c_amd as A, oracle as O, os
rng = random.Random(3)
for rows in ("8", "16"):
    os.environ["AT_ROWS16"] = rows
    os.environ["AT_PACKED_MIN_ROUNDS"] = "0"
    al = A.Aligner()
    for mode, sc in (("global", (1, -1, -4, -1)), ("local", (2, -2, -5, -2))):
        pairs = []
        for k in range(6):
            a = "".join(rng.choice("ACGT") for _ in range(1024))
            b = a[:500] + "".join(rng.choice("ACGT") for _ in range(30)) + a[520:] if k % 2 else "".join(rng.choice("ACGT") for _ in range(1024))
            pairs.append((a, b[:1024].ljust(1024, "A")))
        al.set_scoring(*sc)
        res = al.align_batch(mode, pairs, render=False)
        for k, (a, b) in enumerate(pairs):
            r = O.align(O.MODE_NAMES[mode], a, b, *sc)
            assert (int(res["score"][k]), res["ops"][k]) == (r["score"], r["ops"]), (rows, mode, k)
        print("parity ok", rows, mode, al.last_config[:110])
    al.close()
PY
run C3_k4 python3 bench.py --workload C3 --steps 60 --no-cpu-baseline
run C3_k8 AT_ROWS16=8 python3 bench.py --workload C3 --steps 60 --no-cpu-baseline
run C3_k16 AT_ROWS16=16 python3 bench.py --workload C3 --steps 60 --no-cpu-baseline
run C3_k4_b python3 bench.py --workload C3 --steps 60 --no-cpu-baseline
run C3_k16_b AT_ROWS16=16 python3 bench.py --workload C3 --steps 60 --no-cpu-baseline
run C3_k16_scores AT_ROWS16=16 python3 bench.py --workload C3 --steps 60 --no-cpu-baseline --no-traceback
run C3_k4_scores python3 bench.py --workload C3 --steps 60 --no-cpu-baseline --no-traceback
