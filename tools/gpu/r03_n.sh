#!/bin/bash
# round 3, call N: the N > 1 bookkeeping of bench.py with 4 and 6 ranks (gloo, ranks sharing the card), started as plain `python3 bench.py --gpus N`
set -e
export TMPDIR=/tmp
O=gpurun_out/r03n
rm -rf $O; mkdir -p $O
for n in 4 6; do
  timeout -k 10 500 python3 bench.py --gpus $n --backend gloo --pairs 20000 --steps 9 --warmup 2 --no-cpu-baseline > $O/dist_gloo${n}_C2.json 2> $O/dist_gloo${n}_C2.err || { tail -20 $O/dist_gloo${n}_C2.err; exit 1; }
  python3 -c "import json; d=json.load(open('$O/dist_gloo${n}_C2.json')); g=d['config']['gather']; print('$n ranks:', d['n_gpus'], round(d['value'],1), 'GCUPS (ranks share one card)', g['world'], [r['rank'] for r in g['ranks']], g['steps_per_collective'], g['cigar_bytes_per_rank_and_step'])"
done
timeout -k 10 500 python3 bench.py --gpus 3 --backend gloo --workload C4 --pairs 15000 --steps 7 --warmup 2 --gather-every 3 --no-cpu-baseline > $O/dist_gloo3_C4.json 2> $O/dist_gloo3_C4.err
python3 -c "import json; d=json.load(open('$O/dist_gloo3_C4.json')); print('3 ranks C4:', d['n_gpus'], round(d['value'],1))"
