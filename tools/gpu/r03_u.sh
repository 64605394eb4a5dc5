#!/bin/bash
# round 3, call U: fit -s without the jump state's opening in blocks where no column may open: parity (campaign aimed at fit -s), C4 rates
set -e
export TMPDIR=/tmp
O=gpurun_out/r03u
mkdir -p $O
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fit or jump or golden or frames" 2>&1 | tail -3
AT_FUZZ_MODES=fitj python3 tests/fuzz_parity.py 60000 51 | cut -c1-700
AT_FUZZ_MODES=fitj AT_FUZZ_TB=0 python3 tests/fuzz_parity.py 30000 52 | cut -c1-700
for tb in "" "--no-traceback"; do
  python3 bench.py --workload C4 --steps 40 --warmup 5 --no-cpu-baseline $tb > $O/C4$tb.json 2> $O/C4$tb.err
  python3 -c "import json; d=json.load(open('$O/C4$tb.json')); print('C4 $tb', round(d['value'],1), 'GCUPS', round(d['ms_per_step'],3), 'ms', d['config']['kernel_config'][:90])"
done
python3 bench.py --workload C4 --steps 40 --warmup 5 --no-cpu-baseline --streams 1 > $O/C4s1.json 2> $O/C4s1.err
python3 -c "import json; d=json.load(open('$O/C4s1.json')); print('C4 one launch at a time', round(d['value'],1), 'GCUPS')"
