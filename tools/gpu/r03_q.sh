#!/bin/bash
# round 3, call Q: overlap scores-only and edit distance swept minus the gap ramp (two instructions per cell, no per-step row pick): parity, rates
set -e
export TMPDIR=/tmp
O=gpurun_out/r03q
mkdir -p $O
AT_FUZZ_MODES=overlap AT_FUZZ_TB=0 python3 tests/fuzz_parity.py 60000 42 | cut -c1-900
AT_FUZZ_MODES=overlap,edit AT_FUZZ_TB=0 python3 tests/fuzz_parity.py 40000 43 | cut -c1-900
