#!/bin/bash
# ab_lib.sh NAME "EXTRA FLAGS" -- an A/B variant of libaligntools_hip.so: the two-pass units and the shim compiled with EXTRA, every other
# object taken from the main build (aligntools/c_amd/build).  Result: aligntools/c_amd/exp/libaligntools_hip_NAME.so (load with AT_LIB_PATH).
set -e
cd "$(dirname "$0")/../aligntools/c_amd"
N=$1; EXTRA=$2
B=build_$N
mkdir -p $B exp
FL="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -I../../include $EXTRA"
for u in at_k16_tp at_k16_tp64 at_k16_walk8 at_k16_walk64; do
  /opt/rocm/bin/hipcc $FL -c csrc/$u.hip -o $B/$u.o &
  /opt/rocm/bin/hipcc $FL -DAT_BITS16=8 -c csrc/$u.hip -o $B/${u}_b8.o &
done
/opt/rocm/bin/hipcc $FL -c csrc/at_hip.hip -o $B/at_hip.o &
wait
OBJS=$(ls build/*.o | grep -v -E "/(at_k16_tp|at_k16_tp64|at_k16_walk8|at_k16_walk64|at_k16_tp_b8|at_k16_tp64_b8|at_k16_walk8_b8|at_k16_walk64_b8|at_hip)\.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared $OBJS $B/*.o -o exp/libaligntools_hip_$N.so -ldl
echo built exp/libaligntools_hip_$N.so
