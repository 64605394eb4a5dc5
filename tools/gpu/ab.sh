#!/bin/bash
# ab.sh OUT "VARIANT ..." "WORKLOAD ..." [bench options] -- A/B runs on one box.  A VARIANT is `name=ENV1=val,ENV2=val` (environment
# knobs) or `name=@path/to/lib.so` (a library built by tools/ab_lib.sh, loaded through AT_LIB_PATH); `base=` is the
# tree's own build.  One JSON row per (variant, workload) in gpurun_out/OUT/ab.jsonl.
#   bash tools/gpu/ab.sh r04x "base= onepass=AT_TWO_PASS=0 st=@aligntools/c_amd/exp/libaligntools_hip_st.so" "C2 C3" --steps 30 --streams 1
set -e
export TMPDIR=/tmp
O=gpurun_out/$1; VARS=$2; WLS=$3; shift 3
mkdir -p $O
for W in $WLS; do for V in $VARS; do
  NAME=${V%%=*}; SPEC=${V#*=}
  ENVS=""
  if [ "${SPEC#@}" != "$SPEC" ]; then ENVS="AT_LIB_PATH=${SPEC#@}"; elif [ -n "$SPEC" ]; then ENVS=$(echo "$SPEC" | tr ',' ' '); fi
  env $ENVS timeout -k 10 400 python3 bench.py --workload $W --no-cpu-baseline "$@" 2>> $O/ab.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r={'variant':'$NAME','workload':'$W','gcups':round(d['value'],1),'ms_per_step':round(d['ms_per_step'],4),'alone_gcups':round(d['roofline']['gcups_one_launch_at_a_time'],1),'kernel_config':d['config']['kernel_config']}; print(json.dumps(r))" | tee -a $O/ab.jsonl | cut -c1-200
done; done
