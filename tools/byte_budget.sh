#!/bin/bash
# Byte budget of a launch that searches every point and of the standalone capped search: counter
# passes (FETCH_SIZE, WRITE_SIZE, each a run of its own) over builds that drop one stream of
# traffic at a time (build_ab/libsvicp_drop<bits>.so, made by `make -C tools/.. budget_builds` on
# the CPU beforehand: see tools/build_budget_variants.sh).  Writes gpurun_out/budget/<bits>.json.
set -e
R=$PWD; O=$R/gpurun_out/budget; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-0 1 2 4 8 24 56 63 64}; do
  lib=$R/build_ab/libsvicp_drop$v.so
  [ -f $lib ] || { echo "missing $lib"; exit 1; }
  export SVICP_LIBRARY=$lib
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 120 rocprofv3 --pmc $ctr --output-format csv -d $O/v${v}_$ctr -o p -- python3 $R/tools/byte_budget_probe.py > $O/v${v}_$ctr.log 2>&1
  done
  python3 $R/tools/byte_budget_sum.py $O $v > $O/$v.json
  cat $O/$v.json
  rm -rf $O/v${v}_FETCH_SIZE $O/v${v}_WRITE_SIZE
done
