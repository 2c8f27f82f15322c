#!/bin/bash
# the probe of the byte budget without a profiler, per variant: what each dropped stream is worth in time
set -e
for v in ${VARIANTS:-0 1 2 4 8 24 56 63 64}; do
  echo "== drop $v"
  SVICP_LIBRARY=$PWD/build_ab/libsvicp_drop$v.so timeout -k 10 120 python3 tools/byte_budget_probe.py
done
