#!/bin/bash
# A/B on one GPU box: usage tools/ab_probe.sh lib1.so lib2.so ... (default: base against the tree)
set -e
mkdir -p gpurun_out
libs="$@"; [ -z "$libs" ] && libs="build_ab/libsvicp_base.so stereovision_amd/libsvicp.so"
for lib in $libs; do
  echo "== $lib"
  SVICP_LIBRARY=$PWD/$lib timeout -k 10 120 python tools/search_counts_probe.py
  SVICP_LIBRARY=$PWD/$lib timeout -k 10 120 python tools/lateiter.py
done
