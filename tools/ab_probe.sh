#!/bin/bash
# A/B on one GPU box: the library in the tree against build_ab/libsvicp_base.so.
set -e
mkdir -p gpurun_out
for lib in build_ab/libsvicp_base.so stereovision_amd/libsvicp.so; do
  echo "== $lib"
  SVICP_LIBRARY=$PWD/$lib timeout -k 10 120 python tools/search_counts_probe.py
  SVICP_LIBRARY=$PWD/$lib timeout -k 10 120 python tools/lateiter.py
done
