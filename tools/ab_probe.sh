#!/bin/bash
# A/B on one GPU box, interleaved: usage tools/ab_probe.sh lib1.so lib2.so ... (default: base against the tree)
set -e
libs="$@"; [ -z "$libs" ] && libs="build_ab/libsvicp_base.so stereovision_amd/libsvicp.so"
for round in 1 2; do
  for lib in $libs; do
    SVICP_LIBRARY=$PWD/$lib timeout -k 10 120 python tools/ab_value.py
  done
done
