#!/bin/bash
# usage: tools/ab2.sh lib1.so lib2.so ...   (nn probe and loop probe for each)
set -e
for lib in "$@"; do
  echo "== $lib"
  SVICP_LIBRARY=$PWD/$lib timeout -k 10 120 python tools/nn_probe.py
  SVICP_LIBRARY=$PWD/$lib timeout -k 10 120 python tools/search_counts_probe.py | tail -1
done
