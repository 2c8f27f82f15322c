#!/bin/bash
# A/B of search variants on one GPU box: for each library the headline step (two rounds, in turns),
# the standalone searches at registered positions and the run that searches every point.
# usage: tools/ab_search.sh lib1.so lib2.so ...
set -e
for round in 1 2; do
  for lib in "$@"; do
    SVICP_LIBRARY=$PWD/$lib timeout -k 10 120 python3 tools/ab_value.py
  done
done
for lib in "$@"; do
  SVICP_LIBRARY=$PWD/$lib timeout -k 10 120 python3 tools/nn_registered_probe.py
  echo -n "$lib  "; SVICP_LIBRARY=$PWD/$lib timeout -k 10 120 python3 tools/nn_probe.py | tr '\n' ' '; echo
done
