#!/bin/bash
# CPU side of the byte budget: the product's sources built with -DSVICP_DROP=<bits> into build_ab/.
set -e
cd "$(dirname "$0")/.."
mkdir -p build_ab
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Iinclude -Istereovision_amd/csrc -Wall -Wno-unused-result"
for v in ${VARIANTS:-0 1 2 4 8 24 56 63 64}; do
  ( hipcc $F -DSVICP_DROP=$v -o build_ab/libsvicp_drop$v.so stereovision_amd/csrc/svicp_api.hip stereovision_amd/csrc/svcal_api.hip && echo built $v ) &
  while [ $(jobs -r | wc -l) -ge 4 ]; do sleep 1; done
done
wait
