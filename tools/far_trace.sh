#!/bin/bash
# Development: kernel trace of the exact search at the start pose for settings of the shared-box
# levels (largest box in blocks per level; parts a wave that gives up is split into).
# usage: tools/far_trace.sh "64:128,4:128,1:128" ...   (lanes of an entry : largest box, per level)
R=$PWD
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do

  export SVICP_FAR_LEVELS=$cfg
  rm -rf $R/gpurun_out/fartrace
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/fartrace -o far -- python3 $R/tools/nn_probe.py > /dev/null 2>&1 || exit 1
  echo "$cfg: $(python3 $R/tools/far_trace_sum.py $R/gpurun_out/fartrace/far_kernel_trace.csv)"
done
