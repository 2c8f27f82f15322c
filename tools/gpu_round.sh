#!/bin/bash
# One GPU call: tests, benchmark, rocprofv3 kernel statistics of the same benchmark command, and
# the counter passes (each a run of its own over tools/profile_step.py).  Every step under its
# own time limit; stops at the first failure.  Summaries land in gpurun_out/round/ and are
# copied to profiles/ by hand.
set -e
TAG=${1:-r02}
O=gpurun_out/round
rm -rf $O && mkdir -p $O
R=$PWD
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/tests.log 2>&1
timeout -k 10 400 python bench.py > $O/${TAG}_bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -o stats -- python3 $R/bench.py --cpu-iters 0 > $R/$O/${TAG}_bench_under_rocprof.json 2> $R/$O/rocprof.err
P="python3 $R/tools/profile_step.py"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/p1 -o p -- $P > /dev/null 2> $R/$O/p1.err
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$O/p2 -o p -- $P > /dev/null 2> $R/$O/p2.err
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INSTS_VALU --output-format csv -d $R/$O/p3 -o p -- $P > /dev/null 2> $R/$O/p3.err
timeout -s KILL 120 rocprofv3 --pmc TA_TA_BUSY_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $R/$O/p4 -o p -- $P > /dev/null 2> $R/$O/p4.err
# (a pass whose counters this GPU does not offer is left out of the summary, not fatal)
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $R/$O/p5 -o p -- $P > /dev/null 2> $R/$O/p5.err || rm -rf $R/$O/p5
cd $R
python tools/summarise_profiles.py stats $O/stats $O/$TAG
python tools/summarise_profiles.py regimes $O/${TAG}_counters.json $O/${TAG}_traffic.json $O/p1 $O/p2 $O/p3 $O/p4 $([ -d $O/p5 ] && echo $O/p5)
rm -rf $O/p5 $O/stats $O/p1 $O/p2 $O/p3 $O/p4
tail -n 3 $O/tests.log
