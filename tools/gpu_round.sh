#!/bin/bash
# One GPU call: tests, benchmark, rocprofv3 kernel statistics of the same benchmark command, and
# the two HBM counter passes.  Every step under its own time limit; stops at the first failure.
set -e
O=gpurun_out/round
rm -rf $O && mkdir -p $O
R=$PWD
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $O/tests.log 2>&1
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -o stats -- python3 $R/bench.py --cpu-iters 0 > $R/$O/bench_under_rocprof.json 2> $R/$O/rocprof.err
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/fetch -o fetch -- python3 $R/bench.py --steps 3 --warmup 1 --sharded-side 0 --cpu-iters 0 --cal-views 0 --batch-pairs 0 --extras 0 > /dev/null 2> $R/$O/fetch.err
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$O/write -o write -- python3 $R/bench.py --steps 3 --warmup 1 --sharded-side 0 --cpu-iters 0 --cal-views 0 --batch-pairs 0 --extras 0 > /dev/null 2> $R/$O/write.err
cd $R
python tools/summarise_profiles.py stats $O/stats $O/r01 
python tools/summarise_profiles.py traffic $O/fetch $O/write $O/r01_traffic.json 54525952
rm -rf $O/stats $O/fetch $O/write
tail -3 $O/tests.log
