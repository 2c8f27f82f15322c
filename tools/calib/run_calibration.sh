#!/bin/bash
# FETCH_SIZE per launch of build_ab/fetch_calibration (see tools/calib/fetch_calibration.hip)
R=$PWD; O=$R/gpurun_out/calib; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/a -o p -- $R/build_ab/fetch_calibration > $O/a.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $O/b -o p -- $R/build_ab/fetch_calibration > $O/b.log 2>&1 || echo "second pass failed"
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $O/c -o p -- $R/build_ab/fetch_calibration > $O/c.log 2>&1 || echo "third pass failed"
cd $R
python3 - <<'PY'
import csv, glob, collections
for tag in "abc":
    for f in glob.glob(f"gpurun_out/calib/{tag}/**/*counter_collection.csv", recursive=True):
        rows = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            if "strided" in r["Kernel_Name"]:
                rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
        for n, (i, d) in enumerate(sorted(rows.items())):
            print(tag, "launch", n, "mode", n % 6, d)
PY
grep "^mode" $O/a.log | head -6
rm -rf $O/a $O/b $O/c
