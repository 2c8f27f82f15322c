#!/bin/bash
# counter pass over the iteration kernel, per launch of the last run (development)
R=$PWD; O=$R/gpurun_out/pmcit; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc $1 --output-format csv -d $O/a -o a -- python3 $R/tools/search_counts_probe.py > $O/a.log 2>&1 || exit 1
cd $R
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/pmcit/a/**/*counter_collection.csv", recursive=True):
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "icp_iterate_kernel" in r["Kernel_Name"] or "solve_kernel" in r["Kernel_Name"]:
            rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
            rows[int(r["Dispatch_Id"])]["k"] = r["Kernel_Name"][20:60]
    ids = sorted(rows)[-31:]
    names = [c for c in rows[ids[0]] if c != "k"]
    print("launch", *names)
    for n, i in enumerate(ids):
        print(n, *[int(rows[i].get(c, 0)) for c in names], rows[i]["k"][:30])
PY
rm -rf $O/a
