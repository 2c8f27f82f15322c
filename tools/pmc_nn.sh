#!/bin/bash
# counter passes over the standalone 1-NN kernels (development)
R=$PWD; O=$R/gpurun_out/pmcnn; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD --output-format csv -d $O/a -o a -- python3 $R/tools/nn_probe.py > $O/a.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc TA_TA_BUSY_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $O/b -o b -- python3 $R/tools/nn_probe.py > $O/b.log 2>&1 || exit 1
cd $R
python3 - <<'PY'
import csv, glob, collections
for tag in "ab":
    for f in glob.glob(f"gpurun_out/pmcnn/{tag}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:40]
            if "nn1_kernel" in k or "icp_iterate" in k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in acc.items():
            print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
rm -rf $O/a $O/b
