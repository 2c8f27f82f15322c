"""Per kernel of interest: mean FETCH_SIZE and WRITE_SIZE per launch from the two counter passes
of tools/byte_budget.sh (bytes = 2 * 1024 * FETCH_SIZE + 1024 * WRITE_SIZE, the correction
gfx950 needs: a 128-byte read request is tallied as 64 bytes; both counters are in KB)."""
import collections
import csv
import glob
import json
import sys

out_dir, variant = sys.argv[1], sys.argv[2]
want = {"icp_iterate_kernel<true, false, 4, true": "iterate_searching_every_point",
        "nn1_kernel<false": "nn1_capped"}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{out_dir}/v{variant}_{ctr}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            for key, name in want.items():
                if key in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                    acc[name][ctr].append(float(r["Counter_Value"]))
res = {"variant_bits": int(variant)}
for name, d in acc.items():
    f = sum(d["FETCH_SIZE"]) / max(len(d["FETCH_SIZE"]), 1)
    w = sum(d["WRITE_SIZE"]) / max(len(d["WRITE_SIZE"]), 1)
    res[name] = {"launches": len(d["FETCH_SIZE"]), "fetch_MB": 2 * 1024 * f / 1e6, "write_MB": 1024 * w / 1e6,
                 "total_MB": (2 * 1024 * f + 1024 * w) / 1e6}
print(json.dumps(res))
