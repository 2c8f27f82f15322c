"""Turns rocprofv3 output directories into the summaries kept under profiles/.

  python tools/summarise_profiles.py stats  <rocprof dir> <out prefix>
      copies *_kernel_stats.csv (kernel names shortened) to <prefix>_kernel_stats.csv and writes
      <prefix>_kernel_by_grid.csv: calls and average duration per (kernel, grid size), from
      *_kernel_trace.csv -- the stats file mixes launches of one kernel over different sizes.
  python tools/summarise_profiles.py traffic <fetch dir> <write dir> <out json> [algorithmic bytes]
      mean FETCH_SIZE / WRITE_SIZE per launch per (kernel, grid size) from two --pmc passes.
      FETCH_SIZE is in units of 1 KB and on gfx950 tallies a 128-byte read request as 64 bytes:
      bytes = 2 * 1024 * FETCH_SIZE.  WRITE_SIZE: bytes = 1024 * WRITE_SIZE.
  python tools/summarise_profiles.py regimes <out json> <traffic json> <pass dir> [...]
      counter passes over tools/profile_step.py: means per (kernel, regime of the launch).
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    if not hits:
        raise SystemExit(f"no *{suffix} under {d}")
    return hits


def short(name):
    name = re.sub(r"\(.*", "", name)                 # drop the argument list
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"rocprim::ROCPRIM_\d+_NS::detail::", "rocprim::", name)
    return name[:120]


def stats(src, prefix):
    rows = []
    for f in find(src, "kernel_stats.csv"):
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    with open(prefix + "_kernel_stats.csv", "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                        r["Percentage"], r["MinNs"], r["MaxNs"]])
    agg = defaultdict(list)
    for f in find(src, "kernel_trace.csv"):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                grid = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1)) * int(r.get("Grid_Size_Z", 1))
                agg[(short(r["Kernel_Name"]), grid)].append(
                    int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    with open(prefix + "_kernel_by_grid.csv", "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "GridThreads", "Calls", "AverageNs", "MinNs", "MaxNs"])
        for (name, grid), d in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([name, grid, len(d), round(sum(d) / len(d), 1), min(d), max(d)])


def counters(d, counter):
    agg = defaultdict(list)
    for f in find(d, "counter_collection.csv"):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] != counter:
                    continue
                agg[(short(r["Kernel_Name"]), int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return agg


def traffic(fetch_dir, write_dir, out, algorithmic=None):
    fe, wr = counters(fetch_dir, "FETCH_SIZE"), counters(write_dir, "WRITE_SIZE")
    res = {}
    for key in sorted(set(fe) | set(wr)):
        f = sum(fe[key]) / len(fe[key]) if key in fe else 0.0
        w = sum(wr[key]) / len(wr[key]) if key in wr else 0.0
        res[f"{key[0]} grid={key[1]}"] = {
            "launches": len(fe.get(key, wr.get(key))), "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
            "bytes_per_launch": 2.0 * 1024.0 * f + 1024.0 * w}
    doc = {"method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; mean per "
                     "launch per (kernel, grid size); FETCH_SIZE doubled (gfx950 tallies 128-byte "
                     "read requests as 64 bytes), WRITE_SIZE as reported, both in KB",
           "per_kernel": res}
    if algorithmic:
        doc["algorithmic_bytes_per_launch"] = float(algorithmic)
    doc.update(step_average(res))
    with open(out, "w") as fh:
        json.dump(doc, fh, indent=1)


def step_average(res):
    """The figure bench.py reports.  One 30-iteration step of the benchmark launches the
    iteration kernel once searching every point (<false, false, 4, false>), once with the
    shortcut for points whose correspondence provably did not change while moving the source
    into its working copy (<true, true, 4, false>) and 28 times with the shortcut, in place
    (<true, true, 4, true>)."""
    it = {k: v for k, v in res.items() if "icp_iterate_kernel" in k}
    if not it:
        return {}
    big = max(int(k.split("grid=")[1]) for k in it)
    pick = lambda tag: [v for k, v in it.items() if tag in k and k.endswith(f"grid={big}")]
    first, second, rest = pick("<false, false, 4, false>"), pick("<true, true, 4, false>"), pick("<true, true, 4, true>")
    if not (first and second and rest):
        return {}
    return {"icp_iterate_kernel_bytes_per_launch":
                (first[0]["bytes_per_launch"] + second[0]["bytes_per_launch"]
                 + 28.0 * rest[0]["bytes_per_launch"]) / 30.0,
            "weighting": "1 launch <false, false, 4, false> : 1 launch <true, true, 4, false> : "
                         "28 launches <true, true, 4, true>, as in one step"}


def regimes(pass_dirs, out, traffic_out=None):
    """Counter passes over tools/profile_step.py -> means per (kernel, regime).  A launch of
    the iteration kernel is placed by its position within its step (the launches between two
    solve_kernel launches): 0-2 search every point of the benchmark pair, 8-29 are settled."""
    per = defaultdict(lambda: defaultdict(list))            # (kernel, regime) -> counter -> values
    for d in pass_dirs:
        for f in find(d, "counter_collection.csv"):
            rows = defaultdict(dict)
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    rows[int(r["Dispatch_Id"])].setdefault("name", short(r["Kernel_Name"]))
                    rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
            pos = 0
            for i in sorted(rows):
                row = rows[i]
                name = row.pop("name")
                regime = ""
                if "icp_iterate_kernel" in name:
                    regime = ("searches every point (launches 0-2 of a step)" if pos <= 2 else
                              "settled (launches 8-29 of a step)" if pos >= 8 else "in between (launches 3-7)")
                    name = "svicp::icp_iterate_kernel"
                    pos += 1
                elif "solve_kernel" in name and "batch" not in name and "p2p" not in name and "cal_" not in name:
                    pos = 0
                if not (name.startswith("svicp::icp_iterate_kernel") or "nn1_kernel" in name or "cal_eval" in name
                        or "gather_probe_kernel" in name):
                    continue
                for cname, v in row.items():
                    per[(name, regime)][cname].append(v)
    doc = {"method": "rocprofv3 --pmc passes (one run of tools/profile_step.py each) over the 1 M-point "
                     "benchmark pair; means per launch.  bytes = 2 * 1024 * FETCH_SIZE + 1024 * WRITE_SIZE "
                     "(gfx950 tallies a 128-byte read request as 64 bytes; both counters are in KB).  "
                     "ta_busy_fraction = TA_TA_BUSY_sum / (32 * GRBM_GUI_ACTIVE): the TA counter sums over "
                     "the 256 texture-address units, GRBM_GUI_ACTIVE over the 8 XCDs.  "
                     "valu_busy_fraction = SQ_ACTIVE_INST_VALU / (32 * GRBM_GUI_ACTIVE): quad-cycles with a "
                     "vector-ALU instruction in flight, summed over the 1024 SIMDs, over the SIMD cycles of the launch.  "
                     "l2_hit_rate = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum).  SQ_WAIT_* and "
                     "SQ_WAVE_CYCLES are in quad-cycles summed over waves.",
           "kernels": {}}
    flat = {}
    for (name, regime), cs in sorted(per.items()):
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        entry = {"launches": max(len(v) for v in cs.values()), "counters": m}
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            entry["hbm_bytes_per_launch"] = 2048.0 * m["FETCH_SIZE"] + 1024.0 * m["WRITE_SIZE"]
        if "TA_TA_BUSY_sum" in m and m.get("GRBM_GUI_ACTIVE"):
            entry["ta_busy_fraction"] = m["TA_TA_BUSY_sum"] / (32.0 * m["GRBM_GUI_ACTIVE"])
        if "SQ_ACTIVE_INST_VALU" in m and m.get("GRBM_GUI_ACTIVE"):
            entry["valu_busy_fraction"] = m["SQ_ACTIVE_INST_VALU"] / (32.0 * m["GRBM_GUI_ACTIVE"])
        if m.get("TCC_HIT_sum") is not None and (m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0)) > 0:
            entry["l2_hit_rate"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
        if m.get("SQ_WAVE_CYCLES"):
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if c in m:
                    entry[c.lower() + "_share_of_wave_cycles"] = m[c] / m["SQ_WAVE_CYCLES"]
        if m.get("SQ_WAVES") and m.get("SQ_INSTS_VMEM"):
            entry["vector_memory_instructions_per_wave"] = m["SQ_INSTS_VMEM"] / m["SQ_WAVES"]
        doc["kernels"][f"{name} [{regime}]" if regime else name] = entry
        flat[(name, regime)] = entry
    with open(out, "w") as fh:
        json.dump(doc, fh, indent=1)
    if traffic_out:
        get = lambda reg: next((e.get("hbm_bytes_per_launch") for (n, r), e in flat.items()
                                if n == "svicp::icp_iterate_kernel" and r.startswith(reg)), None)
        full, mid, settled = get("searches"), get("in between"), get("settled")
        t = {"method": doc["method"], "source": os.path.basename(out),
             "icp_iterate_full_search_bytes_per_launch": full,
             "icp_iterate_settled_bytes_per_launch": settled,
             "algorithmic_bytes_per_launch": 24.0 * 1048576 + 28.0 * 1048576}
        if full and mid and settled:
            t["icp_iterate_kernel_bytes_per_launch"] = (3 * full + 5 * mid + 22 * settled) / 30.0
            t["weighting"] = "3 launches that search every point : 5 in between : 22 settled, as in one step"
        for (n, r), e in flat.items():
            if "nn1_kernel<false" in n and "hbm_bytes_per_launch" in e:
                t["nn1_capped_bytes_per_launch"] = e["hbm_bytes_per_launch"]
        with open(traffic_out, "w") as fh:
            json.dump(t, fh, indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "regimes":
        # regimes <out json> <traffic json> <pass dir> [<pass dir> ...]
        regimes(sys.argv[4:], sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "traffic":
        traffic(*sys.argv[2:6])
