"""Development: the last exact search of a kernel trace, launch by launch (us)."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
last = max(i for i, (n, _) in enumerate(seq) if "nn1_kernel<true" in n)
out = []
run = []
for n, d in seq[last:]:
    if "nn1" not in n: break
    run.append((n, d))
for n, d in run:
    short = "first" if "nn1_kernel" in n else "box" if "far" in n else "brute" if "brute" in n else "?"
    out.append(f"{short} {d:.0f}")
print(", ".join(out), " sum %.0f" % sum(d for _, d in run))
