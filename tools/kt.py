"""Prints per-dispatch durations of the iteration kernels from a rocprofv3 kernel trace."""
import csv, glob, sys
d = sys.argv[1]
f = sorted(glob.glob(d + '/**/*kernel_trace.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
out = []
for r in rows:
    n = r['Kernel_Name']
    tag = ('C' if 'icp_stream' in n else 'Q' if 'icp_search' in n else 'R' if 'reduce_rows' in n
           else 'S' if 'solve_kernel' in n else 'I' if 'icp_iterate' in n else None)
    if tag:
        out.append((tag, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
k = int(sys.argv[2]) if len(sys.argv) > 2 else 48
print(' '.join(f"{t}{d:.1f}" for t, d in out[-119:][:k]))
