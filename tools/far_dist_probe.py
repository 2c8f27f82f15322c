"""Development probe: how far the benchmark pair's source points are from the target at the start
pose, in cell edges (the exact search's work grows with the cube of that)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stereovision_amd as sv
from stereovision_amd import synth
tgt, nrm, src, P = synth.make_pair(1024)
c = sv.Context(0)
c.set_target(tgt, nrm)
print("grid", c.grid_info())
idx, d2 = c.knn(src, 1)
d = np.sqrt(d2.reshape(-1).astype(np.float64))
h = float(np.sqrt(5.0)) * 1.002
q = d / h
print("distance / sqrt(threshold): percentiles 10 50 90 99 99.9 100:", np.round(np.percentile(q, [10, 50, 90, 99, 99.9, 100]), 2))
print("share beyond 1, 2, 4, 8, 16 probe radii:", [float((q > t).mean()) for t in (1, 2, 4, 8, 16)])
# per wave of 64 consecutive queries in the library's order is not visible here; per 64 in caller order:
w = q[: len(q) // 64 * 64].reshape(-1, 64)
print("waves (caller order) whose farthest query is beyond 2, 4, 8 radii:", [float((w.max(1) > t).mean()) for t in (2, 4, 8)])
