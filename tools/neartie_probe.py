"""Development probe: among the points that still move late in a registration, how close the
two nearest target points are to a tie, and how close the nearest is to the edge of the probe."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stereovision_amd as sv
from stereovision_amd import synth
tgt, nrm, src, _ = synth.make_pair(1024)
c = sv.Context(0)
c.set_target(tgt, nrm)
c.set_source(src); c.run(20); a = c.get_source()
c.set_source(src); c.run(21); b = c.get_source()
print("searched", c.search_counts().astype(int)[15:21])
moved = np.any(a != b, axis=1)
q = np.ascontiguousarray(b[moved])
idx, d2 = c.knn(q, 2)
d = np.sqrt(d2.astype(np.float64))
gap = (d[:, 1] - d[:, 0]) / d[:, 0]
print("movers", moved.sum(), "relative gap below 1e-6, 1e-5, 1e-4, 1e-3:", [(gap < g).sum() for g in (1e-6, 1e-5, 1e-4, 1e-3)])
cov = 1.001 * np.sqrt(5.0)
edge = np.abs(d[:, 0] - cov) / cov
print("nearest within 1e-5, 1e-4, 1e-3 (relative) of the covered radius:", [(edge < g).sum() for g in (1e-5, 1e-4, 1e-3)])
print("no target point within the covered radius:", (d[:, 0] > cov).sum(), " second beyond it:", (d[:, 1] > cov).sum())
step = np.sqrt(((a[moved].astype(np.float64) - b[moved]) ** 2).sum(1))
print("moves: median %.2e max %.2e" % (np.median(step), step.max()))
