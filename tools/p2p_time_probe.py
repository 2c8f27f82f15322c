"""Development probe: time per iteration of the point-to-point variant on the 1 M-point pair."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stereovision_amd as sv
from stereovision_amd import synth
tgt, nrm, src, P = synth.make_pair(1024)
c = sv.Context(0)
c.set_target(tgt, nrm)
c.set_source(src)
T, info = c.run_point2point(30)
best = 1e9
for _ in range(8):
    c.set_source(src)
    c.sync()
    t0 = time.perf_counter()
    c.run_point2point(30)
    best = min(best, time.perf_counter() - t0)
print(f"{os.environ.get('SVICP_LIBRARY', 'in-tree')}: {best / 30 * 1e6:.2f} us per iteration (best of 8 runs of 30), inliers {info.inliers}")
