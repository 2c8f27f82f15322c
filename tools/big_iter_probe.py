"""Development probe: per-iteration kernel times and searched points of the 16.8 M-point pair."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stereovision_amd as sv
from stereovision_amd import synth
side = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
tgt, nrm, src, P = synth.make_pair(side, inlier_frac=0.7)
c = sv.Context(0, time_kernels=1, use_graph=0)
c.set_target(tgt, nrm)
c.set_source(src)
c.run(30)
c.set_source(src)
T, info = c.run(30)
a, b = c.iteration_times()
s = c.search_counts()
print("us per iteration", np.round(a * 1e3, 1))
print("solve launch us ", np.round(b * 1e3, 1))
print("searched        ", s.astype(np.int64))
print("sum us", a.sum() * 1e3, "pose error", np.abs(T @ P - np.eye(4)).max())
