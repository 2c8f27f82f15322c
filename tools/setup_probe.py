"""Development probe: set-up costs at 1M points (grid build, normal estimation, k-NN)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import stereovision_amd as sv
from stereovision_amd import synth
tgt, nrm, src, _ = synth.make_pair(1024)
c = sv.Context(0)
for label, normals in (("given normals", nrm), ("estimated normals (20-NN + eigenvector)", None)):
    for rep in range(2):
        t0 = time.perf_counter(); c.set_target(tgt, normals); dt = time.perf_counter() - t0
    print(f"set_target, {label}: {dt * 1e3:.1f} ms")
est = c.get_normals()
dot = np.abs((est * nrm).sum(1))
print("estimated vs analytic normals: median |cos| %.5f, 1%% quantile %.4f" % (np.median(dot), np.quantile(dot, 0.01)))
for k in (1, 8, 20):
    t0 = time.perf_counter(); idx, d2 = c.knn(src, k); dt = time.perf_counter() - t0
    print(f"knn k={k}: {dt * 1e3:.1f} ms incl. copies ({len(src) / dt / 1e6:.1f} M queries/s)")
t0 = time.perf_counter(); c.set_source(src); print("set_source: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
