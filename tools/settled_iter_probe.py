"""Development probe: kernel time of each settled launch of the benchmark step (events around eager
launches minus an empty event pair) beside the number of points it searched; the launches with and
without a search apart.  The library is the one named by SVICP_LIBRARY."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stereovision_amd as sv
from stereovision_amd import synth
tgt, nrm, src, _ = synth.make_pair(1024)
c = sv.Context(0); c.set_target(tgt, nrm); c.set_source(src)
for _ in range(3): c.run(30)
c.set_options(time_kernels=1)
its = []
for _ in range(8):
    c.run(30); f, s, _ = c.kernel_times(); its.append(c.iteration_times()[0] * 1e3 - s * 1e3)
it = np.mean(its, axis=0)
cnt = c.search_counts()
print(os.environ.get("SVICP_LIBRARY", "tree"))
print("us  ", np.round(it[5:], 1))
print("cnt ", cnt[5:].astype(int))
print("mean searched>0: %.2f  none: %.2f  all[5:]: %.2f" % (it[5:][cnt[5:] > 0].mean(), it[5:][cnt[5:] == 0].mean(), it[5:].mean()))
