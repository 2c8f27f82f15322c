"""Development probe: the host call svicp_knn with k = 1 and k = 2 on a million queries (upload,
search, download), registered positions and the start pose."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stereovision_amd as sv
from stereovision_amd import synth
tgt, nrm, src, _ = synth.make_pair(1024)
c = sv.Context(0)
c.set_target(tgt, nrm)
c.set_source(src)
c.run(30)
moved = c.get_source()
for name, q in (("registered", moved), ("start pose", src)):
    for k in (1, 2):
        c.knn(q, k)
        t0 = time.perf_counter()
        for _ in range(3):
            i, d = c.knn(q, k)
        print("%s k=%d: %.2f ms per call, checksum %d %.6g" % (name, k, (time.perf_counter() - t0) / 3 * 1e3, int(i[:, 0].sum()), float(d[:, 0].sum())))
