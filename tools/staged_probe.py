"""Development probe: the LDS-staged capped 1-NN against the global-memory one."""
import os, sys
sys.path.insert(0, ".")
import numpy as np
import stereovision_amd as sv
from stereovision_amd import synth
for side in (1024, 300):
    tgt, nrm, src, _ = synth.make_pair(side)
    c = sv.Context(0)
    c.set_target(tgt, nrm)
    for label in ("start", "registered"):
        c.set_source(src)
        if label == "registered":
            c.run(30)
        os.environ.pop("SVICP_NN_STAGED", None)
        i0, d0 = c.correspondences()
        t0 = c.time_nn(False, 30)
        os.environ["SVICP_NN_STAGED"] = "1"
        i1, d1 = c.correspondences()
        t1 = c.time_nn(False, 30)
        os.environ.pop("SVICP_NN_STAGED", None)
        print(side, label, "equal:", np.array_equal(i0, i1) and np.array_equal(d0, d1),
              "hits %.1f%%" % (100 * (i0 >= 0).mean()), "global %.2f us, staged %.2f us" % (t0 * 1e3, t1 * 1e3))
