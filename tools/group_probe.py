"""Development probe: the capped standalone search with a query per lane against the A/B variant
with a query per group of 8 (and 4) lanes: equal answers, times at the start pose and registered."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stereovision_amd as sv
from stereovision_amd import synth
tgt, nrm, src, P = synth.make_pair(1024)
registered = np.ascontiguousarray(((src.astype(np.float64) - P[:3, 3]) @ P[:3, :3]).astype(np.float32))
c = sv.Context(0)
c.set_target(tgt, nrm)
for name, cloud in (("start pose", src), ("registered", registered)):
    c.set_source(cloud)
    os.environ.pop("SVICP_NN1_GROUP", None)
    ref = c.correspondences()
    t_ref = c.time_nn(False, 50) * 1e3
    line = f"{name}: lane per query {t_ref:.2f} us"
    for G in (8, 4):
        os.environ["SVICP_NN1_GROUP"] = str(G)
        got = c.correspondences()
        same = np.array_equal(ref[0], got[0]) and np.array_equal(ref[1], got[1])
        line += f"; group of {G}: {c.time_nn(False, 50) * 1e3:.2f} us, answers {'equal' if same else 'DIFFER'}"
    print(line)
