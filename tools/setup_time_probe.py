"""Development probe: host time of set_target (given normals) with the device drained before and
after, buffers in place, for the library named by SVICP_LIBRARY."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import stereovision_amd as sv
from stereovision_amd import synth
for side in (1024, 256):
    tgt, nrm, src, _ = synth.make_pair(side)
    c = sv.Context(0)
    c.set_target(tgt, nrm); torch.cuda.synchronize()
    ts = []
    for _ in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        c.set_target(tgt, nrm); c.synchronize() if hasattr(c, "synchronize") else None
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(os.environ.get("SVICP_LIBRARY", "tree"), side, "set_target ms", np.round(ts, 2))
    c.close()
