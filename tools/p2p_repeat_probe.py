"""run_point2point several times in a row on one context (its recorded loop is launched again
each time, with a device-to-host copy after each): every run must give the first one's bits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch
import stereovision_amd as sv
from stereovision_amd import synth
tgt, nrm, src, _ = synth.make_pair(96, pair=3)
with sv.Context(0) as ctx:
    ctx.set_target(tgt, nrm); ctx.set_source(src)
    T0 = None
    for r in range(6):
        T, info = ctx.run_point2point(9)
        T0 = T.copy() if T0 is None else T0
        print(f"run {r + 1}: max diff to run 1 {np.abs(T - T0).max():.3g} inliers {info.inliers:.3f}", flush=True)
