"""Development probe: estimated normals of the synthetic surface against its analytic ones, both modes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, stereovision_amd as sv
from stereovision_amd import synth
tgt, nrm, _, _ = synth.make_pair(256)
c = sv.Context(0)
for cn in (0, 1):
    c.set_options(centred_normals=cn); c.set_target(tgt)
    d = np.abs((c.get_normals() * nrm).sum(1))
    print(cn, float(np.median(d)), float((d > 0.99).mean()))
