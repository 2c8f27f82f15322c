"""Development probe: does torch still initialise its GPU context after the library ran?"""
import sys
sys.path.insert(0, ".")
import numpy as np
import stereovision_amd as sv
from stereovision_amd import synth
mode = sys.argv[1] if len(sys.argv) > 1 else "graph"
tgt, nrm, src, _ = synth.make_pair(64)
c = sv.Context(0, use_graph=1 if mode == "graph" else 0)
c.set_target(tgt, nrm)
c.set_source(src)
if mode != "none":
    print(c.run(5)[1].inliers)
import torch
print(torch.cuda.is_available(), torch.cuda.device_count())
x = torch.zeros(4, device="cuda:0")
print("torch ok", x.sum().item())
