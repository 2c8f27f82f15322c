"""Development probe: the exact standalone search at the start pose, for a kernel trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stereovision_amd as sv
from stereovision_amd import synth
tgt, nrm, src, P = synth.make_pair(1024)
c = sv.Context(0)
c.set_target(tgt, nrm)
c.set_source(src)
print("exact at start ms", c.time_nn(True, 5))
