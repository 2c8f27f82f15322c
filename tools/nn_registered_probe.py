"""Development probe: the standalone capped 1-NN kernel and its loads alone (gather probe) on
the REGISTERED benchmark pair, as bench.py's roofline_correspondence measures them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stereovision_amd as sv
from stereovision_amd import synth
tgt, nrm, src, _ = synth.make_pair(1024)
c = sv.Context(0)
c.set_target(tgt, nrm)
c.set_source(src)
c.run(30)
nn = c.time_nn(exact=False, reps=50)
ex = c.time_nn(exact=True, reps=20)
loads, instr = c.time_gather(True, 0, 20)
print(os.environ.get("SVICP_LIBRARY", "tree"), "capped %.2f us  exact %.2f us  own loads %.2f us  wave-level loads %d" % (nn * 1e3, ex * 1e3, loads * 1e3, instr))
