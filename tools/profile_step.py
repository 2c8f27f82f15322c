"""The program the counter passes profile (tools/gpu_round.sh): the benchmark pair at its fixed
start pose, a few steps of 30 iterations (so that the position of a launch within its step says
which regime it is in: launches 0-2 search every point, launches 8-29 are settled), the
standalone capped search at the registered positions, its loads alone (the gather probe),
and calibration evaluations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stereovision_amd as sv
from stereovision_amd import synth
from stereovision_amd.calib import HipCalEngine

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
tgt, nrm, src, _ = synth.make_pair(1024)
c = sv.Context(0)
c.set_target(tgt, nrm)
c.set_source(src)
for _ in range(steps):
    c.run(30)
c.time_nn(False, 8)
c.time_nn(True, 3)
c.time_gather(True, 0, 8)      # the capped search's loads alone, same positions
c.close()
w, l, r, _, start = synth.make_stereo_views(views=2000)
eng = HipCalEngine(w, l, r, "cuda:0")
eng.set_params(start, 5)
eng.ev.time_eval(10)
