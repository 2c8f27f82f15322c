"""Development harness: times the calibration evaluation at the 2000-view size."""
import sys, time
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stereovision_amd import synth
from stereovision_amd.calib import ReprojectionEvaluator

V = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
w, l, r, _, start = synth.make_stereo_views(views=V)
ev = ReprojectionEvaluator(w, l, r)
ev.set_params(start)
for reps in (5, 50, 200):
    print(f"V={V} reps={reps}: {ev.time_eval(reps) * 1e3:.1f} us per evaluation")
