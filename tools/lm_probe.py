"""Development probe: time of the refinement loop (2000 views) and of its parts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stereovision_amd import synth, calib
V = 2000
cw, cl, cr, truth, start = synth.make_stereo_views(views=V)
ev = calib.ReprojectionEvaluator(cw, cl, cr, device=0)
ev.evaluate(start, 5)
t0 = time.perf_counter(); b = ev.evaluate(start, 5, want_residual=False); t1 = time.perf_counter()
step, _ = calib.solve_damped(b, 5, V, 1e-3); t2 = time.perf_counter()
print(f"evaluate + fetch {1e3 * (t1 - t0):.2f} ms, host solve {1e3 * (t2 - t1):.2f} ms")
t0 = time.perf_counter()
out = calib.lm_refine(ev, start, 5, max_iter=20)
print(f"lm_refine: {1e3 * (time.perf_counter() - t0):.1f} ms", [type(o).__name__ for o in (out if isinstance(out, tuple) else (out,))])
