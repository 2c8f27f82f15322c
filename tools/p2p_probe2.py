"""Development probe: point-to-point ICP on a pair that starts close (it settles)."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, time
import stereovision_amd as sv
from stereovision_amd import synth
tgt, _, src, pose = synth.make_pair(1024, pose=synth.small_pose(3))
c = sv.Context(0, dist_threshold=15.0); c.set_target(tgt, tgt); c.set_source(src)
T, info = c.run_point2point(40)
RT, I = c.history_rt()
print("step of iterations 0, 5, 10, 20, 39:", [float(np.round(I[k, 2], 6)) for k in (0, 5, 10, 20, 39)])
for it in (20, 40):
    c.run_point2point(it)
    t = time.perf_counter()
    for _ in range(5): c.run_point2point(it)
    print(it, 'iterations: ms per iteration', (time.perf_counter() - t) / (5 * it) * 1e3)
