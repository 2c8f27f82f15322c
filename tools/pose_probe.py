"""Development probe: the benchmark pairs at the pose that leaves 70 % of the source within the threshold."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stereovision_amd as sv
from stereovision_amd import synth
for side in (1024, 4096):
    for frac in (None, 0.7):
        tgt, nrm, src, P = synth.make_pair(side, inlier_frac=frac)
        c = sv.Context(0)
        c.set_target(tgt, nrm)
        c.set_source(src)
        T, info = c.run(30)
        X, I = c.history()
        s = c.search_counts()
        t0 = time.perf_counter()
        for _ in range(5):
            c.run(30)
        loop = (time.perf_counter() - t0) / 150 * 1e6
        err = np.abs(T @ P - np.eye(4)).max()
        print(f"side {side} frac {frac}: {loop:.2f} us/iter; inliers it0 {I[0,0]/len(src):.3f} last {I[-1,0]/len(src):.3f}; "
              f"searched share {np.round(s[:8]/len(src),3)} last {s[-1]:.0f}; |X| {np.array2string(I[[0,1,2,4,8,16,29],2], precision=2)}; |T P - I| {err:.2e}")
        c.close()
