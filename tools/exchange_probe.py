"""Development probe: per-iteration inlier counts and sums of weights of a run that searches
every point in every iteration against a run with the shortcuts, for several cloud sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stereovision_amd as sv
from stereovision_amd import synth
for side in (40, 100, 195, 320, 512, 1024):
    tgt, nrm, src, _ = synth.make_pair(side)
    c = sv.Context(0)
    c.set_target(tgt, nrm)
    out = []
    for coh in (0, 1):
        c.set_options(coherent=coh)
        c.set_source(src)
        T, info = c.run(30)
        X, I = c.history()
        out.append((T, X, I, c.search_counts()))
    (T0, X0, I0, s0), (T1, X1, I1, s1) = out
    d = I1[:, 0] - I0[:, 0]
    bad = np.nonzero(d)[0]
    print(side, "tiles", (len(src) + 1023) // 1024, "first differing iteration", bad[:1], "d inliers", d[bad[:6]],
          "max|dX|", np.abs(X0 - X1).max(), "searched", s1[:12].astype(int))

    if np.abs(X0 - X1).max() > 1e-10:
        print("   |dX| per iteration", np.array2string(np.abs(X0 - X1).max(1), precision=1))
        print("   d sum w b^2       ", np.array2string(I1[:, 1] - I0[:, 1], precision=2))
        print("   d sum w           ", np.array2string(I1[:, 3] - I0[:, 3], precision=2))
