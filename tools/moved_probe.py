"""Development probe: fraction of points whose fp32 position changes per late iteration."""
import sys
sys.path.insert(0, ".")
import numpy as np
import stereovision_amd as sv
from stereovision_amd import synth
tgt, nrm, src, _ = synth.make_pair(1024)
c = sv.Context(0)
c.set_target(tgt, nrm)
prev = None
for k in (5, 6, 7, 10, 11, 20, 21, 29, 30):
    c.set_source(src)
    c.run(k)
    cur = c.get_source()
    if prev is not None and prev[0] == k - 1:
        moved = np.any(cur != prev[1], axis=1)
        print(f"update applied at iteration {k - 1}: {moved.mean() * 100:.3f}% of the points move")
    prev = (k, cur)
