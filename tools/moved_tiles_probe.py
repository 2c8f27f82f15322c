"""Development probe: per late iteration, how many points change their fp32 position, in how
many 1024-point tiles (in the library's storage order), and the largest count in one tile;
then the per-iteration times of an eager run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stereovision_amd as sv
from stereovision_amd import synth
tgt, nrm, src, _ = synth.make_pair(1024)
c = sv.Context(0)
c.set_target(tgt, nrm)
prev = None
for k in (4, 5, 6, 7, 8, 9, 10, 11, 15, 16, 20, 21, 29, 30):
    c.set_source(src)
    c.run(k)
    cur = c.get_source()
    if prev is not None and prev[0] == k - 1:
        moved = np.any(cur != prev[1], axis=1)
        # caller order -> rough tiles by sorting along the Morton-like order is not available
        # here: count per 1024 consecutive points of the caller's order as a stand-in
        per = moved.reshape(-1, 1024).sum(1)
        print(f"update {k - 1}: {moved.sum()} points move; caller-order rows of 1024 with movers: "
              f"{(per > 0).sum()}, largest {per.max()}, rows with more than 64: {(per > 64).sum()}")
    prev = (k, cur)
c.set_source(src)
c.set_options(time_kernels=1)
its = []
for _ in range(3):
    c.run(30)
    its.append(c.iteration_times()[0] * 1e3)
print("us per iteration (eager, event pair included)", np.round(np.mean(its, axis=0), 1))
