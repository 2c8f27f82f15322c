"""Development probe: searched points per iteration and loop time."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stereovision_amd as sv
from stereovision_amd import synth
tgt, nrm, src, _ = synth.make_pair(1024)
c = sv.Context(0)
c.set_target(tgt, nrm)
c.set_source(src)
c.run(30)
print(c.search_counts()[3:].astype(int))
ts = []
for _ in range(20):
    t0 = time.perf_counter(); c.run(30); ts.append(time.perf_counter() - t0)
print("30 iterations: %.1f us" % (sorted(ts)[10] * 1e6))
