"""Development probe: searched points and time per iteration on the 16.8M-point pair."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import stereovision_amd as sv
from stereovision_amd import synth
side = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
tgt, nrm, src, _ = synth.make_pair(side)
c = sv.Context(0)
c.set_target(tgt, nrm)
c.set_source(src)
c.run(30)
c.set_source(src)
t0 = time.perf_counter(); c.run(30); print("30 iterations: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
S = c.search_counts()
X, I = c.history()
c.set_options(time_kernels=1)
c.set_source(src)
c.run(30)
t = c.iteration_times()[0]
for k in range(30):
    print(k, "searched %9d (%.2f%%)" % (S[k], 100 * S[k] / len(src)), "%.1f us" % (t[k] * 1e3), "step %.3g" % I[k, 2])
