"""Development probe: when does the 1M-point benchmark pair reach a fixed point?"""
import sys
sys.path.insert(0, ".")
import numpy as np
import stereovision_amd as sv
from stereovision_amd import synth
side = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
tgt, nrm, src, pose = synth.make_pair(side)
c = sv.Context(0)
c.set_target(tgt, nrm)
c.set_source(src)
T, info = c.run(30)
X, I = c.history()
S = c.search_counts()
c.set_options(time_kernels=1)
c.set_source(src)
c.run(30)
times = c.iteration_times()[0]
for k in range(30):
    same = k > 0 and np.array_equal(X[k], X[k - 1])
    print(k, "same" if same else "    ", "%.3g" % np.abs(X[k]).max(), "searched %8d" % S[k], "%.1f us" % (times[k] * 1e3), I[k][0])
