"""Development probe: standalone 1-NN kernel times and the loop."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stereovision_amd as sv
from stereovision_amd import synth
tgt, nrm, src, _ = synth.make_pair(1024)
c = sv.Context(0)
c.set_target(tgt, nrm)
c.set_source(src)
print("capped 1-NN: %.2f us, exact: %.2f us" % (c.time_nn(False, 50) * 1e3, c.time_nn(True, 20) * 1e3))
c.set_options(coherent=0)
c.run(30)
t0 = time.perf_counter()
for _ in range(5): c.run(30)
print("full-search iteration: %.2f us" % ((time.perf_counter() - t0) / 150 * 1e6))
