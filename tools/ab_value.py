"""Development probe: the benchmark step (1 M points, 30 iterations) on the library named by
SVICP_LIBRARY: loop time per iteration over several rounds, per-iteration times of an eager run."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stereovision_amd as sv
from stereovision_amd import synth
tgt, nrm, src, _ = synth.make_pair(1024)
c = sv.Context(0)
c.set_target(tgt, nrm)
c.set_source(src)
for _ in range(3):
    c.run(30)
rounds = []
for r in range(5):
    t0 = time.perf_counter()
    for _ in range(20):
        c.run(30)
    rounds.append((time.perf_counter() - t0) / 600 * 1e6)
c.set_options(time_kernels=1)
its = []
for _ in range(5):
    c.run(30)
    its.append(c.iteration_times()[0] * 1e3)
it = np.mean(its, axis=0)
print(os.environ.get("SVICP_LIBRARY", "tree"), "us/iter rounds", np.round(rounds, 2), "first", np.round(it[:5], 1), "settled", round(float(it[8:].mean()), 2))
