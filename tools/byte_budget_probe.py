"""Development probe for the byte budget: the launches whose traffic is budgeted, and nothing
else, on the library named by SVICP_LIBRARY (a -DSVICP_DROP build or the product).  Runs under
rocprofv3 --pmc (tools/byte_budget.sh): the 1 M-point benchmark pair, every point searched in
every iteration (coherent = 0, so the launches from iteration 2 on are one kernel variant in one
regime), then the standalone capped search."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stereovision_amd as sv  # noqa: E402
from stereovision_amd import synth  # noqa: E402

import numpy as np  # noqa: E402

tgt, nrm, src, P = synth.make_pair(1024)
# the source at its registered position, from the generating pose: src = R x + t
registered = np.ascontiguousarray(((src.astype(np.float64) - P[:3, 3]) @ P[:3, :3]).astype(np.float32))
c = sv.Context(0, coherent=0, use_graph=0)
c.set_target(tgt, nrm)


def run3():
    # Launches 0 and 1 as the product runs them; launch 2 (the third of a run, every point
    # searched, coordinates updated in place) is the one a -DSVICP_DROP build takes a stream out
    # of and the one the counters are read for.  (What such a build computes in that launch is
    # nonsense: a solve that fails afterwards is not an error here.)
    c.set_source(src)
    try:
        c.run(3)
    except sv.SvicpError as e:
        print("  (", e, ")")


for _ in range(6):
    run3()
c.set_options(time_kernels=1)
run3()
print("iteration ms", [round(float(x), 4) for x in c.iteration_times()[0]])
c.set_options(time_kernels=0)
c.set_source(registered)
print("nn1 capped ms, registered positions", c.time_nn(False, 8))
c.close()
