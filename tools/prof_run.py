"""Small fixed workload for rocprofv3: the 1M-point benchmark pair, a few runs."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stereovision_amd as sv  # noqa: E402
from stereovision_amd import synth  # noqa: E402

side = int(os.environ.get("SIDE", "1024"))
tgt, nrm, src, P = synth.make_pair(side)
ctx = sv.Context(0, use_graph=int(os.environ.get("GRAPH", "0")))
ctx.set_target(tgt, nrm)
ctx.set_source(src)
for _ in range(int(os.environ.get("RUNS", "3"))):
    ctx.run(30)
ctx.time_nn(False, 5)
ctx.close()
