"""Development probe: the drop-in call as a user of the reference makes it (host buffers in,
transform and moved source out) at 1M points: set_source, run(30), get_source."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stereovision_amd as sv
from stereovision_amd import synth
tgt, nrm, src, _ = synth.make_pair(1024)
c = sv.Context(0)
c.set_target(tgt, nrm)
for rep in range(3):
    t0 = time.perf_counter(); c.set_source(src); t1 = time.perf_counter()
    c.run(30); t2 = time.perf_counter()
    out = c.get_source(); t3 = time.perf_counter()
    print(f"set_source {1e3*(t1-t0):.2f} ms, run(30) {1e3*(t2-t1):.2f} ms, get_source {1e3*(t3-t2):.2f} ms, total {1e3*(t3-t0):.2f} ms")

from stereovision_amd.icp import pinned_empty
psrc = pinned_empty(src.shape); psrc[:] = src
out = pinned_empty(src.shape)
for rep in range(3):
    t0 = time.perf_counter(); c.set_source(psrc); t1 = time.perf_counter()
    c.run(30); t2 = time.perf_counter()
    c.get_source(out); t3 = time.perf_counter()
    print(f"pinned: set_source {1e3*(t1-t0):.2f} ms, run(30) {1e3*(t2-t1):.2f} ms, get_source {1e3*(t3-t2):.2f} ms, total {1e3*(t3-t0):.2f} ms")
