"""Development probe: the loop as a HIP graph against plain stream launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stereovision_amd as sv
from stereovision_amd import synth
tgt, nrm, src, P = synth.make_pair(1024)
for g in (1, 0, 1, 0):
    ctx = sv.Context(0, use_graph=g)
    ctx.set_target(tgt, nrm); ctx.set_source(src)
    for it in (30, 120):
        ctx.run(it)
        ts = []
        for _ in range(10):
            t0 = time.perf_counter(); ctx.run(it); ts.append(time.perf_counter() - t0)
        print(f"use_graph={g} {it} iterations: {sorted(ts)[5] * 1e6:.1f} us")
    ctx.close()
