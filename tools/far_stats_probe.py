"""Development probe (library built with -DSVICP_FAR_STATS): what the shared-box exact search does
per wave at the start pose and at registered positions."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stereovision_amd as sv
from stereovision_amd import synth
tgt, nrm, src, P = synth.make_pair(1024)
c = sv.Context(0)
c.set_target(tgt, nrm)
lib = sv.load()
names = ["waves", "rounds", "rows", "points staged", "lanes in", "lanes settled", "give-ups", "lanes left to the rings", "blocks 10ns", "pieces 10ns", "copy 10ns", "scan 10ns", "-", "-", "-", "-"]
def stats(tag):
    out = (C.c_ulonglong * 16)()
    lib.svicp_debug_far_stats(out, 1)
    v = list(out)
    print(tag, {n: x for n, x in zip(names, v)}, "per wave: rounds %.2f rows %.1f points %.1f" % (v[1] / max(v[0], 1), v[2] / max(v[0], 1), v[3] / max(v[0], 1)))
c.set_source(src)
stats("clear")
ms = c.time_nn(True, 1)
stats("start pose (2 launches)")
print("exact at start ms", c.time_nn(True, 5)); stats("-")
registered = np.ascontiguousarray(((src.astype(np.float64) - P[:3, 3]) @ P[:3, :3]).astype(np.float32))
c.set_source(registered)
ms = c.time_nn(True, 1)
stats("registered (2 launches)")
print("exact registered ms", c.time_nn(True, 5))
