"""Development probe for -DSVICP_TRACE builds: how evenly the tiles of a launch that searches every
point finish (percentiles of the tiles' last stamps, from the launch's first stamp, in us), and
the same per XCD."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stereovision_amd as sv
from stereovision_amd import synth, _lib
tgt, nrm, src, P = synth.make_pair(1024)
ctx = sv.Context(0)
ctx.set_target(tgt, nrm)
ctx.set_source(src)
lib = ctypes.CDLL(_lib.LIB_PATH)
ctx.run(8)
lib.svicp_debug_trace_clear()
ctx.run(8)
NI, NT, NS = 64, 1024, 16
buf = np.zeros(NI * NT * NS, dtype=np.uint64)
lib.svicp_debug_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert lib.svicp_debug_trace(buf.ctypes.data, buf.size) == 0
raw = buf.reshape(NI, NT, NS)
tr = raw[:, :, :8].astype(np.float64) * 0.01
for k in range(5):
    t = tr[k]
    t0 = t[:, 0][t[:, 0] > 0].min()
    last = np.where(t > 0, t, 0).max(axis=1) - t0
    entry = t[:, 0] - t0
    dur = last - entry
    xcc = raw[k, :, 9].astype(np.int64) & 15
    pct = lambda v: " ".join(f"{np.percentile(v, q):6.2f}" for q in (0, 10, 50, 90, 99, 100))
    print(f"iteration {k}: tiles end at (min p10 p50 p90 p99 max) {pct(last)} | start {pct(entry)} | own time {pct(dur)}")
    print("    per XCD, last end:", " ".join(f"{last[xcc == x].max():6.2f}" for x in range(8)), " mean own time:", " ".join(f"{dur[xcc == x].mean():6.2f}" for x in range(8)))
