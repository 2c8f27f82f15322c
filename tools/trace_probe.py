"""Development probe (apply tools/trace_instrumentation.patch, build with -DSVICP_TRACE=1, point SVICP_LIBRARY at it): where does the time
of one iteration go?  Prints, per iteration, the median over blocks of each phase boundary
relative to the earliest block start, in microseconds."""
import ctypes as C
import sys
sys.path.insert(0, ".")
import numpy as np
import stereovision_amd as sv
from stereovision_amd import synth, _lib
tgt, nrm, src, _ = synth.make_pair(1024)
c = sv.Context(0)
c.set_target(tgt, nrm)
c.set_source(src)
c.run(30)
c.set_source(src)
c.run(30)
lib = _lib.load()
buf = np.zeros((64, 1024, 8), np.uint64)
lib.svicp_debug_trace.argtypes = [C.c_void_p]
assert lib.svicp_debug_trace(buf.ctypes.data) == 0
names = ["start", "solved", "moved", "certified", "queued", "searched", "done"]
for it in (0, 1, 2, 3, 4, 5, 10, 20, 29):
    t = buf[it, :, :7].astype(np.float64) * 0.01
    t0 = t[:, 0].min()
    t[t == 0] = np.nan
    rel = t - t0
    med = np.nanmedian(rel, axis=0)
    mx = np.nanmax(rel, axis=0)
    print(f"iter {it:2d} median:", " ".join(f"{n}={v:6.2f}" for n, v in zip(names, med)))
    print(f"        max   :", " ".join(f"{n}={v:6.2f}" for n, v in zip(names, mx)))
    if it in (20, 25):
        tot = buf[it, :, 7].astype(np.int64)
        skipped = np.isnan(rel[:, 6])
        print(f"        tiles skipped: {skipped.sum()}, with queued points: {(tot > 0).sum()} (points {tot.sum()})")
        for label, sel in (("no queued points", (~skipped) & (tot == 0)), ("queued points   ", tot > 0)):
            if sel.any():
                d = np.diff(rel[sel][:, :7], axis=1)
                print(f"        {label}: phase durations median", " ".join(f"{v:5.2f}" for v in np.median(d, axis=0)),
                      "| max", " ".join(f"{v:5.2f}" for v in d.max(axis=0)),
                      "| done median %.2f max %.2f" % (np.median(rel[sel][:, 6]), rel[sel][:, 6].max()))
    if it + 1 < 64:
        nxt = buf[it + 1, :, 0].astype(np.float64) * 0.01
        print(f"        next kernel's first block starts at {nxt.min() - t0:6.2f}")

buf2 = np.zeros((64, 8), np.uint64)
lib.svicp_debug_trace2.argtypes = [C.c_void_p]
assert lib.svicp_debug_trace2(buf2.ctypes.data) == 0
d = np.diff(buf2[5:29, :7].astype(np.float64) * 0.01, axis=1)
print("solve phases (loads+sums, scaling, factorisation, back substitution, sincos, R to LDS), us:",
      " ".join(f"{v:.2f}" for v in np.median(d, axis=0)))
