"""Development probe: one registration of a very large pair (side^2 points each), checked against its generating pose."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stereovision_amd as sv
from stereovision_amd import synth
side = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
t0 = time.perf_counter()
tgt, nrm, src, P = synth.make_pair(side, inlier_frac=0.7)
print(f"{len(src)} points generated in {time.perf_counter() - t0:.1f} s", flush=True)
c = sv.Context(0)
t0 = time.perf_counter(); c.set_target(tgt, nrm); print(f"set_target {time.perf_counter() - t0:.2f} s", flush=True)
t0 = time.perf_counter(); c.set_source(src); print(f"set_source {time.perf_counter() - t0:.2f} s", flush=True)
t0 = time.perf_counter(); T, info = c.run(30); dt = time.perf_counter() - t0
X, I = c.history()
s = c.search_counts()
print(f"30 iterations in {dt * 1e3:.1f} ms; inliers first/last {I[0, 0] / len(src):.3f} {I[-1, 0] / len(src):.3f}; "
      f"searched {s[:4].astype(np.int64)} .. {int(s[-1])}; |T P - I| {np.abs(T @ P - np.eye(4)).max():.2e}; status {info.status}", flush=True)
print("tiles left unread in iterations 4, 10, 29:", c.tiles_left_unread()[[4, 10, 29]], "of", (len(src) + 1023) // 1024, flush=True)
idx, d2 = c.correspondences()
print("matched share", float((idx >= 0).mean()))
t0 = time.perf_counter(); n2 = c.get_normals() if False else None
# normal estimation at this size (a fresh context so that the first one's buffers are gone)
c.close()
c = sv.Context(0)
t0 = time.perf_counter(); c.set_target(tgt); print(f"set_target with normal estimation {time.perf_counter() - t0:.2f} s", flush=True)
ne = c.get_normals()
dots = np.abs((ne * nrm).sum(1))
print("estimated normals against the surface's: median |cos|", float(np.median(dots)), "share > 0.9", float((dots > 0.9).mean()))
