"""Development probe: batched registration (64 pairs of 65536 points), device time per run."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stereovision_amd as sv
from stereovision_amd import synth
pairs = int(os.environ.get("PAIRS", "64"))
ctxs = []
for i in range(pairs):
    t, nrm, s, _ = synth.make_pair(256, pair=i, pose=synth.small_pose(i))
    c = sv.Context(0); c.set_target(t, nrm); c.set_source(s); ctxs.append(c)
b = sv.Batch(ctxs)
for it in (30, 60):
    b.run(it)
    ms = []
    for _ in range(5):
        b.run(it); ms.append(b.last_ms())
    print(f"{pairs} pairs, {it} iterations: {sorted(ms)[2]:.3f} ms per run, {sorted(ms)[2] / it * 1e3:.1f} us per iteration")
