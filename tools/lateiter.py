"""Cost of one converged iteration in graph mode: (time(60 iters) - time(30 iters)) / 30."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stereovision_amd as sv
from stereovision_amd import synth
side = int(os.environ.get("SIDE", "1024"))
tgt, nrm, src, P = synth.make_pair(side)
ctx = sv.Context(0)
ctx.set_target(tgt, nrm)
ctx.set_source(src)
res = {}
for it in (5, 30, 60, 120):
    ctx.run(it)
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); ctx.run(it); ts.append(time.perf_counter() - t0)
    res[it] = sorted(ts)[len(ts) // 2]
    print(it, "iters: %.1f us total, %.2f us/iter" % (res[it] * 1e6, res[it] * 1e6 / it))
print("late iteration: %.2f us" % ((res[120] - res[60]) / 60 * 1e6))
print("first 5 iterations: %.2f us each" % (res[5] / 5 * 1e6))
