"""Development probe: several host threads, each with its own context, registering at the same time."""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stereovision_amd as sv
from stereovision_amd import synth
pairs = [synth.make_pair(96 + 16 * i, pair=i) for i in range(4)]
serial = []
for tgt, nrm, src, _ in pairs:
    with sv.Context(0) as c:
        c.set_target(tgt, nrm); c.set_source(src)
        serial.append(c.run(12)[0])
out = [None] * 4
def work(i):
    tgt, nrm, src, _ = pairs[i]
    res = []
    for rep in range(15):
        with sv.Context(0) as c:
            c.set_target(tgt, nrm)
            c.set_source(src)
            res.append(c.run(12)[0])
            c.knn(src[:200], 5)
    out[i] = res
th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
[t.start() for t in th]; [t.join() for t in th]
ok = all(np.array_equal(r, serial[i]) for i in range(4) for r in out[i])
print("4 threads x 15 contexts each: results equal the serial ones bit for bit:", ok)
