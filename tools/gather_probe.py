"""Development probe: the capped search against the gather probe (its memory instructions alone)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stereovision_amd as sv
from stereovision_amd import synth
tgt, nrm, src, _ = synth.make_pair(1024)
c = sv.Context(0)
c.set_target(tgt, nrm)
c.set_source(src)
for where in ("start", "registered"):
    nn = c.time_nn(False, 30) * 1e3
    real, ins = c.time_gather(True, 0, 30)
    print(f"{where}: capped 1-NN {nn:.1f} us; its loads alone {real * 1e3:.1f} us, {ins:.0f} wave instructions "
          f"({ins / (len(src) / 64):.1f} per wave of queries, {ins / real / 1e6:.1f} G/s)")
    for stride in (2, 4, 8):
        r, ins2 = c.time_gather(True, stride, 30)
        print(f"   every {stride}th lane only: {r * 1e3:.1f} us, {ins2:.0f} wave instructions")
    for nbytes in (12, 8, 4):
        r, _ = c.time_gather(nbytes, 0, 30)
        print(f"   the same addresses with {nbytes}-byte records: {r * 1e3:.1f} us")
    for pq in (8, 16, 24):
        r, ins2 = c.time_gather(False, pq, 30)
        print(f"   {pq} hashed loads per query: {r * 1e3:.1f} us, {ins2 / r / 1e6:.1f} G wave instructions/s")
    c.run(30)
