"""Development probe: device time of an arena batch for two run lengths (the difference is the settled iterations)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stereovision_amd as sv
from stereovision_amd import synth
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 128
side = 256
tg, nr, sr = [], [], []
t0, n0, s0, _ = synth.make_pair(side, pair=1)
rng = np.random.default_rng(5)
for i in range(pairs):
    sh = rng.normal(size=3).astype(np.float32) * 0.3
    tg.append(t0); nr.append(n0); sr.append(s0 + sh)
b = sv.ArenaBatch(tg, nr, sr, device=0)
for iters in (10, 30, 10, 30, 30):
    T, infos = b.run(iters)
    print(f"{pairs} pairs x {side * side} points, {iters} iterations: {b.last_ms():.3f} ms on the device")
print("checksum", float(np.abs(T).sum()))
