"""Development probe: a digest of everything a 30-iteration registration of the 1 M-point pair returns
(transform, X history, statistics, moved source, correspondences), to compare between runs and between boxes."""
import hashlib, os, socket, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stereovision_amd as sv
from stereovision_amd import synth
tgt, nrm, src, _ = synth.make_pair(1024)
h = hashlib.sha256()
with sv.Context(0) as c:
    c.set_target(tgt, nrm); c.set_source(src)
    T, info = c.run(30)
    X, I = c.history()
    idx, d2 = c.correspondences()
    for a in (T, X, I, c.get_source(), idx, d2):
        h.update(np.ascontiguousarray(a).tobytes())
h2 = hashlib.sha256()
with sv.Context(0) as c:
    c.set_target(tgt); n = c.get_normals()
    h2.update(n.tobytes())
import torch
print("device", torch.cuda.get_device_properties(0).name, "uuid-ish", torch.cuda.get_device_properties(0).total_memory)
print("registration digest", h.hexdigest()[:32])
print("estimated-normals digest", h2.hexdigest()[:32])
