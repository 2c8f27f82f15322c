"""Development probe: contexts, batches and communicators made and destroyed in a loop; device memory must come back."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import stereovision_amd as sv
from stereovision_amd import synth, sharded
from stereovision_amd.calib import ReprojectionEvaluator

free0 = None
tgt, nrm, src, _ = synth.make_pair(192)
w, l, r, _, start = synth.make_stereo_views(views=64)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    with sv.Context(0) as c:
        n = 1000 + 700 * (rep % 13)
        c.set_target(tgt[: 20000 + n], nrm[: 20000 + n] if rep % 2 else None)
        c.set_source(src[: 15000 + n])
        c.run(3 + rep % 5)
        c.knn(src[:100], 1 + rep % 40)
        if rep % 4 == 0:
            sharded.init_comm(c)
            c.shard_run(4)
            c.comm_destroy()
        if rep % 5 == 0:
            c.run_point2point(3)
    b = sv.ArenaBatch([tgt[:5000], tgt[:9000]], [nrm[:5000], nrm[:9000]], [src[:4000], src[:9000]])
    b.run(4)
    b.close()
    ev = ReprojectionEvaluator(w, l, r)
    ev.set_params(start)
    ev.eval()
    ev.solve_damped(1e-3)
    ev.close() if hasattr(ev, "close") else None
    del ev
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info(0)
    if rep == 3:
        free0 = free
    if rep % 8 == 7:
        print(f"rep {rep}: free {free / 2**20:.0f} MiB")
print("change of free device memory between repetition 3 and the end:", (free - free0) / 2**20, "MiB")
