"""Development probe: svicp_run against svicp_run_enqueue around the tile count from which the
solve has a launch of its own (4096), run twice each: which results are the same bits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import stereovision_amd as sv
from stereovision_amd import synth

tgt, nrm, src, _ = synth.make_pair(2048, inlier_frac=0.7)
for n in (4096 * 1024, 4095 * 1024, 4096 * 1024 - 5):
    cloud = np.ascontiguousarray(src[:n])
    res = []
    with sv.Context(0) as ref, sv.Context(0) as ctx:
        ref.set_target(tgt, nrm); ctx.set_target(tgt, nrm)
        for rep in range(2):
            ref.set_source(cloud)
            T, info = ref.run(8)
            res.append(("run", T, info.inliers))
        ref.set_options(use_graph=0)
        ref.set_source(cloud)
        T, info = ref.run(8)
        res.append(("run eager", T, info.inliers))
        out = torch.zeros(20, dtype=torch.float64, device="cuda:0")
        for rep in range(2):
            ctx.set_source(cloud)
            ctx.run_enqueue(8, out.data_ptr()); ctx.sync()
            o = out.cpu().numpy()
            res.append(("enqueue", o[:16].reshape(4, 4), o[17]))
        ctx.set_source(cloud)
        T, info = ctx.run(8)
        res.append(("run on the enqueue context", T, info.inliers))
    print("n =", n, "tiles", (n + 1023) // 1024)
    for name, T, inl in res:
        print(f"  {name:28s} inliers {inl:.0f}  max|T - first| {np.abs(T - res[0][1]).max():.3e}")
