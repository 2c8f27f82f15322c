import sys; sys.path.insert(0,".")
import numpy as np, stereovision_amd as sv
from stereovision_amd import synth
for side, iters in ((320, 250), (1024, 120)):
    tgt,nrm,src,_=synth.make_pair(side, inlier_frac=0.7)
    c=sv.Context(0); c.set_target(tgt,nrm)
    out=[]
    for coh in (0,1):
        c.set_options(coherent=coh); c.set_source(src); T,info=c.run(iters); X,I=c.history(); out.append((T,X,I,c.search_counts(),c.get_source()))
    (T0,X0,I0,s0,p0),(T1,X1,I1,s1,p1)=out
    print(side, iters, "max|dX|", np.abs(X0-X1).max(), "inliers equal", np.array_equal(I0[:,0],I1[:,0]), "positions equal", np.array_equal(p0,p1), "searched late", s1[-40:].astype(int).sum(), "of", s0[-40:].astype(int).sum())
    c.close()
