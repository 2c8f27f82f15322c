import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, time
import stereovision_amd as sv
from stereovision_amd import synth
tgt,_,src,pose=synth.make_pair(1024)
c=sv.Context(0,dist_threshold=15.0); c.set_target(tgt,tgt); c.set_source(src)
T,info=c.run_point2point(20)
RT,I=c.history_rt()
np.set_printoptions(linewidth=200,suppress=True)
print(I)
print(T@pose)
c.set_source(src)
t=time.perf_counter()
for _ in range(5): c.run_point2point(20)
print('ms per iteration', (time.perf_counter()-t)/100*1e3)
