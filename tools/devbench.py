"""Developer A/B harness: times the per-iteration kernels for several build
variants (SVICP_VARIANT) in one process on the 1M-point benchmark pair."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stereovision_amd as sv  # noqa: E402
from stereovision_amd import synth  # noqa: E402

side = int(os.environ.get("SIDE", "1024"))
variants = [int(v) for v in os.environ.get("VARIANTS", "0,2,4,8").split(",")]
bpcs = [int(v) for v in os.environ.get("BPC", "4").split(",")]
tgt, nrm, src, P = synth.make_pair(side)
coh = [int(v) for v in os.environ.get("COH", "0,1").split(",")]
ref_T = None
for rnd in range(int(os.environ.get("ROUNDS", "2"))):
    for v in variants:
        for bpc in coh:
            os.environ["SVICP_VARIANT"] = str(v)
            ctx = sv.Context(0, coherent=bpc)
            ctx.set_target(tgt, nrm)
            ctx.set_source(src)
            ctx.run(30)
            ctx.set_options(time_kernels=1)
            f = []
            s = []
            for _ in range(5):
                T, info = ctx.run(30)
                a, b, _ = ctx.kernel_times()
                f.append(a)
                s.append(b)
            fi, si = ctx.iteration_times()
            if rnd == 0:
                print('   per-iter fused us:', ' '.join(f'{x*1e3:.1f}' for x in fi[:8]), '...', f'{fi[-1]*1e3:.1f}')
            ctx.set_options(time_kernels=0)
            t0 = time.perf_counter()
            for _ in range(10):
                T, info = ctx.run(30)
            dt = (time.perf_counter() - t0) / 10
            if ref_T is None:
                ref_T = T
            nn = ctx.time_nn(False, 20)
            print(f"round {rnd} variant {v} coherent {bpc}: fused {np.median(f)*1e3:7.2f} us  solve {np.median(s)*1e3:6.2f} us  "
                  f"graph/iter {dt/30*1e6:7.2f} us  nn1 {nn*1e3:6.2f} us  dT {np.abs(T-ref_T).max():.1e}", flush=True)
            ctx.close()
