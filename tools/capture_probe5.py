"""svicp_run_enqueue recorded with the HIP runtime's own capture calls (no torch in the
process): replays with device-to-host copies between them, against svicp_run."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stereovision_amd as sv
from stereovision_amd import synth

hip = C.CDLL("libamdhip64.so")
def chk(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what}: hip error {rc}")
MODE = int(sys.argv[1]) if len(sys.argv) > 1 else 0      # hipStreamCaptureMode: 0 global, 1 thread local, 2 relaxed
COPIES = int(sys.argv[2]) if len(sys.argv) > 2 else 1
tgt, nrm, src, _ = synth.make_pair(96, pair=3)
with sv.Context(0) as ref, sv.Context(0) as ctx:
    ref.set_target(tgt, nrm); ref.set_source(src)
    T1, _ = ref.run(9)
    ctx.set_target(tgt, nrm); ctx.set_source(src)
    out = C.c_void_p()
    chk(hip.hipMalloc(C.byref(out), 160), "hipMalloc")
    s = C.c_void_p()
    SFLAGS = int(sys.argv[4]) if len(sys.argv) > 4 else 0      # 1: hipStreamNonBlocking
    chk(hip.hipStreamCreateWithFlags(C.byref(s), SFLAGS), "hipStreamCreateWithFlags")
    ctx.set_stream(s.value)
    graph, gexec = C.c_void_p(), C.c_void_p()
    chk(hip.hipStreamBeginCapture(s, MODE), "begin")
    ctx.run_enqueue(9, out.value)
    chk(hip.hipStreamEndCapture(s, C.byref(graph)), "end")
    FLAGS = int(sys.argv[3]) if len(sys.argv) > 3 else -1
    if FLAGS < 0:
        chk(hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, C.c_size_t(0)), "instantiate")
    else:
        chk(hip.hipGraphInstantiateWithFlags(C.byref(gexec), graph, C.c_ulonglong(FLAGS)), "instantiate with flags")
    host = np.empty(20, np.float64)
    for r in range(6):
        chk(hip.hipGraphLaunch(gexec, s), "launch")
        chk(hip.hipStreamSynchronize(s), "sync")
        for _ in range(COPIES):
            chk(hip.hipMemcpy(host.ctypes.data_as(C.c_void_p), out, C.c_size_t(160), 2), "copy")
        print(f"mode {MODE} copies {COPIES} replay {r + 1}: max diff {np.abs(host[:16].reshape(4, 4) - T1).max():.3g} inliers {host[17]:.3f}", flush=True)
    chk(hip.hipGraphExecDestroy(gexec), "exec destroy")
    chk(hip.hipGraphDestroy(graph), "graph destroy")
    ctx.use_own_stream()
    chk(hip.hipStreamDestroy(s), "stream destroy")
    chk(hip.hipFree(out), "free")
