"""Crossed cases: X = the HIP runtime's capture calls with a torch tensor as the output and
torch's device-to-host copy between replays; Y = torch's graph with a hipMalloc'd output and
hipMemcpy between replays."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import stereovision_amd as sv
from stereovision_amd import synth

hip = C.CDLL("libamdhip64.so")
def chk(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what}: hip error {rc}")
CASE = sys.argv[1]
tgt, nrm, src, _ = synth.make_pair(96, pair=3)
with sv.Context(0) as ref, sv.Context(0) as ctx:
    ref.set_target(tgt, nrm); ref.set_source(src)
    T1, _ = ref.run(9)
    ctx.set_target(tgt, nrm); ctx.set_source(src)
    host = np.empty(20, np.float64)
    if CASE == "X":
        out = torch.full((20,), float("nan"), dtype=torch.float64, device="cuda:0")
        s = C.c_void_p()
        chk(hip.hipStreamCreate(C.byref(s)), "hipStreamCreate")
        ctx.set_stream(s.value)
        graph, gexec = C.c_void_p(), C.c_void_p()
        chk(hip.hipStreamBeginCapture(s, 0), "begin")
        ctx.run_enqueue(9, out.data_ptr())
        chk(hip.hipStreamEndCapture(s, C.byref(graph)), "end")
        chk(hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, C.c_size_t(0)), "instantiate")
        for r in range(5):
            chk(hip.hipGraphLaunch(gexec, s), "launch")
            torch.cuda.synchronize()
            host[:] = out.cpu().numpy()
            print(f"X replay {r + 1}: max diff {np.abs(host[:16].reshape(4, 4) - T1).max():.3g} inliers {host[17]:.3f}", flush=True)
    elif CASE == "Z":      # raw capture calls on a stream torch made
        out = C.c_void_p()
        chk(hip.hipMalloc(C.byref(out), 160), "hipMalloc")
        ts = torch.cuda.Stream()
        s = C.c_void_p(ts.cuda_stream)
        ctx.set_stream(s.value)
        graph, gexec = C.c_void_p(), C.c_void_p()
        chk(hip.hipStreamBeginCapture(s, 0), "begin")
        ctx.run_enqueue(9, out.value)
        chk(hip.hipStreamEndCapture(s, C.byref(graph)), "end")
        FLAGS = int(sys.argv[2]) if len(sys.argv) > 2 else 1
        chk(hip.hipGraphInstantiateWithFlags(C.byref(gexec), graph, C.c_ulonglong(FLAGS)), "instantiate")
        for r in range(5):
            chk(hip.hipGraphLaunch(gexec, s), "launch")
            chk(hip.hipDeviceSynchronize(), "sync")
            chk(hip.hipMemcpy(host.ctypes.data_as(C.c_void_p), out, C.c_size_t(160), 2), "copy")
            print(f"Z replay {r + 1}: max diff {np.abs(host[:16].reshape(4, 4) - T1).max():.3g} inliers {host[17]:.3f}", flush=True)
    elif CASE == "W":      # torch's graph on a stream the HIP runtime made
        out = C.c_void_p()
        chk(hip.hipMalloc(C.byref(out), 160), "hipMalloc")
        raw = C.c_void_p()
        chk(hip.hipStreamCreate(C.byref(raw)), "hipStreamCreate")
        s = torch.cuda.ExternalStream(raw.value)
        ctx.set_stream(raw.value)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            ctx.run_enqueue(9, out.value)
        for r in range(5):
            with torch.cuda.stream(s):
                g.replay()
            torch.cuda.synchronize()
            chk(hip.hipMemcpy(host.ctypes.data_as(C.c_void_p), out, C.c_size_t(160), 2), "copy")
            print(f"W replay {r + 1}: max diff {np.abs(host[:16].reshape(4, 4) - T1).max():.3g} inliers {host[17]:.3f}", flush=True)
    else:
        out = C.c_void_p()
        chk(hip.hipMalloc(C.byref(out), 160), "hipMalloc")
        s = torch.cuda.Stream()
        ctx.set_stream(s.cuda_stream)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            ctx.run_enqueue(9, out.value)
        for r in range(5):
            with torch.cuda.stream(s):
                g.replay()
            torch.cuda.synchronize()
            chk(hip.hipMemcpy(host.ctypes.data_as(C.c_void_p), out, C.c_size_t(160), 2), "copy")
            print(f"Y replay {r + 1}: max diff {np.abs(host[:16].reshape(4, 4) - T1).max():.3g} inliers {host[17]:.3f}", flush=True)
    ctx.use_own_stream()
