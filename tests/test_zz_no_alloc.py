"""svicp_run_enqueue (the entry point that allocates nothing) against svicp_run, through the
C ABI: two calls in a row on one context, the second with a smaller source, each compared with
svicp_run on the same inputs and with the CPU oracle.  The workspace is a file-scope device
array that keeps the first call's contents, so the second call also proves that what must
start at zero is cleared.  Tolerance: the transform within 1e-9 (the sums are 64-bit fixed
point and order-independent on both paths; the bound leaves room for nothing but that), status
and inlier count exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _enqueue(ctx, iters):
    """One call and the wait for it.  "Allocates nothing" is checked on the device's own count
    of free memory (hipMemGetInfo: it sees a hipMalloc or hipFree inside the library, which
    torch's allocator statistics would not)."""
    import torch
    out = torch.full((20,), float("nan"), dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    free_before, _ = torch.cuda.mem_get_info(0)
    ctx.run_enqueue(iters, out.data_ptr())
    ctx.sync()
    free_after, _ = torch.cuda.mem_get_info(0)
    assert free_after == free_before, (free_before, free_after)
    return out.cpu().numpy()


def test_enqueue_matches_run_twice_in_a_row(po):
    import stereovision_amd as sv
    from stereovision_amd import synth

    tgt, nrm, src, _ = synth.make_pair(96, pair=3)
    small = np.ascontiguousarray(src[: 40 * 96 + 17])          # ragged last tile, fewer tiles
    tree = po.KDTree(tgt)
    with sv.Context(0) as ref, sv.Context(0) as ctx:
        ref.set_target(tgt, nrm)
        ctx.set_target(tgt, nrm)
        for cloud, iters in ((src, 12), (small, 7)):
            ref.set_source(cloud)
            T_ref, info_ref = ref.run(iters)
            moved_ref = ref.get_source()
            ctx.set_source(cloud)
            out = _enqueue(ctx, iters)
            T = out[:16].reshape(4, 4)
            assert np.abs(T - T_ref).max() < 1e-9
            assert out[16] == 0 and out[17] == info_ref.inliers
            assert abs(out[18] - info_ref.weighted_sq_residual) <= 1e-9 * abs(info_ref.weighted_sq_residual)
            assert abs(out[19] - info_ref.step_norm) < 1e-9
            assert np.array_equal(ctx.get_source(), moved_ref)
            expect = po.icp_point2plane(tree, nrm, cloud, iters)
            assert np.abs(T - expect["trans"]).max() < 1e-8
            with pytest.raises(sv.SvicpError):
                ctx.history()                                     # the history stays in the workspace


def test_enqueue_recorded_in_a_graph_and_replayed(po):
    """The callers this entry point is for: the loop recorded by a stream capture (torch's, in
    its global capture mode, where an allocation or a synchronising call fails the capture) and
    launched as a graph.  Device-to-host copies between the launches and a source set again
    (same size: the recording holds the tile count) must not change what a launch computes."""
    import torch
    import stereovision_amd as sv
    from stereovision_amd import synth

    tgt, nrm, src, _ = synth.make_pair(96, pair=3)
    with sv.Context(0) as ref, sv.Context(0) as ctx:
        ref.set_target(tgt, nrm)
        ref.set_source(src)
        T_ref, info_ref = ref.run(9)
        ctx.set_target(tgt, nrm)
        ctx.set_source(src)
        out = torch.full((20,), float("nan"), dtype=torch.float64, device="cuda:0")
        stream = torch.cuda.Stream()
        ctx.set_stream(stream.cuda_stream)
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                ctx.run_enqueue(9, out.data_ptr())
            for launch in range(5):
                if launch == 3:
                    ctx.set_source(src)
                out.fill_(float("nan"))
                torch.cuda.synchronize()
                with torch.cuda.stream(stream):
                    graph.replay()
                torch.cuda.synchronize()
                got = out.cpu().numpy()
                assert np.abs(got[:16].reshape(4, 4) - T_ref).max() < 1e-9, launch
                assert got[16] == 0 and got[17] == info_ref.inliers, launch
            del graph
        finally:
            ctx.use_own_stream()


def test_enqueue_rejects_what_exceeds_the_workspace(synth32):
    import stereovision_amd as sv
    with sv.Context(0) as ctx:
        ctx.set_target(synth32["target"], synth32["normals"])
        ctx.set_source(synth32["source"])
        import torch
        out = torch.zeros(20, dtype=torch.float64, device="cuda:0")
        for bad in (0, 129):
            with pytest.raises(sv.SvicpError) as e:
                ctx.run_enqueue(bad, out.data_ptr())
            assert e.value.status == sv._lib.ERR_ARG
        with pytest.raises(sv.SvicpError):
            ctx.run_enqueue(5, 0)
        # more points than the workspace's 4096 tiles hold
        big = np.zeros((2050 * 2050, 3), np.float32)
        big[:, 0] = np.arange(len(big), dtype=np.float32) * 1e-3
        ctx.set_source(big)
        with pytest.raises(sv.SvicpError) as e:
            ctx.run_enqueue(5, out.data_ptr())
        assert e.value.status == sv._lib.ERR_ARG and "4194304" in str(e.value)
        ctx.set_source(synth32["source"])
        # svicp_run still works on the context afterwards
        ctx.run_enqueue(5, out.data_ptr())
        ctx.sync()
        T, _ = ctx.run(5)
        assert np.abs(out[:16].cpu().numpy().reshape(4, 4) - T).max() < 1e-9


def test_two_contexts_enqueued_back_to_back_on_their_own_streams(po):
    """Each context launches on a non-blocking stream of its own and the call returns without
    waiting, so two contexts enqueued one after the other from one thread run at the same time.
    Each has a workspace of its own inside the file-scope array: both results equal svicp_run's."""
    import torch
    import stereovision_amd as sv
    from stereovision_amd import synth

    tgt, nrm, src, _ = synth.make_pair(160, pair=5)
    other = np.ascontiguousarray(synth.make_pair(160, pair=6)[2])
    with sv.Context(0) as ref, sv.Context(0) as a, sv.Context(0) as b:
        expect = []
        for cloud in (src, other):
            ref.set_target(tgt, nrm)
            ref.set_source(cloud)
            expect.append(ref.run(25))
        a.set_target(tgt, nrm); a.set_source(src)
        b.set_target(tgt, nrm); b.set_source(other)
        out_a = torch.full((20,), float("nan"), dtype=torch.float64, device="cuda:0")
        out_b = torch.full((20,), float("nan"), dtype=torch.float64, device="cuda:0")
        for rep in range(3):
            a.set_source(src); b.set_source(other)
            torch.cuda.synchronize()
            a.run_enqueue(25, out_a.data_ptr())          # (returns at once: b's launches overlap a's)
            b.run_enqueue(25, out_b.data_ptr())
            a.sync(); b.sync()
            for out, (T_ref, info_ref) in ((out_a, expect[0]), (out_b, expect[1])):
                got = out.cpu().numpy()
                assert np.abs(got[:16].reshape(4, 4) - T_ref).max() < 1e-9, rep
                assert got[16] == 0 and got[17] == info_ref.inliers, rep


def test_a_ninth_context_finds_no_workspace_until_one_is_returned(synth32):
    import torch
    import stereovision_amd as sv
    out = torch.zeros(20, dtype=torch.float64, device="cuda:0")
    held = []
    try:
        for _ in range(9):
            c = sv.Context(0)
            held.append(c)
            c.set_target(synth32["target"], synth32["normals"])
            c.set_source(synth32["source"])
        T_ref, _ = held[0].run(5)
        served, refused = [], []
        for c in held:
            try:
                c.run_enqueue(5, out.data_ptr())
                c.sync()
                assert np.abs(out[:16].cpu().numpy().reshape(4, 4) - T_ref).max() < 1e-9
                served.append(c)
            except sv.SvicpError as e:
                assert e.status == sv._lib.ERR_ARG and "workspace" in str(e)
                refused.append(c)
        # eight workspaces per device: at most eight contexts are served, at least one is refused
        assert 1 <= len(served) <= 8 and len(refused) >= 1
        # a context that is refused stays usable otherwise
        T, _ = refused[0].run(5)
        assert np.abs(T - T_ref).max() < 1e-9
        served[0].close()
        refused[0].run_enqueue(5, out.data_ptr())        # the returned workspace serves it now
        refused[0].sync()
        assert np.abs(out[:16].cpu().numpy().reshape(4, 4) - T_ref).max() < 1e-9
    finally:
        for c in held:
            c.close()


def test_enqueue_with_the_solve_in_a_launch_of_its_own():
    """A source of exactly 4096 tiles: the largest the workspace takes, and the size from which
    the iteration kernel reads R and T left by a separate solve launch."""
    import torch
    import stereovision_amd as sv
    from stereovision_amd import synth

    tgt, nrm, src, _ = synth.make_pair(2048, inlier_frac=0.7)
    assert len(src) == 4096 * 1024
    with sv.Context(0) as ref, sv.Context(0) as ctx:
        ref.set_target(tgt, nrm); ref.set_source(src)
        T_ref, info_ref = ref.run(8)
        moved_ref = ref.get_source()
        ctx.set_target(tgt, nrm); ctx.set_source(src)
        out = _enqueue(ctx, 8)
        assert np.abs(out[:16].reshape(4, 4) - T_ref).max() < 1e-9
        assert out[16] == 0 and out[17] == info_ref.inliers
        assert np.array_equal(ctx.get_source(), moved_ref)
