"""Reprojection residual / Jacobian evaluation of the calibration refinement on the GPU.

`ReprojectionEvaluator` wraps one svcal_ctx (include/svcal.h).  It evaluates
what the lambdas inside CStereoCameraCalibration::doStereoCalibrate
(Calibration/CStereoCameraCalibration.cpp:25-247) and
CMonocularCameraCalibration::refineAllPara
(Calibration/CMonocularCameraCalibration.cpp:737-906) evaluate, and returns
J^T J and J^T e block by block instead of the reference's dense Jacobian.
`assemble_dense` and `lm_refine` are host-side helpers for small problems: the
reference's Levenberg-Marquardt loop (CStereoCameraCalibration.cpp:278-335) on
top of the device evaluation.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import SvicpError

_dp = C.POINTER(C.c_double)


class _Blocks(C.Structure):
    _fields_ = [(n, _dp) for n in ("LL", "RR", "RS", "SS", "LE", "EE", "gL", "gR", "gE", "gS",
                                   "residual")] + [("sq_residual", C.c_double)]


SOLVE_SUMS = 90          # SVCAL_SOLVE_SUMS: a view's contribution to the shared system, nI x (nI + 1), nI <= 9

CAL_SYMBOLS = {
    "svcal_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "svcal_destroy": (C.c_int, [C.c_void_p]),
    "svcal_last_error": (C.c_char_p, [C.c_void_p]),
    "svcal_set_observations": (C.c_int, [C.c_void_p, _dp, C.c_int, _dp, _dp, C.c_int]),
    "svcal_set_params": (C.c_int, [C.c_void_p, _dp, C.c_int]),
    "svcal_eval": (C.c_int, [C.c_void_p]),
    "svcal_fetch": (C.c_int, [C.c_void_p, C.POINTER(_Blocks)]),
    "svcal_sq_residual": (C.c_int, [C.c_void_p, _dp]),
    "svcal_lm_refine": (C.c_int, [C.c_void_p, _dp, C.c_int, C.c_int, C.POINTER(C.c_int), _dp]),
    "svcal_set_update_intrinsics": (C.c_int, [C.c_void_p, C.c_int]),
    "svcal_solve_damped": (C.c_int, [C.c_void_p, C.c_double, _dp, _dp]),
    "svcal_solve_begin": (C.c_int, [C.c_void_p, C.c_double, C.c_void_p]),
    "svcal_solve_end": (C.c_int, [C.c_void_p, C.c_double, C.c_void_p, _dp, _dp]),
    "svcal_time_eval": (C.c_int, [C.c_void_p, C.c_int, _dp]),
    "svcal_set_stream": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "svcal_shard_export": (C.c_int, [C.c_void_p, C.c_void_p]),
    "svcal_shard_import": (C.c_int, [C.c_void_p, C.c_void_p]),
    "svcal_stereo_rectify": (C.c_int, [_dp, _dp, _dp, _dp, _dp, _dp, _dp]),
    "svcal_rectify_maps": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.c_int, _dp, _dp,
                                     _dp, _dp, _dp, _dp, _dp]),
}


def _load():
    lib = _lib.load()
    if not getattr(lib, "_cal_bound", False):
        for name, (res, args) in CAL_SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        lib._cal_bound = True
    return lib


def _d(a):
    return a.ctypes.data_as(_dp)


class ReprojectionEvaluator:
    def __init__(self, world_xy, left_xy, right_xy=None, device=0):
        self._lib = _load()
        self._h = C.c_void_p()
        rc = self._lib.svcal_create(int(device), C.byref(self._h))
        if rc != 0:
            raise SvicpError(rc, self._lib.svcal_last_error(None).decode())
        self.world = np.ascontiguousarray(world_xy, np.float64)
        self.left = np.ascontiguousarray(left_xy, np.float64)
        self.right = None if right_xy is None else np.ascontiguousarray(right_xy, np.float64)
        self.V, self.C = self.left.shape[0], self.left.shape[1]
        assert self.world.shape == (self.C, 2) and self.left.shape == (self.V, self.C, 2)
        assert self.right is None or self.right.shape == self.left.shape
        self._check(self._lib.svcal_set_observations(
            self._h, _d(self.world), self.C, _d(self.left),
            _d(self.right) if self.right is not None else None, self.V))
        self.k = 5

    def _check(self, rc):
        if rc != 0:
            raise SvicpError(rc, self._lib.svcal_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.svcal_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stereo(self):
        return self.right is not None

    def n_params(self, k=5):
        nI = 4 + k
        return 2 * nI + 6 * self.V + 6 if self.stereo else nI + 6 * self.V

    def set_params(self, params, k=5):
        p = np.ascontiguousarray(params, np.float64)
        if p.shape != (self.n_params(k),):
            raise ValueError(f"expected {self.n_params(k)} parameters, got {p.shape}")
        self.k = k
        self._check(self._lib.svcal_set_params(self._h, _d(p), k))

    def eval(self):
        """Launches the evaluation; results stay on the device until fetch()."""
        self._check(self._lib.svcal_eval(self._h))

    def fetch(self, want_residual=True):
        nI, V = 4 + self.k, self.V
        out = dict(LL=np.zeros((nI, nI)), gL=np.zeros(nI), LE=np.zeros((V, nI, 6)),
                   EE=np.zeros((V, 6, 6)), gE=np.zeros((V, 6)))
        if self.stereo:
            out.update(RR=np.zeros((nI, nI)), RS=np.zeros((nI, 6)), SS=np.zeros((6, 6)),
                       gR=np.zeros(nI), gS=np.zeros(6))
        if want_residual:
            out["residual"] = np.zeros(2 * V * self.C * (2 if self.stereo else 1))
        b = _Blocks()
        for name, arr in out.items():
            setattr(b, name, _d(arr))
        self._check(self._lib.svcal_fetch(self._h, C.byref(b)))
        out["sq_residual"] = b.sq_residual
        return out

    def sq_residual(self):
        """|e|^2 of the last eval()."""
        out = C.c_double(0.0)
        self._check(self._lib.svcal_sq_residual(self._h, C.byref(out)))
        return out.value

    def lm_refine(self, params, k=5, max_iter=100):
        """The reference's refinement loop inside the library (svcal_lm_refine): returns
        (refined parameters, final |e|, iterations)."""
        x = np.array(params, np.float64)
        its, rms = C.c_int(0), C.c_double(0.0)
        self._check(self._lib.svcal_lm_refine(self._h, _d(x), int(k), int(max_iter), C.byref(its), C.byref(rms)))
        self.k = int(k)
        return x, rms.value, its.value

    def set_update_intrinsics(self, update_intrinsics):
        """doStereoCalibrate's update_intrinsics: False freezes the 2 (4 + k) intrinsic and
        distortion parameters in the device step (CStereoCameraCalibration.cpp:187-245)."""
        self._check(self._lib.svcal_set_update_intrinsics(self._h, 1 if update_intrinsics else 0))

    def solve_damped(self, lam):
        """Device counterpart of solve_damped() below for the blocks of the last eval():
        (step in the reference's parameter order, step^T (lam step - G))."""
        step = np.empty(self.n_params(self.k), np.float64)
        pred = C.c_double(0.0)
        self._check(self._lib.svcal_solve_damped(self._h, C.c_double(lam), _d(step), C.byref(pred)))
        return step, pred.value

    def solve_begin(self, lam, dev_ptr):
        """First half of the damped step for views split over contexts: this context's views
        eliminated, their SOLVE_SUMS summed contributions copied to the device buffer."""
        self._check(self._lib.svcal_solve_begin(self._h, C.c_double(lam), C.c_void_p(int(dev_ptr))))

    def solve_end(self, lam, dev_ptr):
        """Second half, from the contributions added over all contexts: (this context's step,
        the shared parameters' part of step^T (lam step - G), its own views' part)."""
        step = np.empty(self.n_params(self.k), np.float64)
        pred = np.zeros(2, np.float64)
        self._check(self._lib.svcal_solve_end(self._h, C.c_double(lam), C.c_void_p(int(dev_ptr)),
                                              _d(step), _d(pred)))
        return step, float(pred[0]), float(pred[1])

    def evaluate(self, params, k=5, want_residual=True):
        self.set_params(params, k)
        self.eval()
        return self.fetch(want_residual)

    def set_stream(self, handle=None):
        """handle: a HIP stream handle (0 = the default stream); None: the context's own."""
        if handle is None:
            self._check(self._lib.svcal_set_stream(self._h, None, 1))
        else:
            self._check(self._lib.svcal_set_stream(self._h, C.c_void_p(int(handle)), 0))

    def shard_export(self, dev_ptr):
        self._check(self._lib.svcal_shard_export(self._h, C.c_void_p(int(dev_ptr))))

    def shard_import(self, dev_ptr):
        self._check(self._lib.svcal_shard_import(self._h, C.c_void_p(int(dev_ptr))))

    def time_eval(self, reps=20):
        ms = C.c_double()
        self._check(self._lib.svcal_time_eval(self._h, int(reps), C.byref(ms)))
        return ms.value


def assemble_dense(blocks, k, V, stereo=True):
    """Dense A = J^T J and G = J^T e in the reference's column order (small V only)."""
    nI = 4 + k
    if stereo:
        P = 2 * nI + 6 * V + 6
        A, G = np.zeros((P, P)), np.zeros(P)
        A[:nI, :nI] = blocks["LL"]
        A[nI:2 * nI, nI:2 * nI] = blocks["RR"]
        s0 = 2 * nI + 6 * V
        A[nI:2 * nI, s0:] = blocks["RS"]
        A[s0:, nI:2 * nI] = blocks["RS"].T
        A[s0:, s0:] = blocks["SS"]
        G[:nI], G[nI:2 * nI], G[s0:] = blocks["gL"], blocks["gR"], blocks["gS"]
        e0 = 2 * nI
    else:
        P = nI + 6 * V
        A, G = np.zeros((P, P)), np.zeros(P)
        A[:nI, :nI] = blocks["LL"]
        G[:nI] = blocks["gL"]
        e0 = nI
    for v in range(V):
        c = e0 + 6 * v
        A[:nI, c:c + 6] = blocks["LE"][v]
        A[c:c + 6, :nI] = blocks["LE"][v].T
        A[c:c + 6, c:c + 6] = blocks["EE"][v]
        G[c:c + 6] = blocks["gE"][v]
    return A, G


def solve_damped(blocks, k, V, lam, stereo=True, update_intrinsics=True):
    """step = -(A + lam diag(A))^-1 G without forming A: the per-view 6 x 6 blocks are
    eliminated first (Schur complement), which leaves a system in the 2 (4 + k) + 6 shared
    parameters (4 + k monocular).  Equal to the dense solve of the reference
    (CStereoCameraCalibration.cpp:288-291) up to rounding, at O(V) cost instead of O(V^3).
    update_intrinsics=False (:187-245): A holds the views' and the pose's columns only, i.e.
    the views' 6 x 6 blocks and the pose's 6 x 6 block, each solved alone; the intrinsic
    entries of the step are zero.
    Returns (step in the reference's parameter order, step^T (lam step - G))."""
    nI = 4 + k
    if not update_intrinsics:
        damp = lambda M: M + lam * M * np.eye(6)
        de = -np.linalg.solve(damp(blocks["EE"]), blocks["gE"][:, :, None])[:, :, 0]
        parts, grads = [np.zeros(2 * nI if stereo else nI), de.ravel()], [blocks["gE"].ravel()]
        if stereo:
            ds = -np.linalg.solve(damp(blocks["SS"]), blocks["gS"])
            parts.append(ds)
            grads.append(blocks["gS"])
        step, G = np.concatenate(parts), np.concatenate(grads)
        free = step[(2 * nI if stereo else nI):]
        return step, float(free @ (lam * free - G))
    nC = 2 * nI + 6 if stereo else nI
    Acc, gc = np.zeros((nC, nC)), np.zeros(nC)
    Acc[:nI, :nI] = blocks["LL"]
    gc[:nI] = blocks["gL"]
    if stereo:
        Acc[nI:2 * nI, nI:2 * nI] = blocks["RR"]
        Acc[nI:2 * nI, 2 * nI:] = blocks["RS"]
        Acc[2 * nI:, nI:2 * nI] = blocks["RS"].T
        Acc[2 * nI:, 2 * nI:] = blocks["SS"]
        gc[nI:2 * nI], gc[2 * nI:] = blocks["gR"], blocks["gS"]
    Acc = Acc + lam * np.diag(np.diag(Acc))
    EE = blocks["EE"] + lam * np.einsum("vij,ij->vij", blocks["EE"], np.eye(6))
    LE, gE = blocks["LE"], blocks["gE"]                         # (V, nI, 6), (V, 6)
    Einv_LEt = np.linalg.solve(EE, np.transpose(LE, (0, 2, 1)))  # (V, 6, nI)
    Einv_g = np.linalg.solve(EE, gE[:, :, None])[:, :, 0]        # (V, 6)
    S = Acc.copy()
    S[:nI, :nI] -= np.einsum("vaq,vqb->ab", LE, Einv_LEt)
    rhs = gc.copy()
    rhs[:nI] -= np.einsum("vaq,vq->a", LE, Einv_g)
    dc = -np.linalg.solve(S, rhs)
    de = -(Einv_g + np.einsum("vqa,a->vq", Einv_LEt, dc[:nI]))
    if stereo:
        step = np.concatenate([dc[:2 * nI], de.ravel(), dc[2 * nI:]])
        G = np.concatenate([gc[:2 * nI], gE.ravel(), gc[2 * nI:]])
    else:
        step = np.concatenate([dc, de.ravel()])
        G = np.concatenate([gc, gE.ravel()])
    return step, float(step @ (lam * step - G))


def lm_refine(evaluator, params, k=5, max_iter=100, verbose=False, dense=False, host_solve=False,
              update_intrinsics=True):
    """The reference's damped Gauss-Newton loop (CStereoCameraCalibration.cpp:278-335):
    lambda0 = 1e-3, A + lambda diag(A), gain-ratio update, stops on |step| < 1e-5,
    |e| < 1e-3 or 100 iterations.  One device evaluation per trial step.  The linear solve
    runs on the device, block-sparse (svcal_solve_damped), and only the step and |e|^2 come
    back; host_solve=True fetches the blocks and solves in numpy (solve_damped below, or
    densely with dense=True, small V only)."""
    x = np.array(params, np.float64)
    V, stereo = evaluator.V, evaluator.stereo
    if not (host_solve or dense) and hasattr(evaluator, "solve_damped"):
        evaluator.set_update_intrinsics(update_intrinsics)
        try:
            return _lm_refine_device(evaluator, x, k, max_iter, verbose)
        finally:
            evaluator.set_update_intrinsics(True)

    def solve(b, lam):
        if dense:
            A, G = assemble_dense(b, k, V, stereo)
            if not update_intrinsics:                       # the columns from offset 8 + 2 k on (:284-285)
                off = (2 if stereo else 1) * (4 + k)
                A, G = A[off:, off:], G[off:]
                free = -np.linalg.solve(A + lam * np.diag(np.diag(A)), G)
                return np.concatenate([np.zeros(off), free]), float(free @ (lam * free - G))
            step = -np.linalg.solve(A + lam * np.diag(np.diag(A)), G)
            return step, float(step @ (lam * step - G))
        return solve_damped(b, k, V, lam, stereo, update_intrinsics)

    b = evaluator.evaluate(x, k, want_residual=False)
    err2 = b["sq_residual"]
    lam, v = 1.0e-3, 2.0
    for it in range(max_iter):
        step, denom = solve(b, lam)
        if np.linalg.norm(step) < 1.0e-5:
            break
        nb = evaluator.evaluate(x + step, k, want_residual=False)
        r = (err2 - nb["sq_residual"]) / denom
        if verbose:
            print(f"iter {it}: |e| {np.sqrt(err2):.6g} -> {np.sqrt(nb['sq_residual']):.6g}, gain {r:.3g}")
        if r > 0:
            x = x + step
            b, err2 = nb, nb["sq_residual"]
            if np.sqrt(err2) < 1.0e-3:
                break
            lam *= max(1.0 / 3.0, 1.0 - (2.0 * r - 1.0) ** 3)
            v = 2.0
        else:
            lam *= v
            v *= 2.0
    return x, np.sqrt(err2)


def _lm_refine_device(evaluator, x, k, max_iter, verbose):
    """lm_refine with the blocks left on the device.  A rejected trial step costs one more
    evaluation (the device holds the trial's blocks, the loop needs the old ones back)."""
    def evaluate(p):
        evaluator.set_params(p, k)
        evaluator.eval()
        return evaluator.sq_residual()

    err2 = evaluate(x)
    lam, v = 1.0e-3, 2.0
    for it in range(max_iter):
        step, denom = evaluator.solve_damped(lam)
        if np.linalg.norm(step) < 1.0e-5:
            break
        new2 = evaluate(x + step)
        r = (err2 - new2) / denom
        if verbose:
            print(f"iter {it}: |e| {np.sqrt(err2):.6g} -> {np.sqrt(new2):.6g}, gain {r:.3g}")
        if r > 0:
            x = x + step
            err2 = new2
            if np.sqrt(err2) < 1.0e-3:
                break
            lam *= max(1.0 / 3.0, 1.0 - (2.0 * r - 1.0) ** 3)
            v = 2.0
        else:
            evaluate(x)
            lam *= v
            v *= 2.0
    return x, np.sqrt(err2)


def split_views(params, k, V, stereo, begin, end):
    """The parameter vector of views [begin, end) of a V-view problem."""
    nI = 4 + k
    e0 = 2 * nI if stereo else nI
    p = np.asarray(params, np.float64)
    parts = [p[:e0], p[e0 + 6 * begin:e0 + 6 * end]]
    if stereo:
        parts.append(p[e0 + 6 * V:])
    return np.concatenate(parts)


def view_bounds(V, world, rank):
    """Contiguous view range of a rank: sizes differ by at most one."""
    base, extra = divmod(V, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


class HipCalEngine:
    """One rank's views on one GPU.  export_sums / import_sums move the 272 sums over the
    rank's views between the library and a torch tensor on the same device."""

    def __init__(self, world_xy, left_xy, right_xy, device):
        import torch
        self.dev = torch.device(device)
        self.ev = ReprojectionEvaluator(world_xy, left_xy, right_xy, self.dev.index or 0)
        self.stereo = self.ev.stereo
        self.buf = torch.zeros(272, dtype=torch.float64, device=self.dev)
        self.cbuf = torch.zeros(SOLVE_SUMS, dtype=torch.float64, device=self.dev)
        self.ev.set_stream(torch.cuda.current_stream(self.dev).cuda_stream)

    def set_params(self, p, k):
        self.ev.set_params(p, k)

    def eval(self):
        self.ev.eval()

    def export_sums(self):
        self.ev.shard_export(self.buf.data_ptr())
        return self.buf

    def import_sums(self, t):
        self.ev.shard_import(t.data_ptr())

    def fetch(self):
        return self.ev.fetch(want_residual=False)

    def sq_residual(self):
        return self.ev.sq_residual()

    def solve_begin(self, lam):
        self.ev.solve_begin(lam, self.cbuf.data_ptr())
        return self.cbuf

    def solve_end(self, lam, t):
        return self.ev.solve_end(lam, t.data_ptr())


def _all_reduce_sum(t, group=None):
    """Sum over the ranks, in place: on the device with the nccl backend, through the host
    otherwise."""
    import torch.distributed as dist
    if t.is_cuda and dist.get_backend(group) != "nccl":
        host = t.cpu()
        dist.all_reduce(host, group=group)
        t.copy_(host)
    else:
        dist.all_reduce(t, group=group)
    return t


def evaluate_sharded(engine, params, k, V_total, begin, end, group=None, fetch=True):
    """One evaluation with the views split over the ranks of a torch.distributed group.
    The only exchange is one all-reduce of the engine's sums (RCCL on the device for the
    nccl backend; staged through the host for gloo).  Returns the shared blocks of the whole
    problem and the per-view blocks of this rank's views."""
    import torch.distributed as dist
    engine.set_params(split_views(params, k, V_total, engine.stereo, begin, end), k)
    engine.eval()
    t = _all_reduce_sum(engine.export_sums(), group)
    engine.import_sums(t)
    return engine.fetch() if fetch else None


def solve_damped_sharded(engine, lam, k, V_total, begin, end, group=None):
    """The damped step of the whole problem with the views split over the ranks, after
    evaluate_sharded.  Two exchanges: the views' contributions to the shared system
    (SOLVE_SUMS doubles) before it is solved, the same on every rank; then the step itself,
    every rank filling in its own views (and rank 0 the shared parameters), with the ranks'
    parts of step^T (lam step - G) in one more slot.  Returns (step of the whole problem,
    step^T (lam step - G)), the same on every rank."""
    import torch
    import torch.distributed as dist
    t = _all_reduce_sum(engine.solve_begin(lam), group)
    local, pred_shared, pred_views = engine.solve_end(lam, t)
    nI = 4 + k
    e0 = 2 * nI if engine.stereo else nI
    P = e0 + 6 * V_total + (6 if engine.stereo else 0)
    whole = np.zeros(P + 1)
    whole[e0 + 6 * begin:e0 + 6 * end] = local[e0:e0 + 6 * (end - begin)]
    whole[P] = pred_views
    if dist.get_rank(group) == 0:
        whole[:e0] = local[:e0]
        if engine.stereo:
            whole[e0 + 6 * V_total:P] = local[e0 + 6 * (end - begin):]
        whole[P] += pred_shared
    w = torch.from_numpy(whole)
    if dist.get_backend(group) == "nccl":
        w = w.to(t.device)
    dist.all_reduce(w, group=group)
    whole = w.cpu().numpy()
    return whole[:P].copy(), float(whole[P])


def lm_refine_sharded(engine, params, k, V_total, begin, end, group=None, max_iter=100, verbose=False):
    """lm_refine with the views split over the ranks: every rank runs the same loop on the
    whole parameter vector and evaluates and solves its own views; per trial step one
    all-reduce of 272 sums (evaluation), one of SOLVE_SUMS and one of the step (solve)."""
    x = np.array(params, np.float64)

    def evaluate(p):
        evaluate_sharded(engine, p, k, V_total, begin, end, group, fetch=False)
        return engine.sq_residual()

    err2 = evaluate(x)
    lam, v = 1.0e-3, 2.0
    for it in range(max_iter):
        step, denom = solve_damped_sharded(engine, lam, k, V_total, begin, end, group)
        if np.linalg.norm(step) < 1.0e-5:
            break
        new2 = evaluate(x + step)
        r = (err2 - new2) / denom
        if verbose:
            print(f"iter {it}: |e| {np.sqrt(err2):.6g} -> {np.sqrt(new2):.6g}, gain {r:.3g}")
        if r > 0:
            x = x + step
            err2 = new2
            if np.sqrt(err2) < 1.0e-3:
                break
            lam *= max(1.0 / 3.0, 1.0 - (2.0 * r - 1.0) ** 3)
            v = 2.0
        else:
            evaluate(x)
            lam *= v
            v *= 2.0
    return x, np.sqrt(err2)


def stereo_rectify_native(KL, KR, R, T):
    """svcal_stereo_rectify: doStereoRectify in the library (host arithmetic; no GPU needed)."""
    lib = _load()
    arrs = [np.ascontiguousarray(a, np.float64).reshape(-1) for a in (KL, KR, R, T)]
    lp, rp, Q = np.zeros(9), np.zeros(9), np.zeros(16)
    rc = lib.svcal_stereo_rectify(*[_d(a) for a in arrs], _d(lp), _d(rp), _d(Q))
    if rc != 0:
        raise _lib.SvicpError(rc, "svcal_stereo_rectify")
    return lp.reshape(3, 3), rp.reshape(3, 3), Q.reshape(4, 4)


def stereo_rectify(KL, KR, R, T):
    """CStereoCameraRectification::doStereoRectify (CStereoCameraRectification.cpp:13-58) on the
    host (a few 3 x 3 products): returns (left_proj, right_proj, Q)."""
    from scipy.spatial.transform import Rotation
    KL, KR = np.asarray(KL, np.float64).reshape(3, 3), np.asarray(KR, np.float64).reshape(3, 3)
    Rinv = np.linalg.inv(np.asarray(R, np.float64).reshape(3, 3))
    half = Rotation.from_rotvec(-0.5 * Rotation.from_matrix(Rinv).as_rotvec()).as_matrix()
    peer = half @ Rinv
    t = -peer @ np.asarray(T, np.float64).reshape(3)
    e1 = t / np.linalg.norm(t)
    e2 = np.array([-t[1], t[0], 0.0]) / np.hypot(t[0], t[1])
    rect = np.stack([e1, e2, np.cross(e1, e2)])
    f = (KL[0, 0] + KL[1, 1] + KR[0, 0] + KR[1, 1]) * 0.25
    newK = np.array([[f, 0, (KL[0, 2] + KR[0, 2]) * 0.5], [0, f, (KL[1, 2] + KR[1, 2]) * 0.5], [0, 0, 1.0]])
    Q = np.eye(4)
    Q[0, 3], Q[1, 3], Q[2, 2], Q[2, 3] = -newK[0, 2], -newK[1, 2], 0.0, f
    Q[3, 2], Q[3, 3] = -1.0 / float((rect @ t)[0]), 0.0
    return newK @ rect @ half, newK @ rect @ peer, Q


def rectify_maps(width, height, KL, distL, KR, distR, projL, projR, device=0, return_ms=False):
    """CStereoCameraRectification::doImageRectify on the GPU: returns (xmap_left, ymap_left,
    xmap_right, ymap_right), each (height, width) float64, INVALID_MAP = 9.99e4 outside."""
    lib = _load()
    h = C.c_void_p()
    rc = lib.svcal_create(int(device), C.byref(h))
    if rc != 0:
        raise SvicpError(rc, lib.svcal_last_error(None).decode())
    try:
        arr = [np.ascontiguousarray(a, np.float64).ravel() for a in (KL, distL, KR, distR, projL, projR)]
        k = len(arr[1])
        if len(arr[3]) != k:
            raise ValueError("both cameras need the same number of distortion terms")
        maps = [np.empty((int(height), int(width)), np.float64) for _ in range(4)]
        ms = C.c_double()
        rc = lib.svcal_rectify_maps(h, int(width), int(height), _d(arr[0]), _d(arr[1]), _d(arr[2]),
                                    _d(arr[3]), k, _d(arr[4]), _d(arr[5]), *[_d(m) for m in maps],
                                    C.byref(ms))
        if rc != 0:
            raise SvicpError(rc, lib.svcal_last_error(h).decode())
    finally:
        lib.svcal_destroy(h)
    return (*maps, ms.value) if return_ms else tuple(maps)
