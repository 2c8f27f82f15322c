"""ctypes loader for the CPU oracle (oracle/liboracle.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The shipped package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_NORMALS_K = 20


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("icp_oracle.c", "cal_oracle.c", "p2p_oracle.c")]
    if force or not os.path.exists(so) or any(
        os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so) for src in srcs
    ):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        fp = C.POINTER(C.c_float)
        ip = C.POINTER(C.c_int32)
        dp = C.POINTER(C.c_double)
        L.orc_kdtree_build.restype = C.c_void_p
        L.orc_kdtree_build.argtypes = [fp, C.c_int64]
        L.orc_kdtree_free.argtypes = [C.c_void_p]
        L.orc_kdtree_knn.argtypes = [C.c_void_p, fp, C.c_int64, C.c_int, ip, fp, C.c_int]
        L.orc_knn_brute.argtypes = [fp, C.c_int64, fp, C.c_int64, C.c_int, ip, fp]
        L.orc_eig3_smallest.argtypes = [dp, dp, dp]
        L.orc_calc_normals.argtypes = [C.c_void_p, fp, C.c_int64, fp, ip, C.c_int]
        L.orc_icp_point2plane.restype = C.c_int
        L.orc_icp_point2plane.argtypes = [
            C.c_void_p, fp, fp, fp, C.c_int64, C.c_int, dp, dp, ip, fp, dp, C.c_int, C.c_int,
        ]
        L.orc_parse_txt.restype = C.c_int64
        L.orc_parse_txt.argtypes = [C.c_char_p, C.c_int64, fp, C.c_int64]
        L.orc_max_threads.restype = C.c_int
        _LIB = L
    return _LIB


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _cloud(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] == 3
    return a


class KDTree:
    """Exact kd-tree over an (M, 3) fp32 cloud; stands in for flann KDTREE_SINGLE."""

    def __init__(self, ref):
        self.ref = _cloud(ref)
        self._h = lib().orc_kdtree_build(_f(self.ref), self.ref.shape[0])

    def __del__(self):
        try:
            if self._h:
                lib().orc_kdtree_free(self._h)
                self._h = None
        except Exception:
            pass

    def knn(self, q, k=1, threads=1):
        q = _cloud(q)
        n = q.shape[0]
        idx = np.empty((n, k), np.int32)
        d2 = np.empty((n, k), np.float32)
        lib().orc_kdtree_knn(self._h, _f(q), n, k, _i(idx), _f(d2), threads)
        return idx, d2


def knn_brute(ref, q, k=1):
    ref = _cloud(ref)
    q = _cloud(q)
    n = q.shape[0]
    idx = np.empty((n, k), np.int32)
    d2 = np.empty((n, k), np.float32)
    lib().orc_knn_brute(_f(ref), ref.shape[0], _f(q), n, k, _i(idx), _f(d2))
    return idx, d2


def eig3_smallest(W):
    W = np.ascontiguousarray(W, dtype=np.float64).reshape(9)
    v = np.empty(3, np.float64)
    ev = C.c_double(0)
    lib().orc_eig3_smallest(_d(W), _d(v), C.byref(ev))
    return v, ev.value


def calc_normals(tree, threads=1, want_knn=False):
    m = tree.ref.shape[0]
    normals = np.empty((m, 3), np.float32)
    knn = np.empty((m, _NORMALS_K), np.int32) if want_knn else None
    lib().orc_calc_normals(
        tree._h, _f(tree.ref), m, _f(normals), _i(knn) if want_knn else None, threads
    )
    return (normals, knn) if want_knn else normals


def icp_point2plane(tree, normals, src, iters, faithful=True, threads=1, history=True):
    """Runs the restated CPoint2PlaneICP::run.  Returns a dict; src is copied, the
    updated copy is returned as 'src' (the reference updates the caller's buffer)."""
    src = _cloud(src).copy()
    normals = _cloud(normals)
    n = src.shape[0]
    trans = np.empty(16, np.float64)
    X = np.empty((iters, 6), np.float64)
    info = np.empty((iters, 4), np.float64)
    idx = np.empty((iters, n), np.int32) if history else None
    d2 = np.empty((iters, n), np.float32) if history else None
    rc = lib().orc_icp_point2plane(
        tree._h, _f(tree.ref), _f(normals), _f(src), n, iters, _d(trans), _d(X),
        _i(idx) if history else None, _f(d2) if history else None, _d(info),
        1 if faithful else 0, threads,
    )
    return dict(rc=rc, trans=trans.reshape(4, 4), X=X, info=info, idx=idx, d2=d2, src=src)


def set_constants(dist_threshold=5.0, normals_k=20):
    """The two numbers the reference hard-codes (CPoint2PlaneICP.cpp:68, :125)."""
    L = lib()
    L.orc_set_constants.argtypes = [C.c_float, C.c_int]
    L.orc_set_constants(float(dist_threshold), int(normals_k))
    global _NORMALS_K
    _NORMALS_K = int(normals_k)


def set_solve_perturbation(amplitude=0.0, seed=0):
    """Sensitivity studies: displaces every solve's X by +-amplitude per component (0: off)."""
    L = lib()
    L.orc_set_solve_perturbation.argtypes = [C.c_double, C.c_uint64]
    L.orc_set_solve_perturbation(float(amplitude), int(seed))


def parse_txt(text: bytes):
    n = lib().orc_parse_txt(text, len(text), None, 0)
    out = np.empty(n, np.float32)
    lib().orc_parse_txt(text, len(text), _f(out), n)
    return out[: (n // 3) * 3].reshape(-1, 3)


def max_threads():
    return lib().orc_max_threads()


# ---------------------------------------------------------------- calibration
def _cal_lib():
    L = lib()
    if not getattr(L, "_cal_bound", False):
        dp = C.POINTER(C.c_double)
        L.orc_rodrigues.argtypes = [dp, dp]
        L.orc_cal_value.argtypes = [dp, dp, C.c_int, dp, C.c_double, C.c_double, dp]
        L.orc_stereo_residual.argtypes = [dp, C.c_int, C.c_int, C.c_int, dp, dp, dp, dp]
        L.orc_stereo_jacobian_dense.argtypes = [dp, C.c_int, C.c_int, C.c_int, dp, dp]
        L.orc_stereo_jacobian_dense_fixed_intrinsics.argtypes = [dp, C.c_int, C.c_int, C.c_int, dp, dp]
        L.orc_stereo_eval_sparse.restype = C.c_double
        L.orc_stereo_eval_sparse.argtypes = [dp, C.c_int, C.c_int, C.c_int, dp, dp, dp] + [dp] * 11
        L.orc_mono_residual_jacobian_dense.argtypes = [dp, C.c_int, C.c_int, dp, dp, dp, dp]
        L._cal_bound = True
    return L


def _dd(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def rodrigues(r):
    r = _dd(r).reshape(3)
    R = np.empty(9)
    _cal_lib().orc_rodrigues(_d(r), _d(R))
    return R.reshape(3, 3)


def stereo_residual(vec, k, world, left, right):
    vec, world, left, right = _dd(vec), _dd(world), _dd(left), _dd(right)
    V, Cn = left.shape[0], left.shape[1]
    res = np.empty(4 * V * Cn)
    _cal_lib().orc_stereo_residual(_d(vec), k, V, Cn, _d(world), _d(left), _d(right), _d(res))
    return res


def stereo_jacobian_dense(vec, k, world, V, update_intrinsics=True):
    """The reference's dense Jacobian: every column (update_intrinsics, :95-186) or the
    6 V + 6 columns of the views and the pose only (:187-245)."""
    vec, world = _dd(vec), _dd(world)
    Cn = world.shape[0]
    if update_intrinsics:
        J = np.empty((4 * V * Cn, len(vec)))
        _cal_lib().orc_stereo_jacobian_dense(_d(vec), k, V, Cn, _d(world), _d(J))
    else:
        J = np.empty((4 * V * Cn, 6 * V + 6))
        _cal_lib().orc_stereo_jacobian_dense_fixed_intrinsics(_d(vec), k, V, Cn, _d(world), _d(J))
    return J


def stereo_eval_sparse(vec, k, world, left, right, want_residual=True):
    vec, world, left, right = _dd(vec), _dd(world), _dd(left), _dd(right)
    V, Cn = left.shape[0], left.shape[1]
    nI = 4 + k
    out = dict(LL=np.empty((nI, nI)), RR=np.empty((nI, nI)), RS=np.empty((nI, 6)),
               SS=np.empty((6, 6)), LE=np.empty((V, nI, 6)), EE=np.empty((V, 6, 6)),
               gL=np.empty(nI), gR=np.empty(nI), gE=np.empty((V, 6)), gS=np.empty(6))
    res = np.empty(4 * V * Cn) if want_residual else None
    ss = _cal_lib().orc_stereo_eval_sparse(
        _d(vec), k, V, Cn, _d(world), _d(left), _d(right),
        *[_d(out[n]) for n in ("LL", "RR", "RS", "SS", "LE", "EE", "gL", "gR", "gE", "gS")],
        _d(res) if want_residual else None)
    out["residual"] = res
    out["sq_residual"] = ss
    return out


def mono_residual_jacobian_dense(vec, world, img, want_jacobian=True):
    vec, world, img = _dd(vec), _dd(world), _dd(img)
    V, Cn = img.shape[0], img.shape[1]
    res = np.empty(2 * V * Cn)
    J = np.empty((2 * V * Cn, len(vec))) if want_jacobian else None
    _cal_lib().orc_mono_residual_jacobian_dense(_d(vec), V, Cn, _d(world), _d(img), _d(res),
                                                _d(J) if want_jacobian else None)
    return res, J


# ---------------------------------------------------------------- point-to-point ICP
def kabsch_rotation(W):
    L = lib()
    dp = C.POINTER(C.c_double)
    L.orc_kabsch_rotation.argtypes = [dp, dp]
    W = _dd(W).reshape(9)
    R = np.empty(9)
    L.orc_kabsch_rotation(_d(W), _d(R))
    return R.reshape(3, 3)


def svd3(W):
    L = lib()
    dp = C.POINTER(C.c_double)
    L.orc_svd3.argtypes = [dp, dp, dp, dp]
    W = _dd(W).reshape(9)
    U, S, V = np.empty(9), np.empty(3), np.empty(9)
    L.orc_svd3(_d(W), _d(U), _d(S), _d(V))
    return U.reshape(3, 3), S, V.reshape(3, 3)


def icp_point2point(tree, src, iters, faithful=True, threads=1, history=True):
    """Runs the restated CPoint2PointICP::run; src is copied, the updated copy is returned."""
    L = lib()
    fp, ip, dp = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_double)
    L.orc_icp_point2point.restype = C.c_int
    L.orc_icp_point2point.argtypes = [C.c_void_p, fp, fp, C.c_int64, C.c_int, dp, dp, dp, ip, fp,
                                      dp, C.c_int, C.c_int]
    src = _cloud(src).copy()
    n = src.shape[0]
    trans = np.empty(16, np.float64)
    RT = np.empty((iters, 12), np.float64)
    W = np.empty((iters, 9), np.float64)
    info = np.empty((iters, 4), np.float64)
    idx = np.empty((iters, n), np.int32) if history else None
    d2 = np.empty((iters, n), np.float32) if history else None
    rc = L.orc_icp_point2point(tree._h, _f(tree.ref), _f(src), n, iters, _d(trans), _d(RT), _d(W),
                               _i(idx) if history else None, _f(d2) if history else None,
                               _d(info), 1 if faithful else 0, threads)
    return dict(rc=rc, trans=trans.reshape(4, 4), RT=RT, W=W.reshape(iters, 3, 3), info=info,
                idx=idx, d2=d2, src=src)


# ---------------------------------------------------------------- rectification
def stereo_rectify(KL, KR, R, T):
    L = lib()
    dp = C.POINTER(C.c_double)
    L.orc_stereo_rectify.restype = C.c_int
    L.orc_stereo_rectify.argtypes = [dp] * 7
    a = [_dd(x).ravel() for x in (KL, KR, R, T)]
    pl, pr, Q = np.empty(9), np.empty(9), np.empty(16)
    ok = L.orc_stereo_rectify(_d(a[0]), _d(a[1]), _d(a[2]), _d(a[3]), _d(pl), _d(pr), _d(Q))
    return ok, pl.reshape(3, 3), pr.reshape(3, 3), Q.reshape(4, 4)


def rectify_maps(w, h, KL, dL, KR, dR, projL, projR):
    L = lib()
    dp = C.POINTER(C.c_double)
    L.orc_rectify_maps.restype = C.c_int
    L.orc_rectify_maps.argtypes = [C.c_int, C.c_int, dp, dp, dp, dp, C.c_int, dp, dp, dp, dp, dp, dp]
    a = [_dd(x).ravel() for x in (KL, dL, KR, dR, projL, projR)]
    maps = [np.empty((h, w)) for _ in range(4)]
    ok = L.orc_rectify_maps(w, h, _d(a[0]), _d(a[1]), _d(a[2]), _d(a[3]), len(a[1]), _d(a[4]),
                            _d(a[5]), *[_d(m) for m in maps])
    return ok, maps
