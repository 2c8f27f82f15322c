"""Host-side mirror of the reference's registration interface over the C ABI.

`Context` is a thin object wrapper of one svicp_ctx.  `CPoint2PlaneICP`,
`CKNNSearch` and `CTxtReader` keep the reference's method names, argument
meaning and side effects (PointCloud/Registration/CPoint2PlaneICP.h:11-17,
CKNNSearch.h:12-17, CTxtReader.h:6-13); a cloud is an (N, 3) float32 numpy
array where the reference passes a flann::Matrix<float> view.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Info, Options, SvicpError


def _cloud(a, name):
    a = np.asarray(a)
    if a.ndim != 2 or a.shape[1] != 3:
        raise ValueError(f"{name}: expected an (N, 3) array, got shape {a.shape}")
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def host_register(array):
    """Pins a numpy array in place (svicp_host_register); pair with host_unregister."""
    lib = _lib.load()
    rc = lib.svicp_host_register(C.c_void_p(array.ctypes.data), array.nbytes)
    if rc != 0:
        raise SvicpError(rc, lib.svicp_last_error(None).decode())


def host_unregister(array):
    lib = _lib.load()
    rc = lib.svicp_host_unregister(C.c_void_p(array.ctypes.data))
    if rc != 0:
        raise SvicpError(rc, lib.svicp_last_error(None).decode())


class _PinnedBlock:
    """Owner of one svicp_host_alloc block; numpy views keep it alive through .base."""

    def __init__(self, nbytes):
        self._lib = _lib.load()
        p = C.c_void_p()
        rc = self._lib.svicp_host_alloc(int(nbytes), C.byref(p))
        if rc != 0:
            raise SvicpError(rc, self._lib.svicp_last_error(None).decode())
        self.ptr, self.nbytes = p.value, int(nbytes)
        self.buffer = (C.c_char * self.nbytes).from_address(self.ptr)

    def __del__(self):
        try:
            self._lib.svicp_host_free(C.c_void_p(self.ptr))
        except Exception:
            pass


def pinned_empty(shape, dtype=np.float32):
    """A numpy array in pinned host memory (svicp_host_alloc): clouds kept in such arrays go
    to the device and back at the speed of the link."""
    shape = tuple(int(x) for x in np.atleast_1d(shape))
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    block = _PinnedBlock(max(n, 1))
    arr = np.frombuffer(block.buffer, dtype=dtype, count=int(np.prod(shape))).reshape(shape).view(_PinnedArray)
    arr._block = block          # views of `arr` keep `arr` (their base) and with it the block alive
    return arr


class _PinnedArray(np.ndarray):
    """ndarray over a _PinnedBlock; the only addition is the reference that keeps the block."""
    _block = None


class Context:
    """One svicp_ctx: a target grid, a resident source and the iteration state."""

    def __init__(self, device=0, **options):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        rc = self._lib.svicp_create(int(device), C.byref(self._h))
        if rc != 0:
            raise SvicpError(rc, self._lib.svicp_last_error(None).decode())
        if options:
            self.set_options(**options)

    def _check(self, rc):
        if rc != 0:
            raise SvicpError(rc, self._lib.svicp_last_error(self._h).decode())

    @classmethod
    def _borrowed(cls, lib, handle):
        """A view of a context that something else owns (a pair of an ArenaBatch)."""
        self = cls.__new__(cls)
        self._lib = lib
        self._h = handle
        self._is_borrowed = True
        return self

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            if not getattr(self, "_is_borrowed", False):
                self._lib.svicp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- options --
    def options(self):
        o = Options()
        self._check(self._lib.svicp_get_options(self._h, C.byref(o)))
        return o

    def set_options(self, **kw):
        o = self.options()
        for k, v in kw.items():
            if not hasattr(o, k):
                raise TypeError(f"unknown option {k!r}")
            setattr(o, k, v)
        self._check(self._lib.svicp_set_options(self._h, C.byref(o)))

    # -- clouds --
    def set_target(self, xyz, normals=None):
        xyz = _cloud(xyz, "target")
        if normals is not None:
            normals = _cloud(normals, "normals")
            if normals.shape != xyz.shape:
                raise ValueError("normals must match the target's shape")
        self._check(self._lib.svicp_set_target(
            self._h, _fp(xyz), xyz.shape[0], _fp(normals) if normals is not None else None))

    def get_normals(self):
        m, _ = self.sizes()
        out = np.empty((m, 3), np.float32)
        self._check(self._lib.svicp_get_normals(self._h, _fp(out)))
        return out

    def set_source(self, xyz):
        xyz = _cloud(xyz, "source")
        self._check(self._lib.svicp_set_source(self._h, _fp(xyz), xyz.shape[0]))

    def get_source(self, out=None):
        _, n = self.sizes()
        if out is None:
            out = np.empty((n, 3), np.float32)
        assert out.dtype == np.float32 and out.shape == (n, 3) and out.flags.c_contiguous
        self._check(self._lib.svicp_get_source(self._h, _fp(out)))
        return out

    def sizes(self):
        m, n = C.c_int64(), C.c_int64()
        self._check(self._lib.svicp_sizes(self._h, C.byref(m), C.byref(n)))
        return m.value, n.value

    def grid_info(self):
        """(ranked blocks, bytes of the probe records, records kept in full) of the target's
        search tables."""
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        self._check(self._lib.svicp_grid_info(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    # -- registration --
    def run(self, iterations):
        """Returns (4x4 transform, Info)."""
        T = np.empty(16, np.float64)
        info = Info()
        self._check(self._lib.svicp_run(self._h, int(iterations), _dp(T), C.byref(info)))
        return T.reshape(4, 4), info

    def last_iterations(self):
        """Rows the history calls return: iterations of the last successful run."""
        k = C.c_int()
        self._check(self._lib.svicp_last_iterations(self._h, C.byref(k)))
        return k.value

    def run_enqueue(self, iterations, dev_out20):
        """svicp_run_enqueue: queues the loop on the context's stream and returns at once.
        dev_out20: address of 20 doubles of device memory (transform, status, statistics)."""
        self._check(self._lib.svicp_run_enqueue(self._h, int(iterations), C.c_void_p(int(dev_out20))))

    def run_point2point(self, iterations):
        """CPoint2PointICP::run on this context.  Returns (4x4 transform, Info)."""
        T = np.empty(16, np.float64)
        info = Info()
        self._check(self._lib.svicp_run_point2point(self._h, int(iterations), _dp(T), C.byref(info)))
        return T.reshape(4, 4), info

    def history_rt(self):
        """Per-iteration {R, T} (iters x 12) and {count, sum w d2, step, sum w} (iters x 4)
        of the last point-to-point run."""
        iters = self.last_iterations()
        RT = np.empty((iters, 12), np.float64)
        I = np.empty((iters, 4), np.float64)
        self._check(self._lib.svicp_get_history_rt(self._h, iters, _dp(RT), _dp(I)))
        return RT, I

    def history(self):
        """Per-iteration X (iters x 6) and {inliers, sum w b^2, |X|, sum w} (iters x 4)."""
        iters = self.last_iterations()
        X = np.empty((iters, 6), np.float64)
        I = np.empty((iters, 4), np.float64)
        self._check(self._lib.svicp_get_history(self._h, iters, _dp(X), _dp(I)))
        return X, I

    def search_counts(self):
        """Points searched per iteration of the last run()."""
        iters = self.last_iterations()
        out = np.zeros(iters, np.float64)
        self._check(self._lib.svicp_search_counts(self._h, iters, _dp(out)))
        return out

    def tiles_left_unread(self):
        """Tiles left unread per iteration of the last run() (large sources only)."""
        iters = self.last_iterations()
        out = np.zeros(iters, np.float64)
        self._check(self._lib.svicp_tiles_left_unread(self._h, iters, _dp(out)))
        return out

    def correspondences(self):
        _, n = self.sizes()
        idx = np.empty(n, np.int32)
        d2 = np.empty(n, np.float32)
        self._check(self._lib.svicp_correspondences(self._h, _ip(idx), _fp(d2)))
        return idx, d2

    def knn(self, query, k=1):
        q = _cloud(query, "query")
        k = min(int(k), 128)
        idx = np.empty((q.shape[0], k), np.int32)
        d2 = np.empty((q.shape[0], k), np.float32)
        self._check(self._lib.svicp_knn(self._h, _fp(q), q.shape[0], k, _ip(idx), _fp(d2)))
        return idx, d2

    # -- sharded stepping --
    def set_stream(self, stream_handle):
        """Launch on this hipStream_t from now on (0 is the default stream)."""
        self._check(self._lib.svicp_set_stream(self._h, C.c_void_p(int(stream_handle))))

    def use_own_stream(self):
        self._check(self._lib.svicp_use_own_stream(self._h))

    def sync(self):
        self._check(self._lib.svicp_synchronize(self._h))

    # -- sharded registration inside the library (RCCL) --
    def comm_unique_id(self):
        """128 bytes that identify a new communicator (rank 0 makes them, every rank gets them)."""
        buf = C.create_string_buffer(128)
        rc = self._lib.svicp_comm_unique_id(buf)
        if rc != 0:
            raise SvicpError(rc, self._lib.svicp_last_error(None).decode())
        return buf.raw

    def comm_init(self, unique_id, rank, world_size):
        assert len(unique_id) == 128
        self._check(self._lib.svicp_comm_init(self._h, C.c_char_p(bytes(unique_id)), int(rank), int(world_size)))

    def comm_destroy(self):
        self._check(self._lib.svicp_comm_destroy(self._h))

    def shard_run(self, iterations):
        """svicp_shard_run: returns (4x4 transform, Info), the same on every rank."""
        T = np.empty(16, np.float64)
        info = Info()
        self._check(self._lib.svicp_shard_run(self._h, int(iterations), _dp(T), C.byref(info)))
        return T.reshape(4, 4), info

    def shard_run_mode(self):
        """(captured in a graph?, world size) of the last shard_run."""
        a, b = C.c_int(), C.c_int()
        self._check(self._lib.svicp_shard_run_mode(self._h, C.byref(a), C.byref(b)))
        return bool(a.value), b.value

    def shard_set_total(self, total_points):
        self._check(self._lib.svicp_shard_set_total(self._h, int(total_points)))

    def shard_begin(self):
        self._check(self._lib.svicp_shard_begin(self._h))

    def shard_accumulate(self, dev_ptr):
        self._check(self._lib.svicp_shard_accumulate(self._h, C.c_void_p(dev_ptr)))

    def shard_update(self, dev_ptr):
        self._check(self._lib.svicp_shard_update(self._h, C.c_void_p(dev_ptr)))

    def shard_finish(self):
        T = np.empty(16, np.float64)
        info = Info()
        self._check(self._lib.svicp_shard_finish(self._h, _dp(T), C.byref(info)))
        return T.reshape(4, 4), info

    # -- measurement --
    def kernel_times(self):
        a, b, c = C.c_double(), C.c_double(), C.c_int()
        self._check(self._lib.svicp_kernel_times(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def iteration_times(self):
        """Per-iteration (fused_ms, solve_ms) arrays of the last timed run."""
        n = self.kernel_times()[2]
        a = np.zeros(n, np.float64)
        b = np.zeros(n, np.float64)
        self._check(self._lib.svicp_iteration_times(self._h, _dp(a), _dp(b)))
        return a, b

    def time_nn(self, exact=False, reps=10):
        ms = C.c_double()
        self._check(self._lib.svicp_time_nn(self._h, 1 if exact else 0, int(reps), C.byref(ms)))
        return ms.value


    def time_gather(self, real_addresses=True, per_query=16, reps=10):
        """(average ms, wave-level vector-memory instructions per run) of the gather probe."""
        ms, ins = C.c_double(), C.c_double()
        # an integer 12, 8 or 4: the search's addresses with records of that many bytes
        mode = real_addresses if type(real_addresses) is int else (1 if real_addresses else 0)
        self._check(self._lib.svicp_time_gather(self._h, mode, int(per_query),
                                                int(reps), C.byref(ms), C.byref(ins)))
        return ms.value, ins.value


class Batch:
    """Many independent pairs advanced together, one kernel launch per iteration
    (svicp_batch_* of include/svicp.h).  `contexts` already hold target and source."""

    def __init__(self, contexts):
        self._lib = _lib.load()
        self.contexts = list(contexts)
        arr = (C.c_void_p * len(self.contexts))(*[c._h for c in self.contexts])
        self._h = C.c_void_p()
        rc = self._lib.svicp_batch_create(arr, len(self.contexts), C.byref(self._h))
        if rc != 0:
            msg = self.contexts[0]._lib.svicp_last_error(self.contexts[0]._h) if self.contexts else b""
            raise SvicpError(rc, (msg or b"").decode())

    def run(self, iterations):
        """Returns (count x 4 x 4 transforms, list of Info)."""
        n = len(self.contexts)
        T = np.empty((n, 16), np.float64)
        infos = (Info * n)()
        rc = self._lib.svicp_batch_run(self._h, int(iterations), _dp(T), infos)
        if rc != 0:
            raise SvicpError(rc, self._lib.svicp_batch_last_error(self._h).decode())
        return T.reshape(n, 4, 4), list(infos)

    def last_ms(self):
        ms = C.c_double()
        self._lib.svicp_batch_last_ms(self._h, C.byref(ms))
        return ms.value

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.svicp_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ArenaBatch(Batch):
    """svicp_batch_create_from_arrays: hundreds of pairs from flat arrays, one stream and one
    arena of device memory for all of them.  targets / normals / sources: lists of (n_i, 3)
    float32 arrays (normals may be None: estimated per pair)."""

    def __init__(self, targets, normals, sources, device=0, **options):
        self._lib = _lib.load()
        self.contexts = []
        count = len(targets)
        assert count >= 1 and len(sources) == count and (normals is None or len(normals) == count)
        t_off = np.zeros(count + 1, np.int64)
        s_off = np.zeros(count + 1, np.int64)
        t_off[1:] = np.cumsum([len(t) for t in targets])
        s_off[1:] = np.cumsum([len(s) for s in sources])
        tgt = np.ascontiguousarray(np.concatenate([_cloud(t, "target") for t in targets]))
        src = np.ascontiguousarray(np.concatenate([_cloud(s, "source") for s in sources]))
        nrm = None if normals is None else np.ascontiguousarray(
            np.concatenate([_cloud(n, "normals") for n in normals]))
        assert nrm is None or nrm.shape == tgt.shape
        o = Options()
        self._lib.svicp_default_options(C.byref(o))
        for k, val in options.items():
            if not hasattr(o, k):
                raise TypeError(f"unknown option {k!r}")
            setattr(o, k, val)
        i64 = C.POINTER(C.c_int64)
        self._h = C.c_void_p()
        rc = self._lib.svicp_batch_create_from_arrays(
            int(device), count, t_off.ctypes.data_as(i64), _fp(tgt), _fp(nrm) if nrm is not None else None,
            s_off.ctypes.data_as(i64), _fp(src), C.byref(o), C.byref(self._h))
        if rc != 0:
            raise SvicpError(rc, self._lib.svicp_last_error(None).decode())
        self.count = count

    def pair(self, index):
        """Pair `index`'s context, on loan: get_source, history, correspondences."""
        h = C.c_void_p()
        if self._lib.svicp_batch_pair(self._h, int(index), C.byref(h)) != 0:
            raise IndexError(index)
        return Context._borrowed(self._lib, h)

    def info(self):
        """(pairs, set-up ms on the host, bytes of device memory in the arena)."""
        n, ms, by = C.c_int(), C.c_double(), C.c_double()
        self._lib.svicp_batch_info(self._h, C.byref(n), C.byref(ms), C.byref(by))
        return n.value, ms.value, by.value

    def run(self, iterations):
        T = np.empty((self.count, 16), np.float64)
        infos = (Info * self.count)()
        rc = self._lib.svicp_batch_run(self._h, int(iterations), _dp(T), infos)
        if rc != 0:
            raise SvicpError(rc, self._lib.svicp_batch_last_error(self._h).decode())
        return T.reshape(self.count, 4, 4), list(infos)


class CKNNSearch:
    """CKNNSearch (CKNNSearch.h:9-25): exact k-NN, squared fp32 distances."""

    def __init__(self, device=0):
        self._ctx = Context(device)
        self.m_k = 1                       # CKNNSearch.cpp:5
        self._query = None

    def setTreeData(self, data):
        self._ctx.set_target(data, normals=np.zeros_like(_cloud(data, "data")))

    def setQueryData(self, querydata):
        self._query = _cloud(querydata, "querydata")

    def setKCount(self, count):
        self.m_k = 128 if count > 128 else int(count)   # CKNNSearch.cpp:60-65

    def getKNeighbour(self):
        """Returns (indices N x k int32, squared distances N x k float32)."""
        if self._query is None:
            raise SvicpError(_lib.ERR_ARG, "getKNeighbour: setQueryData was not called")
        return self._ctx.knn(self._query, self.m_k)


class CPoint2PlaneICP:
    """CPoint2PlaneICP (CPoint2PlaneICP.h:8-27).

    run() transforms the caller's move array in place (count - 1 times), as the
    reference does to the buffer behind its flann::Matrix view, and prints the
    accumulated transform.  getTransform()/getInfo() are additions: the
    reference only prints.
    """

    def __init__(self, device=0, verbose=True):
        self._ctx = Context(device)
        self.m_count = 1
        self.m_eps = 0.0
        self._move = None
        self._T = None
        self._info = None
        self._verbose = verbose

    def setReferenceData(self, ref, normals=None):
        self._ctx.set_target(ref, normals)

    def setMoveData(self, move):
        if not (isinstance(move, np.ndarray) and move.dtype == np.float32
                and move.ndim == 2 and move.shape[1] == 3 and move.flags.c_contiguous):
            raise ValueError("setMoveData: pass a C-contiguous (N, 3) float32 array; "
                             "run() updates it in place")
        self._unpin()
        self._move = move
        if move.nbytes >= self.PIN_FROM_BYTES:
            try:
                host_register(move)
                self._pinned = move
            except SvicpError:
                self._pinned = None            # (not pinnable: the copies take the pageable path)

    PIN_FROM_BYTES = 1 << 20    # run() copies the buffer in and out on every call: pin what is large

    def _unpin(self):
        if getattr(self, "_pinned", None) is not None:
            try:
                host_unregister(self._pinned)
            except SvicpError:
                pass
            self._pinned = None

    def __del__(self):
        # (at interpreter shutdown the library loader itself may be gone: nothing to undo then.
        # The array stays alive as long as this object refers to it, so the registration never
        # outlives the memory.)
        try:
            self._unpin()
        except Exception:       # noqa: BLE001
            pass

    def setEpsilon(self, eps):
        self.m_eps = float(eps)            # stored and never read, as in the reference

    def setMaxIterateCount(self, count):
        self.m_count = int(count)

    def run(self):
        if self._move is None:
            raise SvicpError(_lib.ERR_ARG, "run: setMoveData was not called")
        self._ctx.set_source(self._move)
        self._T, self._info = self._ctx.run(self.m_count)
        self._ctx.get_source(self._move)
        if self._verbose:
            print("Current Transformation:")
            print(self._T)

    def getTransform(self):
        return self._T

    def getInfo(self):
        return self._info

    def getNormals(self):
        return self._ctx.get_normals()


class CPoint2PointICP(CPoint2PlaneICP):
    """CPoint2PointICP (CPoint2PointICP.h:8-27): same surface as CPoint2PlaneICP; the
    threshold of its weight function is 15 (CPoint2PointICP.cpp:90) and no normals are used."""

    def __init__(self, device=0, verbose=True):
        super().__init__(device, verbose)
        self._ctx.set_options(dist_threshold=15.0)

    def setReferenceData(self, ref, normals=None):
        ref = _cloud(ref, "ref")
        self._ctx.set_target(ref, ref)      # the second argument only suppresses normal estimation

    def run(self):
        if self._move is None:
            raise SvicpError(_lib.ERR_ARG, "run: setMoveData was not called")
        self._ctx.set_source(self._move)
        self._T, self._info = self._ctx.run_point2point(self.m_count)
        self._ctx.get_source(self._move)
        if self._verbose:
            print("Current Transformation:")
            print(self._T)


class CTxtReader:
    """CTxtReader (CTxtReader.cpp:13-61): space-separated floats, three per row."""

    def read(self, path):
        vals = []
        try:
            f = open(path, "r")
        except OSError:
            return np.empty((0, 3), np.float32)   # the reference returns an empty view
        with f:
            for line in f:
                for tok in line.rstrip("\n").split(" "):
                    if tok:
                        vals.append(_atof(tok))
        rows = len(vals) // 3
        return np.asarray(vals[: rows * 3], dtype=np.float64).astype(np.float32).reshape(rows, 3)


    # additions (no counterpart in the reference): raw little-endian float32 x y z triples
    def readBinary(self, path):
        try:
            raw = np.fromfile(path, dtype="<f4")
        except OSError:
            return np.empty((0, 3), np.float32)
        return np.ascontiguousarray(raw[: (raw.size // 3) * 3].reshape(-1, 3))

    @staticmethod
    def writeBinary(path, cloud):
        np.ascontiguousarray(cloud, dtype="<f4").tofile(path)


def _atof(tok):
    """C atof: the longest numeric prefix, 0.0 when there is none."""
    try:
        return float(tok)
    except ValueError:
        import re
        m = re.match(r"\s*[-+]?(\d+\.?\d*([eE][-+]?\d+)?|\.\d+([eE][-+]?\d+)?|inf(inity)?|nan)", tok, re.I)
        return float(m.group(0)) if m else 0.0
