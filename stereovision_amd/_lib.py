"""ctypes binding of libsvicp.so (the C ABI in include/svicp.h).

The library is the only compute path: loading fails loudly when the shared
object is missing, and every call raises SvicpError on a non-zero status.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SVICP_LIBRARY") or os.path.join(_HERE, "libsvicp.so")   # the override serves A/B builds

OK, ERR_ARG, ERR_HIP, ERR_GRID, ERR_SINGULAR, ERR_NO_DEVICE, ERR_COMM = range(7)


class SvicpError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"svicp status {status}: {message}")
        self.status = status


class Options(C.Structure):
    _fields_ = [
        ("dist_threshold", C.c_float),
        ("normals_k", C.c_int),
        ("use_graph", C.c_int),
        ("time_kernels", C.c_int),
        ("sort_source", C.c_int),
        ("coherent", C.c_int),
        ("centred_normals", C.c_int),
    ]


class Info(C.Structure):
    _fields_ = [
        ("iterations", C.c_int),
        ("status", C.c_int),
        ("inliers", C.c_double),
        ("weighted_sq_residual", C.c_double),
        ("step_norm", C.c_double),
        ("weight_sum", C.c_double),
    ]


# every symbol include/svicp.h declares: (restype, argtypes)
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)
_dp = C.POINTER(C.c_double)
_ctx = C.c_void_p
SYMBOLS = {
    "svicp_create": (C.c_int, [C.c_int, C.POINTER(_ctx)]),
    "svicp_destroy": (C.c_int, [_ctx]),
    "svicp_last_error": (C.c_char_p, [_ctx]),
    "svicp_default_options": (C.c_int, [C.POINTER(Options)]),
    "svicp_set_options": (C.c_int, [_ctx, C.POINTER(Options)]),
    "svicp_get_options": (C.c_int, [_ctx, C.POINTER(Options)]),
    "svicp_set_target": (C.c_int, [_ctx, _fp, C.c_int64, _fp]),
    "svicp_get_normals": (C.c_int, [_ctx, _fp]),
    "svicp_set_source": (C.c_int, [_ctx, _fp, C.c_int64]),
    "svicp_get_source": (C.c_int, [_ctx, _fp]),
    "svicp_host_register": (C.c_int, [C.c_void_p, C.c_int64]),
    "svicp_host_unregister": (C.c_int, [C.c_void_p]),
    "svicp_host_alloc": (C.c_int, [C.c_int64, C.POINTER(C.c_void_p)]),
    "svicp_host_free": (C.c_int, [C.c_void_p]),
    "svicp_run": (C.c_int, [_ctx, C.c_int, _dp, C.POINTER(Info)]),
    "svicp_run_enqueue": (C.c_int, [_ctx, C.c_int, C.c_void_p]),
    "svicp_last_iterations": (C.c_int, [_ctx, C.POINTER(C.c_int)]),
    "svicp_get_history": (C.c_int, [_ctx, C.c_int, _dp, _dp]),
    "svicp_correspondences": (C.c_int, [_ctx, _ip, _fp]),
    "svicp_knn": (C.c_int, [_ctx, _fp, C.c_int64, C.c_int, _ip, _fp]),
    "svicp_set_stream": (C.c_int, [_ctx, C.c_void_p]),
    "svicp_use_own_stream": (C.c_int, [_ctx]),
    "svicp_synchronize": (C.c_int, [_ctx]),
    "svicp_comm_unique_id": (C.c_int, [C.c_char_p]),
    "svicp_comm_init": (C.c_int, [_ctx, C.c_char_p, C.c_int, C.c_int]),
    "svicp_comm_destroy": (C.c_int, [_ctx]),
    "svicp_shard_run": (C.c_int, [_ctx, C.c_int, _dp, C.POINTER(Info)]),
    "svicp_shard_run_mode": (C.c_int, [_ctx, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "svicp_shard_set_total": (C.c_int, [_ctx, C.c_int64]),
    "svicp_shard_begin": (C.c_int, [_ctx]),
    "svicp_shard_accumulate": (C.c_int, [_ctx, C.c_void_p]),
    "svicp_shard_update": (C.c_int, [_ctx, C.c_void_p]),
    "svicp_shard_finish": (C.c_int, [_ctx, _dp, C.POINTER(Info)]),
    "svicp_kernel_times": (C.c_int, [_ctx, _dp, _dp, C.POINTER(C.c_int)]),
    "svicp_iteration_times": (C.c_int, [_ctx, _dp, _dp]),
    "svicp_time_nn": (C.c_int, [_ctx, C.c_int, C.c_int, _dp]),
    "svicp_time_gather": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int, _dp, _dp]),
    "svicp_sizes": (C.c_int, [_ctx, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "svicp_grid_info": (C.c_int, [_ctx, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "svicp_search_counts": (C.c_int, [_ctx, C.c_int, _dp]),
    "svicp_tiles_left_unread": (C.c_int, [_ctx, C.c_int, _dp]),
    "svicp_debug_use_tile_boxes": (C.c_int, [_ctx, C.c_int]),
    "svicp_run_point2point": (C.c_int, [_ctx, C.c_int, _dp, C.POINTER(Info)]),
    "svicp_get_history_rt": (C.c_int, [_ctx, C.c_int, _dp, _dp]),
    "svicp_batch_create": (C.c_int, [C.POINTER(_ctx), C.c_int, C.POINTER(C.c_void_p)]),
    "svicp_batch_create_from_arrays": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int64), _fp, _fp,
                                                 C.POINTER(C.c_int64), _fp, C.POINTER(Options),
                                                 C.POINTER(C.c_void_p)]),
    "svicp_batch_pair": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(_ctx)]),
    "svicp_batch_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), _dp, _dp]),
    "svicp_batch_destroy": (C.c_int, [C.c_void_p]),
    "svicp_batch_last_error": (C.c_char_p, [C.c_void_p]),
    "svicp_batch_run": (C.c_int, [C.c_void_p, C.c_int, _dp, C.POINTER(Info)]),
    "svicp_batch_last_ms": (C.c_int, [C.c_void_p, _dp]),
}

_LIB = None


def load():
    """Loads libsvicp.so and binds every declared symbol.  No GPU is touched."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C stereovision_amd/csrc` "
                "(hipcc, --offload-arch=gfx950). There is no CPU fallback."
            )
        # A process that also uses PyTorch must load PyTorch's bundled HIP runtime before
        # this library pulls in the system one: with the other order torch finds no GPU.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)   # AttributeError if the export is missing
            fn.restype = res
            fn.argtypes = args
        _LIB = lib
    return _LIB
