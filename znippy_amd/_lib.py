"""ctypes binding of libznippy_hip.so (the C ABI in include/znippy_hip.h).

There is NO CPU fallback: if the HIP library is missing or fails to load, importing the
product path raises.  Device memory / streams come from PyTorch-ROCm (plumbing only).
"""
import ctypes as C
import os

import numpy as np

from . import _build

_HERE = os.path.dirname(os.path.abspath(__file__))

OK = 0
E_INVAL, E_HIP, E_NOMEM, E_DST_SMALL, E_CORRUPT, E_UNSUPPORTED, E_CHECKSUM = -1, -2, -3, -4, -5, -6, -7
_ERRNAMES = {E_INVAL: "invalid argument", E_HIP: "HIP runtime error", E_NOMEM: "out of device memory",
             E_DST_SMALL: "destination too small", E_CORRUPT: "corrupt frame",
             E_UNSUPPORTED: "unsupported frame", E_CHECKSUM: "content checksum mismatch"}


class ZnippyError(RuntimeError):
    def __init__(self, code, what, detail=""):
        self.code = code
        super().__init__(f"{what}: {_ERRNAMES.get(code, code)} ({code}) {detail}".strip())


class VerifyCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("total_chunks", "total_written_bytes", "verified_bytes",
                                          "corrupt_bytes", "corrupt_rows", "decode_errors")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


_lib = None
vp = C.c_void_p

EXPORTS = [
    "znippy_ctx_create", "znippy_ctx_destroy", "znippy_last_error", "znippy_ctx_sync",
    "znippy_compress_bound", "znippy_get_decompressed_size", "znippy_decompress", "znippy_compress",
    "znippy_blake3", "znippy_rows_create", "znippy_rows_destroy", "znippy_decode_verify_rows",
    "znippy_decode_verify_rows_async", "znippy_rows_results", "znippy_rows_digests",
    "znippy_rounds_create", "znippy_rounds_destroy", "znippy_rounds_blob_bound",
    "znippy_encode_hash_rounds", "znippy_encode_hash_rounds_async", "znippy_rounds_results",
    "znippy_rounds_results_view", "znippy_rows_results_lagged", "znippy_rows_set_blob_cap", "znippy_rounds_results_lagged", "znippy_rounds_set_store_incompressible", "znippy_hash_rounds", "znippy_last_kernel_times", "znippy_measure_blake3_pass_ns", "znippy_last_shader_ghz", "znippy_ctx_set_kernel_timing", "znippy_rows_foreign_stats", "znippy_ctx_set_level", "znippy_ctx_level",
]


def lib_path():
    return _build.SO


def lib():
    """Load (never build-on-import on a GPU box: the .so travels with the snapshot)."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm ships its own libamdhip64; load it FIRST so this library's hip* symbols bind to the
    # same runtime instance (two HIP runtimes in one process: the second one finds no device).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback for the product path.")
    L = C.CDLL(path)
    L.znippy_ctx_create.argtypes = [C.c_int, vp, C.POINTER(vp)]
    L.znippy_ctx_destroy.argtypes = [vp]
    L.znippy_ctx_destroy.restype = None
    L.znippy_last_error.argtypes = [vp]
    L.znippy_last_error.restype = C.c_char_p
    L.znippy_ctx_sync.argtypes = [vp]
    L.znippy_compress_bound.argtypes = [C.c_size_t]
    L.znippy_compress_bound.restype = C.c_size_t
    L.znippy_get_decompressed_size.argtypes = [vp, C.c_size_t, C.POINTER(C.c_uint64)]
    L.znippy_decompress.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.znippy_compress.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.znippy_blake3.argtypes = [vp, vp, C.c_size_t, vp]
    L.znippy_rows_create.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_uint64, C.c_uint64, C.POINTER(vp)]
    L.znippy_rows_destroy.argtypes = [vp]
    L.znippy_rows_destroy.restype = None
    L.znippy_decode_verify_rows.argtypes = [vp, vp, vp, C.c_uint64, vp, C.c_uint64, C.POINTER(VerifyCounters),
                                            vp, C.c_uint64, vp]
    L.znippy_decode_verify_rows_async.argtypes = [vp, vp, vp, C.c_uint64, vp, C.c_uint64]
    L.znippy_rows_results.argtypes = [vp, vp, C.POINTER(VerifyCounters), vp, C.c_uint64, vp]
    L.znippy_rows_results_lagged.argtypes = [vp, vp, C.c_uint, C.POINTER(VerifyCounters)]
    L.znippy_rounds_results_lagged.argtypes = [vp, vp, C.c_uint, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp),
                                               C.POINTER(C.c_uint64)]
    L.znippy_rows_set_blob_cap.argtypes = [vp, C.c_uint64]
    L.znippy_rows_digests.argtypes = [vp, vp, vp]
    L.znippy_rounds_create.argtypes = [vp, vp, vp, vp, C.c_uint64, C.POINTER(vp)]
    L.znippy_rounds_destroy.argtypes = [vp]
    L.znippy_rounds_destroy.restype = None
    L.znippy_rounds_blob_bound.argtypes = [vp]
    L.znippy_rounds_blob_bound.restype = C.c_uint64
    L.znippy_encode_hash_rounds.argtypes = [vp, vp, vp, vp, C.c_uint64, vp, vp, vp, vp, C.POINTER(C.c_uint64)]
    L.znippy_encode_hash_rounds_async.argtypes = [vp, vp, vp, vp, C.c_uint64]
    L.znippy_rounds_results.argtypes = [vp, vp, vp, vp, vp, vp, C.POINTER(C.c_uint64)]
    L.znippy_rounds_results_view.argtypes = [vp, vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_uint64)]
    L.znippy_rounds_set_store_incompressible.argtypes = [vp, C.c_int]
    L.znippy_hash_rounds.argtypes = [vp, vp, vp, vp]
    L.znippy_last_kernel_times.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.c_int]
    L.znippy_measure_blake3_pass_ns.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.znippy_last_shader_ghz.argtypes = [vp, C.POINTER(C.c_float)]
    L.znippy_ctx_set_kernel_timing.argtypes = [vp, C.c_int]
    L.znippy_ctx_set_level.argtypes = [vp, C.c_int]
    L.znippy_ctx_level.argtypes = [vp]
    L.znippy_rows_foreign_stats.argtypes = [vp, vp, C.POINTER(C.c_uint64)]
    _lib = L
    return L


def np_ptr(a):
    return a.ctypes.data_as(vp)


def as_np(x, dtype):
    return np.ascontiguousarray(np.asarray(x, dtype=dtype))
