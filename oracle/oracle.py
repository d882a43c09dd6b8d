"""ctypes front-end to oracle/liboracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product path (znippy_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("blake3_ref.c", "zstd_ref.c", "loops_ref.c")]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs)):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


class VerifyStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("total_chunks", "total_written_bytes", "verified_bytes",
                                          "corrupt_bytes", "corrupt_rows", "decode_errors")]


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.oracle_blake3.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.oracle_blake3.restype = None
        L.oracle_xxh64.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
        L.oracle_xxh64.restype = C.c_uint64
        L.oracle_zstd_get_decompressed_size.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]
        L.oracle_zstd_get_decompressed_size.restype = C.c_int
        L.oracle_zstd_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.oracle_zstd_decompress.restype = C.c_int64
        L.oracle_have_libzstd.restype = C.c_int
        L.oracle_libzstd_compress_bound.argtypes = [C.c_size_t]
        L.oracle_libzstd_compress_bound.restype = C.c_size_t
        L.oracle_libzstd_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
        L.oracle_libzstd_compress.restype = C.c_int64
        L.oracle_libzstd_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.oracle_libzstd_decompress.restype = C.c_int64
        L.oracle_decompress_rows.argtypes = [C.c_void_p] * 7 + [C.c_uint64, C.c_uint64, C.c_void_p, C.c_int,
                                                               C.c_int, C.POINTER(VerifyStats), C.c_void_p,
                                                               C.c_size_t]
        L.oracle_decompress_rows.restype = C.c_int
        L.oracle_compress_rounds.argtypes = [C.c_void_p] * 4 + [C.c_uint64, C.c_int, C.c_int, C.c_void_p,
                                                               C.c_size_t] + [C.c_void_p] * 4
        L.oracle_compress_rounds.restype = C.c_int64
        _lib = L
    return _lib


def _buf(b):
    """bytes / bytearray / numpy -> (keepalive, void*, nbytes)."""
    a = np.frombuffer(b, dtype=np.uint8) if not isinstance(b, np.ndarray) else b
    a = np.ascontiguousarray(a)
    return a, a.ctypes.data_as(C.c_void_p), a.nbytes


def blake3(data) -> bytes:
    a, p, n = _buf(data)
    out = (C.c_uint8 * 32)()
    lib().oracle_blake3(p, n, out)
    return bytes(out)


def xxh64(data, seed=0) -> int:
    a, p, n = _buf(data)
    return lib().oracle_xxh64(p, n, seed)


def zstd_decompressed_size(frame) -> int:
    a, p, n = _buf(frame)
    v = C.c_uint64()
    rc = lib().oracle_zstd_get_decompressed_size(p, n, C.byref(v))
    if rc:
        raise ValueError(f"oracle getDecompressedSize: error {rc}")
    return v.value


def zstd_decompress(frame, cap=None) -> bytes:
    a, p, n = _buf(frame)
    if cap is None:
        cap = zstd_decompressed_size(frame)
    out = np.empty(max(cap, 1), dtype=np.uint8)
    r = lib().oracle_zstd_decompress(out.ctypes.data_as(C.c_void_p), cap, p, n)
    if r < 0:
        raise ValueError(f"oracle decompress: error {r}")
    return out[:r].tobytes()


def have_libzstd() -> bool:
    return bool(lib().oracle_have_libzstd())


def libzstd_compress(data, level=19) -> bytes:
    a, p, n = _buf(data)
    cap = lib().oracle_libzstd_compress_bound(n)
    out = np.empty(cap, dtype=np.uint8)
    r = lib().oracle_libzstd_compress(out.ctypes.data_as(C.c_void_p), cap, p, n, level)
    if r < 0:
        raise RuntimeError(f"libzstd compress failed {r}")
    return out[:r].tobytes()


def libzstd_decompress(frame, cap) -> bytes:
    a, p, n = _buf(frame)
    out = np.empty(max(cap, 1), dtype=np.uint8)
    r = lib().oracle_libzstd_decompress(out.ctypes.data_as(C.c_void_p), cap, p, n)
    if r < 0:
        raise ValueError(f"libzstd decompress failed {r}")
    return out[:r].tobytes()


def decompress_rows(blobs, blob_offset, blob_size, uncompressed_size, out_offset, compressed_bitmap,
                    checksum, row_begin, row_end, out=None, n_threads=1, use_libzstd=False,
                    corrupt_cap=1024):
    """Restated read loop (decompress.rs:L113-192) over in-memory columns."""
    ks = []

    def ptr(x, dt):
        arr = np.ascontiguousarray(np.asarray(x, dtype=dt))
        ks.append(arr)
        return arr.ctypes.data_as(C.c_void_p)

    st = VerifyStats()
    corrupt = np.zeros(corrupt_cap, dtype=np.uint64)
    rc = lib().oracle_decompress_rows(
        ptr(blobs, np.uint8), ptr(blob_offset, np.uint64), ptr(blob_size, np.uint64),
        ptr(uncompressed_size, np.uint64), ptr(out_offset, np.uint64), ptr(compressed_bitmap, np.uint8),
        ptr(checksum, np.uint8), row_begin, row_end,
        out.ctypes.data_as(C.c_void_p) if out is not None else None, n_threads, int(use_libzstd),
        C.byref(st), corrupt.ctypes.data_as(C.c_void_p), corrupt_cap)
    if rc:
        raise RuntimeError(f"oracle_decompress_rows rc={rc}")
    stats = {n: getattr(st, n) for n, _ in VerifyStats._fields_}
    return stats, np.sort(corrupt[:min(st.corrupt_rows, corrupt_cap)])


def compress_rounds(src, off, length, skip, level=19, n_threads=1):
    """Restated write loop (stream_packer.rs:L215-284). Returns dict of per-round columns."""
    src = np.ascontiguousarray(np.asarray(src, dtype=np.uint8))
    off = np.ascontiguousarray(np.asarray(off, dtype=np.uint64))
    length = np.ascontiguousarray(np.asarray(length, dtype=np.uint64))
    skip = np.ascontiguousarray(np.asarray(skip, dtype=np.uint8))
    n = len(off)
    # ZSTD_compressBound(n) = n + n/256 + (n < 128 KiB ? (128 KiB - n)/2048 : 0), summed over rounds
    ln = length.astype(np.int64)
    cap = int(np.sum(ln + (ln >> 8) + np.where(ln < (128 << 10), ((128 << 10) - ln) >> 11, 0))) + 64
    blob = np.empty(max(cap, 1), dtype=np.uint8)
    bo = np.zeros(n, dtype=np.uint64)
    bs = np.zeros(n, dtype=np.uint64)
    ck = np.zeros((n, 32), dtype=np.uint8)
    cm = np.zeros(n, dtype=np.uint8)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    r = lib().oracle_compress_rounds(vp(src), vp(off), vp(length), vp(skip), n, level, n_threads,
                                     vp(blob), cap, vp(bo), vp(bs), vp(ck), vp(cm))
    if r < 0:
        raise RuntimeError(f"oracle_compress_rounds rc={r}")
    return dict(blobs=blob[:r], blob_offset=bo, blob_size=bs, checksum=ck, compressed=cm)
