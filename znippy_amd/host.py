"""ctypes binding of the compiled host layer (include/znippy_host.h, csrc/host/*.cpp)."""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from . import index as ix

vp = C.c_void_p


class _CR(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("total_files", "compressed_files", "uncompressed_files", "total_dirs",
                                          "total_bytes_in", "total_bytes_out", "compressed_bytes", "uncompressed_bytes",
                                          "chunks")] + [("compression_ratio", C.c_float)]


class _VR(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("total_files", "verified_files", "corrupt_files", "total_bytes",
                                          "verified_bytes", "corrupt_bytes", "chunks")]


class _ME(C.Structure):
    _fields_ = [("pkg_type", C.c_int8), ("repo", C.c_char_p), ("module_name", C.c_char_p), ("index_offset", C.c_uint64),
                ("index_len", C.c_uint64), ("row_count", C.c_uint64)]


HOST_EXPORTS = [
    "znippy_host_last_error", "znippy_compress_stream", "znippy_stream_send", "znippy_stream_send_packed", "znippy_stream_finish",
    "znippy_compress_dir", "znippy_decompress_archive", "znippy_archive_open", "znippy_archive_file_count", "znippy_archive_file_size",
    "znippy_archive_extract_file", "znippy_archive_extract_file_verified", "znippy_archive_close", "znippy_index_open", "znippy_index_rows",
    "znippy_index_manifest_len", "znippy_index_manifest_entry", "znippy_index_row", "znippy_index_metadata",
    "znippy_index_close", "znippy_interpret_footer", "znippy_write_manifest_bytes",
]

_L = None


def lib():
    global _L
    if _L is None:
        L = _lib.lib()
        L.znippy_host_last_error.restype = C.c_char_p
        L.znippy_compress_stream.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(vp)]
        L.znippy_stream_send.argtypes = [vp, C.c_char_p, vp, C.c_size_t, C.c_int, C.c_char_p]
        L.znippy_stream_send_packed.argtypes = [vp, C.c_uint64, vp, vp, vp, vp, vp, C.c_char_p]
        L.znippy_stream_finish.argtypes = [vp, C.POINTER(_CR)]
        L.znippy_compress_dir.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(_CR)]
        L.znippy_decompress_archive.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_uint32, C.c_uint32,
                                                C.POINTER(_VR), vp, C.c_uint64, C.POINTER(C.c_uint64)]
        L.znippy_archive_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
        L.znippy_archive_file_count.argtypes = [vp]
        L.znippy_archive_file_count.restype = C.c_uint64
        L.znippy_archive_file_size.argtypes = [vp, C.c_char_p]
        L.znippy_archive_file_size.restype = C.c_int64
        L.znippy_archive_extract_file.argtypes = [vp, C.c_char_p, vp, C.c_size_t, C.POINTER(C.c_size_t)]
        L.znippy_archive_extract_file_verified.argtypes = [vp, C.c_char_p, vp, C.c_size_t, C.POINTER(C.c_size_t)]
        L.znippy_archive_close.argtypes = [vp]
        L.znippy_archive_close.restype = None
        L.znippy_index_open.argtypes = [C.c_char_p, C.POINTER(vp)]
        L.znippy_index_rows.argtypes = [vp]
        L.znippy_index_rows.restype = C.c_uint64
        L.znippy_index_manifest_len.argtypes = [vp]
        L.znippy_index_manifest_len.restype = C.c_uint64
        L.znippy_index_manifest_entry.argtypes = [vp, C.c_uint64, C.POINTER(_ME)]
        L.znippy_index_row.argtypes = [vp, C.c_uint64, C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64),
                                       C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                       C.POINTER(C.c_uint64), C.POINTER(C.POINTER(C.c_uint8))]
        L.znippy_index_metadata.argtypes = [vp, C.c_char_p]
        L.znippy_index_metadata.restype = C.c_char_p
        L.znippy_index_close.argtypes = [vp]
        L.znippy_index_close.restype = None
        L.znippy_interpret_footer.argtypes = [vp, C.c_size_t, C.POINTER(C.c_uint64)]
        L.znippy_write_manifest_bytes.argtypes = [C.POINTER(_ME), C.c_size_t, vp, C.c_size_t]
        L.znippy_write_manifest_bytes.restype = C.c_size_t
        _L = L
    return _L


class HostError(RuntimeError):
    pass


def _chk(rc, what):
    if rc:
        raise HostError(f"{what}: rc={rc} {lib().znippy_host_last_error().decode()}")


class StreamCompressor:
    """compress_stream(&output, no_skip) on the compiled host layer."""

    def __init__(self, output, no_skip=False, device=0):
        self.h = vp()
        _chk(lib().znippy_compress_stream(str(output).encode(), int(no_skip), device, C.byref(self.h)), "compress_stream")

    def sender(self):
        return self

    def send(self, entry):
        data = bytes(entry.data)
        _chk(lib().znippy_stream_send(self.h, entry.relative_path.encode(), data, len(data),
                                      -1 if entry.pkg_type is None else int(entry.pkg_type),
                                      None if entry.repo is None else entry.repo.encode()), "stream_send")

    def send_packed(self, paths, data, data_off, pkg_type=None, repo=None):
        """n entries in one call: `paths` a list of str, `data` one contiguous buffer (bytes / numpy uint8), `data_off`
        n + 1 offsets into it.  The per-entry work (chunking, the one copy into staging) is the same as send()'s; what goes
        away is n trips through the binding."""
        import numpy as np
        enc = [p.encode() for p in paths]
        n = len(enc)
        poff = np.zeros(n + 1, np.uint64)
        np.cumsum([len(p) for p in enc], out=poff[1:])
        pblob = b"".join(enc)
        doff = np.ascontiguousarray(np.asarray(data_off, dtype=np.uint64))
        assert doff.size == n + 1
        buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
        assert int(doff[-1]) <= buf.size
        pk = None if pkg_type is None else np.ascontiguousarray(np.asarray(pkg_type, dtype=np.int32))
        _chk(lib().znippy_stream_send_packed(self.h, n, pblob, poff.ctypes.data_as(vp), buf.ctypes.data_as(vp), doff.ctypes.data_as(vp),
                                             None if pk is None else pk.ctypes.data_as(vp),
                                             None if repo is None else repo.encode()), "stream_send_packed")

    def finish(self) -> ix.CompressionReport:
        r = _CR()
        h, self.h = self.h, None
        _chk(lib().znippy_stream_finish(h, C.byref(r)), "stream_finish")
        return _report(r)


def _report(r) -> ix.CompressionReport:
    return ix.CompressionReport(**{n: (float(getattr(r, n)) if n == "compression_ratio" else int(getattr(r, n)))
                                   for n, _ in _CR._fields_})


def compress_stream(output, no_skip=False, device=0):
    return StreamCompressor(output, no_skip, device)


def compress_dir(input_dir, output, no_skip=False, plugin=None, repo=None, device=0) -> ix.CompressionReport:
    """compress_dir(&input_dir, &output, no_skip, plugin, repo) on the compiled host layer."""
    if plugin is not None:
        raise NotImplementedError("metadata plugins are outside the hot path (SURVEY §2 #12)")
    r = _CR()
    _chk(lib().znippy_compress_dir(str(input_dir).encode(), str(output).encode(), int(no_skip),
                                   None if repo is None else repo.encode(), device, C.byref(r)), "compress_dir")
    return _report(r)


def decompress_archive(index_path, save_data, out_dir, device=0, rank=0, world=1) -> ix.VerifyReport:
    r = _VR()
    cap = 1 << 16
    corrupt = np.zeros(cap, dtype=np.uint64)
    n = C.c_uint64()
    _chk(lib().znippy_decompress_archive(str(index_path).encode(), int(save_data), str(out_dir).encode(), device, rank,
                                         world, C.byref(r), corrupt.ctypes.data_as(vp), cap, C.byref(n)), "decompress_archive")
    return ix.VerifyReport(**{k: int(getattr(r, k)) for k, _ in _VR._fields_},
                           corrupt_rows=[int(x) for x in corrupt[:min(n.value, cap)]])


class ZnippyArchive:
    def __init__(self, path, device=0):
        self.h = vp()
        _chk(lib().znippy_archive_open(str(path).encode(), device, C.byref(self.h)), "archive_open")

    @classmethod
    def open(cls, path, device=0):
        return cls(path, device)

    def file_count(self):
        return int(lib().znippy_archive_file_count(self.h))

    def contains(self, rel):
        return lib().znippy_archive_file_size(self.h, rel.encode()) >= 0

    def file_size(self, rel):
        s = lib().znippy_archive_file_size(self.h, rel.encode())
        return None if s < 0 else int(s)

    def extract_file(self, rel, verify=False) -> bytes:
        """verify=True: every chunk's BLAKE3 is checked against the index (the reference has no such option)."""
        size = self.file_size(rel)
        if size is None:
            raise KeyError(f"file not found in archive: {rel}")
        buf = np.empty(max(size, 1), dtype=np.uint8)
        w = C.c_size_t()
        f = lib().znippy_archive_extract_file_verified if verify else lib().znippy_archive_extract_file
        _chk(f(self.h, rel.encode(), buf.ctypes.data_as(vp), size, C.byref(w)), "extract_file")
        return buf[:w.value].tobytes()

    def close(self):
        if self.h:
            lib().znippy_archive_close(self.h)
            self.h = None

    __del__ = close


def read_index(path):
    """-> (rows: list of dicts, manifest: list[ManifestEntry], metadata getter) via the C++ Arrow IPC reader."""
    h = vp()
    _chk(lib().znippy_index_open(str(path).encode(), C.byref(h)), "index_open")
    try:
        L = lib()
        rows = []
        for i in range(L.znippy_index_rows(h)):
            p = C.c_char_p(); seq = C.c_uint32(); fo = C.c_uint64(); cm = C.c_int(); us = C.c_uint64()
            bo = C.c_uint64(); bs = C.c_uint64(); ck = C.POINTER(C.c_uint8)()
            L.znippy_index_row(h, i, C.byref(p), C.byref(seq), C.byref(fo), C.byref(cm), C.byref(us), C.byref(bo),
                               C.byref(bs), C.byref(ck))
            rows.append(dict(relative_path=p.value.decode(), chunk_seq=seq.value, fdata_offset=fo.value,
                             compressed=bool(cm.value), uncompressed_size=us.value, blob_offset=bo.value,
                             blob_size=bs.value, checksum=bytes(ck[:32])))
        manifest = []
        for i in range(L.znippy_index_manifest_len(h)):
            e = _ME()
            L.znippy_index_manifest_entry(h, i, C.byref(e))
            manifest.append(ix.ManifestEntry(e.pkg_type, e.repo.decode(), e.module_name.decode(), e.index_offset,
                                             e.index_len, e.row_count))
        keys = ["znippy_format_version", "max_core_in_flight", "max_core_in_compress", "max_mem_allowed",
                "min_free_memory_ratio", "file_split_block_size", "max_chunks", "compression_level",
                "zstd_output_buffer_size", "checksum_group_0"]
        md = {}
        for k in keys:
            v = L.znippy_index_metadata(h, k.encode())
            if v is not None:
                md[k] = v.decode()
        return rows, manifest, md
    finally:
        lib().znippy_index_close(h)


def write_manifest_bytes(entries) -> bytes:
    arr = (_ME * max(len(entries), 1))()
    keep = []
    for i, e in enumerate(entries):
        r, m = e.repo.encode(), e.module_name.encode()
        keep += [r, m]
        arr[i] = _ME(e.pkg_type, r, m, e.index_offset, e.index_len, e.row_count)
    n = lib().znippy_write_manifest_bytes(arr, len(entries), None, 0)
    buf = np.zeros(n, dtype=np.uint8)
    lib().znippy_write_manifest_bytes(arr, len(entries), buf.ctypes.data_as(vp), n)
    return buf.tobytes()


def interpret_footer(tail: bytes):
    off = C.c_uint64()
    multi = lib().znippy_interpret_footer(tail, len(tail), C.byref(off))
    return ("multi" if multi else "single", off.value)
