"""Object wrappers over the C ABI (include/znippy_hip.h).  Device buffers are torch uint8 CUDA
tensors (PyTorch-ROCm is only the allocator/stream provider); every compute call goes through
libznippy_hip.so."""
import ctypes as C
import weakref

import numpy as np

from . import _lib
from ._lib import VerifyCounters, ZnippyError, as_np, np_ptr, vp


def _pinned(shape, dtype):
    """Host result buffer; pinned when torch can (direct DMA on D2H), reused across calls."""
    try:
        import torch
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        t = torch.empty(max(n, 1), dtype=torch.uint8, pin_memory=torch.cuda.is_available())
        return t.numpy()[:n].view(dtype).reshape(shape), t
    except Exception:  # no torch / no pinning: plain numpy
        return np.zeros(shape, dtype=dtype), None


def _dptr(t):
    if t is None:
        return None
    if isinstance(t, int):
        return vp(t)
    assert t.is_cuda and t.is_contiguous(), "device buffers must be contiguous CUDA tensors"
    # The context runs on its own non-blocking stream: whatever torch still has queued on ITS stream for this
    # buffer (a torch.zeros fill, a copy) must have landed before our kernels touch it.
    import torch
    ts = torch.cuda.current_stream(t.device)
    if not ts.query():
        ts.synchronize()
    return vp(t.data_ptr())


class Context:
    """One per worker / GPU — the analogue of one CompressCtx per thread (codec.rs:L8-28)."""

    def __init__(self, device=0, stream=None):
        self.L = _lib.lib()
        h = vp()
        rc = self.L.znippy_ctx_create(int(device), vp(stream) if stream else None, C.byref(h))
        if rc:
            raise ZnippyError(rc, "znippy_ctx_create")
        self.h = h
        self.device = int(device)
        self._tables = weakref.WeakSet()  # row / round tables created on this context: they die before it does

    def close(self):
        if getattr(self, "h", None):
            for t in list(getattr(self, "_tables", ())):  # a table destroyed after its context is a use-after-free in C
                t.close()
            self.L.znippy_ctx_destroy(self.h)
            self.h = None

    __del__ = close

    def _chk(self, rc, what):
        if rc:
            raise ZnippyError(rc, what, (self.L.znippy_last_error(self.h) or b"").decode())

    def sync(self):
        self._chk(self.L.znippy_ctx_sync(self.h), "znippy_ctx_sync")

    def kernel_times(self):
        names = (C.c_char_p * 48)()
        ms = (C.c_float * 48)()
        n = self.L.znippy_last_kernel_times(self.h, names, ms, 48)
        return [(names[i].decode(), float(ms[i])) for i in range(n)]

    def set_level(self, level):
        """CompressCtx::new(compression_level), codec.rs:L16-28: levels 1-3 fast tier, 4-22 higher effort tier."""
        self._chk(self.L.znippy_ctx_set_level(self.h, int(level)), "znippy_ctx_set_level")

    @property
    def level(self):
        return int(self.L.znippy_ctx_level(self.h))

    def set_kernel_timing(self, level):
        """2 = HIP events around every kernel (default), 1 = around the dominant read kernels only, 0 = none."""
        self._chk(self.L.znippy_ctx_set_kernel_timing(self.h, int(level)), "znippy_ctx_set_kernel_timing")

    def blake3_pass_ns(self):
        """Measured VALU floor: ns per 64-lane compress pass per SIMD (compressions only, nothing else running)."""
        v, g = C.c_float(), C.c_float()
        self._chk(self.L.znippy_measure_blake3_pass_ns(self.h, C.byref(v), C.byref(g)), "znippy_measure_blake3_pass_ns")
        self.ubench_ghz = float(g.value)
        return float(v.value)

    def last_shader_ghz(self):
        """Shader clock one wave of the small-row read kernel saw in the last run (context created with ZNIPPY_DBG & 32768)."""
        g = C.c_float()
        self._chk(self.L.znippy_last_shader_ghz(self.h, C.byref(g)), "znippy_last_shader_ghz")
        return float(g.value)

    # ---- single-chunk shims (codec.rs semantics, host buffers) ----
    def compress_bound(self, n):
        return int(self.L.znippy_compress_bound(n))

    def blake3(self, data) -> bytes:
        a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        out = (C.c_uint8 * 32)()
        self._chk(self.L.znippy_blake3(self.h, np_ptr(a) if a.size else None, a.size, out), "znippy_blake3")
        return bytes(out)

    def decompress(self, frame) -> bytes:
        a = np.frombuffer(frame, dtype=np.uint8)
        n = get_decompressed_size(frame)
        out = np.empty(max(n, 1), dtype=np.uint8)
        w = C.c_size_t()
        self._chk(self.L.znippy_decompress(self.h, np_ptr(a), a.size, np_ptr(out), n, C.byref(w)), "znippy_decompress")
        return out[:w.value].tobytes()

    def compress(self, data) -> bytes:
        a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        cap = self.compress_bound(a.size)
        out = np.empty(cap, dtype=np.uint8)
        w = C.c_size_t()
        self._chk(self.L.znippy_compress(self.h, np_ptr(a) if a.size else None, a.size, np_ptr(out), cap, C.byref(w)),
                  "znippy_compress")
        return out[:w.value].tobytes()


def get_decompressed_size(frame) -> int:
    a = np.frombuffer(frame, dtype=np.uint8)
    v = C.c_uint64()
    rc = _lib.lib().znippy_get_decompressed_size(np_ptr(a), a.size, C.byref(v))
    if rc:
        raise ZnippyError(rc, "znippy_get_decompressed_size")
    return v.value


class RowTable:
    """Device-resident slice [row_begin,row_end) of the index columns + its work plan."""

    def __init__(self, ctx, blob_offset, blob_size, uncompressed_size, out_offset, compressed_bitmap=None,
                 checksum=None, row_begin=0, row_end=None):
        self.ctx = ctx
        bo = as_np(blob_offset, np.uint64)
        bs = as_np(blob_size, np.uint64)
        us = as_np(uncompressed_size, np.uint64)
        oo = as_np(out_offset, np.uint64)
        n = len(bo)
        row_end = n if row_end is None else row_end
        bm = as_np(compressed_bitmap, np.uint8) if compressed_bitmap is not None else None
        ck = as_np(checksum, np.uint8).reshape(-1) if checksum is not None else None
        h = vp()
        ctx._chk(ctx.L.znippy_rows_create(ctx.h, np_ptr(bo), np_ptr(bs), np_ptr(bm) if bm is not None else None,
                                          np_ptr(us), np_ptr(oo), np_ptr(ck) if ck is not None else None,
                                          row_begin, row_end, C.byref(h)), "znippy_rows_create")
        self.h = h
        ctx._tables.add(self)
        self.row_begin, self.row_end = row_begin, row_end
        self.n = row_end - row_begin
        self._corrupt, self._k1 = _pinned((max(self.n, 1),), np.uint64)
        self._status, self._k2 = _pinned((max(self.n, 1),), np.int32)

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "h", None):
                self.ctx.L.znippy_rows_destroy(self.h)
            self.h = None

    __del__ = close

    def decode_verify_async(self, d_blobs, d_out, blob_base=0, out_cap=None, blob_cap=None):
        out_cap = d_out.numel() if out_cap is None else out_cap
        if blob_cap is None and not isinstance(d_blobs, int):
            blob_cap = d_blobs.numel()
        if blob_cap is not None:  # the blob region's size is known: rows pointing outside it become error codes
            self.ctx._chk(self.ctx.L.znippy_rows_set_blob_cap(self.h, int(blob_cap)), "znippy_rows_set_blob_cap")
        self.ctx._chk(self.ctx.L.znippy_decode_verify_rows_async(self.ctx.h, self.h, _dptr(d_blobs), blob_base,
                                                                 _dptr(d_out), out_cap),
                      "znippy_decode_verify_rows_async")

    def results(self, want_status=True):
        c = VerifyCounters()
        corrupt, status = self._corrupt, self._status
        self.ctx._chk(self.ctx.L.znippy_rows_results(self.ctx.h, self.h, C.byref(c), np_ptr(corrupt), corrupt.size,
                                                     np_ptr(status) if want_status else None), "znippy_rows_results")
        return c.as_dict(), corrupt[:min(c.corrupt_rows, corrupt.size)].copy(), status[:self.n]

    def results_lagged(self, lag=1):
        """Counters of the run `lag` runs before the latest queued one (waits for that run only)."""
        c = VerifyCounters()
        self.ctx._chk(self.ctx.L.znippy_rows_results_lagged(self.ctx.h, self.h, lag, C.byref(c)),
                      "znippy_rows_results_lagged")
        return c.as_dict()

    def foreign_stats(self):
        """Last run's two-phase path for foreign multi-block frames: pool use, frames it decoded, blocks it gave up."""
        st = (C.c_uint64 * 8)()
        self.ctx._chk(self.ctx.L.znippy_rows_foreign_stats(self.ctx.h, self.h, st), "znippy_rows_foreign_stats")
        return dict(lit_pool_bytes=int(st[0]), seq_pool_records=int(st[1]), frames=int(st[2]), blocks_given_up=int(st[3]),
                    given_up_error=int(st[4]), given_up_table_far=int(st[5]), given_up_pool=int(st[6]), given_up_range=int(st[7]))

    def decode_verify(self, d_blobs, d_out, blob_base=0, out_cap=None, blob_cap=None):
        self.decode_verify_async(d_blobs, d_out, blob_base, out_cap, blob_cap)
        return self.results()

    def digests(self):
        out = np.zeros((max(self.n, 1), 32), dtype=np.uint8)
        self.ctx._chk(self.ctx.L.znippy_rows_digests(self.ctx.h, self.h, np_ptr(out)), "znippy_rows_digests")
        return out[:self.n]


class RoundTable:
    """Device-resident batch of Rounds (offset,len,skip) over one staging buffer."""

    def __init__(self, ctx, src_offset, length, skip=None):
        self.ctx = ctx
        so = as_np(src_offset, np.uint64)
        ln = as_np(length, np.uint64)
        sk = as_np(skip, np.uint8) if skip is not None else None
        self.n = len(so)
        self._compressed = (1 - sk).astype(np.uint8) if sk is not None else np.ones(self.n, np.uint8)
        h = vp()
        ctx._chk(ctx.L.znippy_rounds_create(ctx.h, np_ptr(so), np_ptr(ln), np_ptr(sk) if sk is not None else None,
                                            self.n, C.byref(h)), "znippy_rounds_create")
        self.h = h
        ctx._tables.add(self)

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "h", None):
                self.ctx.L.znippy_rounds_destroy(self.h)
            self.h = None

    __del__ = close

    def set_store_incompressible(self, on=True):
        """Opt-in: rounds whose frame is not smaller than the input are emitted as-is (compressed=0)."""
        self.ctx._chk(self.ctx.L.znippy_rounds_set_store_incompressible(self.h, int(on)), "set_store_incompressible")
        self._store_inc = bool(on)

    def blob_bound(self):
        return int(self.ctx.L.znippy_rounds_blob_bound(self.h))

    def hash(self, d_src):
        out = np.zeros((max(self.n, 1), 32), dtype=np.uint8)
        self.ctx._chk(self.ctx.L.znippy_hash_rounds(self.ctx.h, self.h, _dptr(d_src), np_ptr(out)), "znippy_hash_rounds")
        return out[:self.n]

    def encode_hash_async(self, d_src, d_blob_out, blob_cap=None):
        blob_cap = d_blob_out.numel() if blob_cap is None else blob_cap
        self.ctx._chk(self.ctx.L.znippy_encode_hash_rounds_async(self.ctx.h, self.h, _dptr(d_src), _dptr(d_blob_out),
                                                                 blob_cap), "znippy_encode_hash_rounds_async")

    def results(self):
        """Per-round outputs as numpy VIEWS of the table's pinned result mirror (zero-copy; they are
        overwritten by the next encode call on this table — copy what must outlive it)."""
        bo, bs, ck = vp(), vp(), vp()
        total = C.c_uint64()
        self.ctx._chk(self.ctx.L.znippy_rounds_results_view(self.ctx.h, self.h, C.byref(bo), C.byref(bs), C.byref(ck),
                                                            C.byref(total)), "znippy_rounds_results_view")
        k = self.n
        if k == 0:
            return dict(blob_offset=np.zeros(0, np.uint64), blob_size=np.zeros(0, np.uint64),
                        checksum=np.zeros((0, 32), np.uint8), compressed=np.zeros(0, np.uint8), blob_bytes=0)
        mk = lambda p, nbytes, dt: np.frombuffer((C.c_uint8 * nbytes).from_address(p.value), dtype=dt)
        comp = self._compressed
        if getattr(self, "_store_inc", False):  # the device decided per round: take the full (copying) result call
            comp = np.zeros(k, dtype=np.uint8)
            self.ctx._chk(self.ctx.L.znippy_rounds_results(self.ctx.h, self.h, None, None, None, np_ptr(comp), None),
                          "znippy_rounds_results")
        return dict(blob_offset=mk(bo, 8 * k, np.uint64), blob_size=mk(bs, 8 * k, np.uint64),
                    checksum=mk(ck, 32 * k, np.uint8).reshape(k, 32), compressed=comp, blob_bytes=int(total.value))

    def results_lagged(self, lag=1):
        """Views of the results of the run `lag` runs before the latest queued one (compressed[] as known to the host)."""
        bo, bs, ck = vp(), vp(), vp()
        total = C.c_uint64()
        self.ctx._chk(self.ctx.L.znippy_rounds_results_lagged(self.ctx.h, self.h, lag, C.byref(bo), C.byref(bs),
                                                              C.byref(ck), C.byref(total)), "znippy_rounds_results_lagged")
        k = self.n
        mk = lambda p, nbytes, dt: np.frombuffer((C.c_uint8 * nbytes).from_address(p.value), dtype=dt)
        return dict(blob_offset=mk(bo, 8 * k, np.uint64), blob_size=mk(bs, 8 * k, np.uint64),
                    checksum=mk(ck, 32 * k, np.uint8).reshape(k, 32), compressed=self._compressed,
                    blob_bytes=int(total.value))

    def encode_hash(self, d_src, d_blob_out, blob_cap=None):
        self.encode_hash_async(d_src, d_blob_out, blob_cap)
        return self.results()
