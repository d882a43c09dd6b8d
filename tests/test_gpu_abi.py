"""The C ABI's contract at its edges (include/znippy_hip.h): status codes instead of faults for bad arguments,
empty tables, foreign tables, short destinations, frames the codec does not support — codec.rs error behaviour
(a compress / decompress error is a Result::Err the caller sees, never a crash)."""
import ctypes as C

import numpy as np
import pytest

import gen

pytestmark = pytest.mark.gpu


def test_null_and_mismatched_arguments(gpu_ctx):
    from znippy_amd import _lib, hip
    L = _lib.lib()
    assert L.znippy_ctx_create(0, None, None) == _lib.E_INVAL
    assert L.znippy_decompress(None, None, 0, None, 0, None) == _lib.E_INVAL
    assert L.znippy_rows_create(gpu_ctx.h, None, None, None, None, None, None, 0, 1, None) == _lib.E_INVAL
    sz = C.c_uint64()
    assert L.znippy_get_decompressed_size(b"\x28\xb5", 2, C.byref(sz)) != 0            # truncated header
    assert L.znippy_get_decompressed_size(b"notzstd!!", 9, C.byref(sz)) != 0           # wrong magic
    # a table belongs to the context that made it
    other = hip.Context(0)
    rt = hip.RowTable(other, np.zeros(1, np.uint64), np.full(1, 9, np.uint64), np.full(1, 1, np.uint64), np.zeros(1, np.uint64))
    assert L.znippy_decode_verify_rows_async(gpu_ctx.h, rt.h, None, 0, None, 0) == _lib.E_INVAL
    rt.close(); other.close()


def test_empty_tables_are_fine(gpu_ctx):
    import torch
    from znippy_amd import hip
    e64 = np.zeros(0, np.uint64)
    rt = hip.RowTable(gpu_ctx, e64, e64, e64, e64)
    c, corrupt, status = rt.decode_verify(torch.zeros(64, dtype=torch.uint8, device="cuda"), torch.zeros(64, dtype=torch.uint8, device="cuda"))
    assert c["total_chunks"] == 0 and len(corrupt) == 0 and len(status) == 0
    ro = hip.RoundTable(gpu_ctx, e64, e64)
    assert ro.blob_bound() == 0
    enc = ro.encode_hash(torch.zeros(64, dtype=torch.uint8, device="cuda"), torch.zeros(64, dtype=torch.uint8, device="cuda"))
    assert int(enc["blob_bytes"]) == 0 and len(enc["blob_size"]) == 0


def test_short_destination_and_unsupported_frames(gpu_ctx, oracle):
    from znippy_amd import _lib
    L = _lib.lib()
    data = gen.pseudo_text(5000, seed=1)
    frame = gpu_ctx.compress(data)
    dst = (C.c_uint8 * 100)()
    w = C.c_size_t()
    assert L.znippy_decompress(gpu_ctx.h, frame, len(frame), dst, 100, C.byref(w)) == _lib.E_DST_SMALL
    out = (C.c_uint8 * 10)()
    assert L.znippy_compress(gpu_ctx.h, data, len(data), out, 10, C.byref(w)) == _lib.E_DST_SMALL
    # a frame without Frame_Content_Size cannot size the output (zl_get_decompressed_size would fail): unsupported
    nofcs = bytes([0x28, 0xB5, 0x2F, 0xFD, 0x00, 0x58, 0x01, 0x00, 0x00])     # window descriptor, empty raw last block
    sz = C.c_uint64()
    assert L.znippy_get_decompressed_size(nofcs, len(nofcs), C.byref(sz)) == _lib.E_UNSUPPORTED
    # dictionary id present: unsupported
    withdict = bytearray(frame)
    withdict[4] |= 1
    assert L.znippy_decompress(gpu_ctx.h, bytes(withdict), len(withdict), dst, 100, C.byref(w)) in (_lib.E_UNSUPPORTED, _lib.E_CORRUPT, _lib.E_DST_SMALL)
    # the error text of the context is readable
    assert isinstance(L.znippy_last_error(gpu_ctx.h), (bytes, type(None)))


def test_bound_covers_every_input(gpu_ctx):
    for n in (0, 1, 7, 1000, 131072, 131073, 1 << 20):
        for data in (gen.random_lcg(n), bytes(n)):
            assert len(gpu_ctx.compress(data)) <= gpu_ctx.compress_bound(n)
