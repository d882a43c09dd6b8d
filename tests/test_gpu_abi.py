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


def _table(gpu_ctx, oracle, entries, comp, level=3):
    """Rows over a packed blob region: entries[i] compressed with libzstd when comp[i] else stored."""
    frames = [oracle.libzstd_compress(e, level) if c else e for e, c in zip(entries, comp)]
    bs = np.array([len(f) for f in frames], dtype=np.uint64)
    bo = np.concatenate([[0], np.cumsum(bs)[:-1]]).astype(np.uint64)
    us = np.array([len(e) for e in entries], dtype=np.uint64)
    oo = np.concatenate([[0], np.cumsum(us)[:-1]]).astype(np.uint64)
    ck = np.stack([np.frombuffer(oracle.blake3(e), dtype=np.uint8) for e in entries])
    return b"".join(frames), bo, bs, us, oo, np.packbits(np.asarray(comp, dtype=bool), bitorder="little"), ck


def test_stored_row_is_its_blob_whatever_the_index_says(gpu_ctx, oracle):
    """a stored row (compressed = 0) is hashed, copied and counted with blob_size bytes — what the
    reference does (`&read_buf`, decompress.rs:L143-166) — so an index row whose uncompressed_size disagrees can
    not make the store-path kernels read past the blob."""
    import torch
    from znippy_amd import hip
    entries = [gen.incompressible(3, 5000), gen.text(10240), gen.incompressible(4, 200000)]
    blob, bo, bs, us, oo, bitmap, ck = _table(gpu_ctx, oracle, entries, [0, 1, 0])
    lied = us.copy()
    lied[0] = 50_000            # small stored row claims 10x its blob
    lied[2] = 900_000           # big stored row (slices) claims 4.5x its blob
    d_blobs = torch.from_numpy(np.frombuffer(blob, dtype=np.uint8).copy()).cuda()      # NO padding behind the blobs
    d_out = torch.zeros(int(us.sum()) + 64, dtype=torch.uint8, device="cuda")
    rt = hip.RowTable(gpu_ctx, bo, bs, lied, oo, bitmap, ck)
    counters, corrupt, status = rt.decode_verify(d_blobs, d_out)
    assert (status == 0).all() and len(corrupt) == 0
    assert counters["total_written_bytes"] == int(us.sum()) == counters["verified_bytes"]
    assert d_out[:int(us.sum())].cpu().numpy().tobytes() == b"".join(entries)


def test_rows_outside_the_blob_or_output_region_are_error_codes(gpu_ctx, oracle):
    """Bad source ranges (crafted / truncated index) and short destinations, for stored and compressed rows, small
    and big: ZNIPPY_E_CORRUPT / ZNIPPY_E_DST_SMALL per row, counted as decode errors, nothing written for them,
    nothing written past out_cap (guard bytes), the good rows unaffected."""
    import torch
    from znippy_amd import _lib, hip
    entries = [gen.text(10240), gen.incompressible(1, 3000), gen.binary(300000), gen.incompressible(2, 150000),
               gen.pseudo_text(20000, 3), gen.incompressible(5, 777)]
    comp = [1, 0, 1, 0, 1, 0]
    blob, bo, bs, us, oo, bitmap, ck = _table(gpu_ctx, oracle, entries, comp)
    total = int(us.sum())
    d_blobs = torch.from_numpy(np.frombuffer(blob, dtype=np.uint8).copy()).cuda()
    # (1) sources outside the region: offsets past the end, a size that runs over the end, offset + size wrapping
    bad_bo, bad_bs = bo.copy(), bs.copy()
    bad_bo[1] = len(blob) + 10                    # small stored row: starts behind the region
    bad_bs[3] = np.uint64(len(blob))              # big stored row: runs over the end
    bad_bo[4] = np.uint64(2**64 - 100)            # compressed row: offset + size wraps
    d_out = torch.zeros(total + 64, dtype=torch.uint8, device="cuda")
    rt = hip.RowTable(gpu_ctx, bad_bo, bad_bs, us, oo, bitmap, ck)
    counters, corrupt, status = rt.decode_verify(d_blobs, d_out, blob_cap=len(blob))
    assert list(status) == [0, _lib.E_CORRUPT, 0, _lib.E_CORRUPT, _lib.E_CORRUPT, 0]
    assert counters["decode_errors"] == 3 and counters["total_chunks"] == 6 and counters["corrupt_rows"] == 0
    good = [0, 2, 5]
    assert counters["verified_bytes"] == sum(len(entries[i]) for i in good)
    host = d_out.cpu().numpy()
    for i in good:
        assert host[int(oo[i]):int(oo[i] + us[i])].tobytes() == entries[i]
    for i in (1, 3, 4):
        assert not host[int(oo[i]):int(oo[i] + us[i])].any()      # untouched
    rt.close()
    # (2) destination too short: rows that would end behind out_cap are refused, the guard bytes stay
    cut = int(oo[3]) + 1000                        # rows 3, 4, 5 do not fit
    d_out = torch.full((total + 64,), 0xAB, dtype=torch.uint8, device="cuda")
    rt = hip.RowTable(gpu_ctx, bo, bs, us, oo, bitmap, ck)
    counters, corrupt, status = rt.decode_verify(d_blobs, d_out, out_cap=cut, blob_cap=len(blob))
    assert list(status) == [0, 0, 0, _lib.E_DST_SMALL, _lib.E_DST_SMALL, _lib.E_DST_SMALL]
    host = d_out.cpu().numpy()
    assert (host[int(oo[3]):] == 0xAB).all()
    assert host[:int(oo[3])].tobytes() == b"".join(entries[:3])
    assert counters["decode_errors"] == 3 and counters["verified_bytes"] == int(oo[3])
    # the same table run against a big enough destination afterwards: the verdicts are per (region, cap), not sticky
    d_out.zero_()
    counters, corrupt, status = rt.decode_verify(d_blobs, d_out, blob_cap=len(blob))
    assert (status == 0).all() and counters["verified_bytes"] == total
    rt.close()


def test_lagged_results_pipeline_two_runs_in_flight(gpu_ctx, oracle):
    """znippy_rows_results_lagged / znippy_rounds_results_lagged: run k's results while run k+1 executes."""
    import torch
    from znippy_amd import _lib, hip
    from znippy_amd._lib import ZnippyError
    n, sz = 600, 10240
    chunk = np.frombuffer(gen.text(sz), dtype=np.uint8)
    d_src = torch.from_numpy(np.tile(chunk, n)).cuda()
    lens = np.full(n, sz, np.uint64)
    offs = np.arange(n, dtype=np.uint64) * sz
    rounds = hip.RoundTable(gpu_ctx, offs, lens)
    d_blob = torch.zeros(rounds.blob_bound() + 64, dtype=torch.uint8, device="cuda")
    with pytest.raises(ZnippyError):
        rounds.results_lagged(0)                  # nothing queued yet
    rounds.encode_hash_async(d_src, d_blob)
    first = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in rounds.results_lagged(0).items()}
    want_ck = oracle.blake3(chunk)
    assert all(first["checksum"][i].tobytes() == want_ck for i in (0, n // 2, n - 1))
    for _ in range(3):                             # k+1 queued, k read
        rounds.encode_hash_async(d_src, d_blob)
        prev = rounds.results_lagged(1)
        assert prev["blob_bytes"] == first["blob_bytes"] and np.array_equal(prev["blob_size"], first["blob_size"])
        assert np.array_equal(prev["checksum"], first["checksum"])
    last = rounds.results()
    assert np.array_equal(last["blob_offset"], first["blob_offset"])
    rows = hip.RowTable(gpu_ctx, first["blob_offset"], first["blob_size"], lens, offs, None, first["checksum"])
    d_out = torch.zeros(n * sz + 64, dtype=torch.uint8, device="cuda")
    with pytest.raises(ZnippyError):
        rows.results_lagged(0)
    rows.decode_verify_async(d_blob, d_out)
    with pytest.raises(ZnippyError):
        rows.results_lagged(1)                     # only one run so far
    for _ in range(4):
        rows.decode_verify_async(d_blob, d_out)
        c = rows.results_lagged(1)
        assert c["verified_bytes"] == n * sz and c["corrupt_rows"] == 0 and c["total_chunks"] == n
    c, corrupt, status = rows.results()
    assert c["verified_bytes"] == n * sz and (status == 0).all()
    assert torch.equal(d_out[:n * sz], d_src)


def test_skippable_frames_in_front_of_a_frame(gpu_ctx, oracle):
    """the size query steps over leading skippable frames (RFC 8878 3.1.2) — so does the decoder."""
    import struct
    from znippy_amd import hip
    data = gen.pseudo_text(30000, 9)
    frame = oracle.libzstd_compress(data, 3)
    skippable = struct.pack("<II", 0x184D2A53, 11) + b"hello world" + struct.pack("<II", 0x184D2A50, 0)
    assert hip.get_decompressed_size(skippable + frame) == len(data)
    assert gpu_ctx.decompress(skippable + frame) == data
    from znippy_amd._lib import ZnippyError
    with pytest.raises(ZnippyError):
        hip.get_decompressed_size(struct.pack("<II", 0x184D2A50, 500) + frame[:20])   # skippable frame runs past the input


def test_tables_outliving_their_context_are_closed_with_it():
    """A table collected after Context.close() used to call its destroy entry point with a freed context."""
    import gc
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from znippy_amd import hip
    ctx = hip.Context(0)
    rt = hip.RowTable(ctx, np.zeros(1, np.uint64), np.array([4], np.uint64), np.array([4], np.uint64), np.zeros(1, np.uint64),
                      np.zeros(1, np.uint8), np.zeros((1, 32), np.uint8))
    rd = hip.RoundTable(ctx, np.zeros(1, np.uint64), np.array([4], np.uint64))
    ctx.close()
    assert rt.h is None and rd.h is None
    del rt, rd
    gc.collect()
    ctx.close()  # idempotent

def test_c_abi_tables_keep_a_destroyed_context_alive():
    """The C ABI's lifetime rule (include/znippy_hip.h), straight through ctypes with no wrapper in between: a table holds
    a reference to its context.  znippy_ctx_destroy with tables alive closes the context — every later call on it is
    ZNIPPY_E_INVAL, never a use of freed memory — and the last znippy_rows_destroy / znippy_rounds_destroy releases it."""
    import ctypes as C
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from znippy_amd import _lib
    L = _lib.lib()
    E_INVAL = -1
    ctx = C.c_void_p()
    assert L.znippy_ctx_create(0, None, C.byref(ctx)) == 0
    u64 = lambda *v: (C.c_uint64 * len(v))(*v)
    rows, rounds = C.c_void_p(), C.c_void_p()
    ck = (C.c_uint8 * 32)()
    assert L.znippy_rows_create(ctx, u64(0), u64(4), None, u64(4), u64(0), ck, 0, 1, C.byref(rows)) == 0
    assert L.znippy_rounds_create(ctx, u64(0), u64(4), None, 1, C.byref(rounds)) == 0
    L.znippy_ctx_destroy(ctx)                      # tables alive: the context is closed, not freed
    d = torch.zeros(256, dtype=torch.uint8, device="cuda")
    assert L.znippy_decode_verify_rows_async(ctx, rows, C.c_void_p(d.data_ptr()), 0, C.c_void_p(d.data_ptr() + 128), 64) == E_INVAL
    dig = (C.c_uint8 * 32)()
    assert L.znippy_hash_rounds(ctx, rounds, C.c_void_p(d.data_ptr()), dig) == E_INVAL
    assert L.znippy_blake3(ctx, b"abc", 3, dig) == E_INVAL
    more = C.c_void_p()
    assert L.znippy_rows_create(ctx, u64(0), u64(4), None, u64(4), u64(0), ck, 0, 1, C.byref(more)) == E_INVAL
    L.znippy_ctx_destroy(ctx)                      # idempotent while closing
    L.znippy_rows_destroy(rows)
    L.znippy_rounds_destroy(rounds)                # the last table: the context goes with it
    ctx2 = C.c_void_p()                            # the device is fine afterwards
    assert L.znippy_ctx_create(0, None, C.byref(ctx2)) == 0
    assert L.znippy_blake3(ctx2, b"abc", 3, dig) == 0
    assert bytes(dig).hex().startswith("6437b3ac38465133")
    L.znippy_ctx_destroy(ctx2)


@pytest.mark.parametrize("tiles", [
    ("......", "......", "......", "v.v..."), ("v.v...", "......", "......", "......"), ("vv....", "vvvvvv", "v.....", "vv.v.v"),
    ("......", "..v..v", "......", "......"), ("v.v.v.", "v.v.v.", "v.v.v.", "v.v.v."), ("vvvvvv", "......", ".v.v.v", "......")])
def test_valid_rows_between_undecodable_ones_keep_their_digests(gpu_ctx, oracle, tiles):
    """Tiles of six 10 KiB rows in which valid ('v') and undecodable ('.') rows alternate: the parent fold works on a
    table of the workgroup's HASHED units, compacted per tile; a unit that is not hashed between two that are pushed an
    inactive entry into the compacted table and the last hashed unit out of it — and when that entry was the table's
    last, the table still looked well-formed: the left-out row's digest was never written and a valid row came back
    corrupt.  Four tiles = one workgroup of the small-row kernel."""
    import torch
    from znippy_amd import hip
    data = gen.text(10240)
    good = gpu_ctx.compress(data)
    bad = bytearray(good); bad[0] ^= 0xFF                       # magic number: every decoder refuses it
    pattern = "".join(tiles)
    rows = [(good if ch == "v" else bytes(bad)) for ch in pattern]
    n = len(rows)
    bs = np.array([len(f) for f in rows], np.uint64)
    bo = np.concatenate([[0], np.cumsum(bs)[:-1]]).astype(np.uint64)
    us = np.full(n, len(data), np.uint64)
    oo = np.arange(n, dtype=np.uint64) * len(data)
    ck = np.tile(np.frombuffer(oracle.blake3(data), dtype=np.uint8), (n, 1))
    d_blobs = torch.from_numpy(np.frombuffer(b"".join(rows) + bytes(64), dtype=np.uint8).copy()).cuda()
    d_out = torch.zeros(n * len(data) + 64, dtype=torch.uint8, device="cuda")
    rt = hip.RowTable(gpu_ctx, bo, bs, us, oo, None, ck)
    c, corrupt, status = rt.decode_verify(d_blobs, d_out)
    valid = np.array([ch == "v" for ch in pattern])
    assert (status[valid] == 0).all() and (status[~valid] < 0).all(), status
    assert c["corrupt_rows"] == 0, [int(x) for x in corrupt]
    assert c["decode_errors"] == int((~valid).sum()) and c["verified_bytes"] == int(valid.sum()) * len(data)
    assert (rt.digests()[valid] == ck[valid]).all()


def test_front_to_back_tables_are_built_from_their_size_columns(oracle):
    """A table whose offsets are the running sums of its sizes goes to the device as two 32-bit columns and the 64-bit columns
    are made there (k_rows_unpack_*); ZNIPPY_NO_PACK=1 sends the four columns as they are.  Same results either way — for a
    front-to-back table, for one with a gap (sent as it is in both contexts), with stored rows, with a row_begin window."""
    import os
    import torch
    import gen
    from znippy_amd import hip
    rng = np.random.default_rng(8)
    n = 1500
    entries = [gen.pseudo_text(int(rng.integers(0, 6000)), seed=i) if i % 3 else gen.text(int(rng.integers(1, 12000))) for i in range(n)]
    ctx0 = hip.Context(0)
    frames = [ctx0.compress(e) if i % 7 else e for i, e in enumerate(entries)]   # every 7th row stored
    comp = np.array([1 if i % 7 else 0 for i in range(n)], np.uint8)
    ctx0.close()
    bs = np.array([len(f) for f in frames], np.uint64)
    us = np.array([len(e) for e in entries], np.uint64)
    ck = np.stack([np.frombuffer(oracle.blake3(e), dtype=np.uint8) for e in entries])
    bm = np.packbits(comp.astype(bool), bitorder="little")
    results = {}
    for gap in (0, 3):
        bo = (np.cumsum(bs) - bs).astype(np.uint64)
        oo = (np.cumsum(us) - us).astype(np.uint64)
        if gap:
            bo[n // 2:] += np.uint64(gap); oo[n // 3:] += np.uint64(16)
        blob = np.zeros(int(bo[-1] + bs[-1]) + 64, np.uint8)
        for i, f in enumerate(frames):
            blob[int(bo[i]):int(bo[i]) + len(f)] = np.frombuffer(f, np.uint8)
        d_blobs = torch.from_numpy(blob).cuda()
        for mode in ("pack", "nopack"):
            if mode == "nopack": os.environ["ZNIPPY_NO_PACK"] = "1"
            try:
                ctx = hip.Context(0)
            finally:
                os.environ.pop("ZNIPPY_NO_PACK", None)
            d_out = torch.zeros(int(oo[-1] + us[-1]) + 64, dtype=torch.uint8, device="cuda")
            rt = hip.RowTable(ctx, bo, bs, us, oo, bm, ck)
            c, corrupt, status = rt.decode_verify(d_blobs, d_out)
            out = d_out.cpu().numpy()
            assert c["corrupt_rows"] == 0 and c["decode_errors"] == 0 and (status == 0).all(), (gap, mode, c)
            for i in (0, 1, n // 2 - 1, n // 2, n - 1):
                assert out[int(oo[i]):int(oo[i] + us[i])].tobytes() == entries[i]
            results[(gap, mode)] = (dict(c), out.copy())
            rt.close(); ctx.close()
        assert results[(gap, "pack")][0] == results[(gap, "nopack")][0]
        assert (results[(gap, "pack")][1] == results[(gap, "nopack")][1]).all()
