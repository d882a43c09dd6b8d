"""Host side of the table constructors (no GPU): the vectorised one-pass reductions that size a table's device arrays
(csrc/host/extents.cc) against a plain restatement of the rules they encode — Round pieces (stream_packer.rs:L184-202
chunk rule -> 128 KiB zstd blocks / 64 KiB store-path pieces), the compress bound, and a row table's extents
(decompress.rs:L143-190: where a row's blob lies and where its bytes land)."""
import ctypes as C

import numpy as np

from znippy_amd import _lib


def _slot(n):
    return ((16 + 16 + 2 * n + (n >> 2) + 64) + 15) & ~15


def test_rounds_totals_match_the_per_round_rules():
    L = _lib.lib()
    L.zn_rounds_totals.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    rng = np.random.default_rng(3)
    cases = [np.array([], np.uint64), np.array([0], np.uint64), np.full(1000, 10240, np.uint64),
             np.array([0, 1, 16384, 16385, 131071, 131072, 131073, 262144, 8 << 20, (200 << 20) + 5, 65535, 65536, 65537], np.uint64),
             rng.integers(0, 3 << 20, size=5000).astype(np.uint64)]
    for lens in cases:
        for skip in (None, (rng.integers(0, 3, size=len(lens)) == 0).astype(np.uint8)):
            out = (C.c_uint64 * 8)()
            L.zn_rounds_totals(lens.ctypes.data_as(C.c_void_p), None if skip is None else skip.ctypes.data_as(C.c_void_p), len(lens), out)
            items = prov = small = wide = bound = enc = 0
            for i, n in enumerate(int(x) for x in lens):
                if skip is not None and skip[i]:
                    items += max(1, -(-n // 65536)); bound += n
                    continue
                nb = max(1, -(-n // 131072))
                items += nb
                for k in range(nb):
                    bl = min(131072, n - k * 131072)
                    prov += _slot(bl)
                    if bl > 16384: wide += 1
                    else: small += 1
                bound += n + 3 * (n // 131072 + 1) + 19
                enc += n
            odd = int(any(int(x) & 15 for x in lens[:-1]))
            assert list(out) == [items, prov, small, wide, bound, int(lens.sum()), enc, odd], (len(lens), skip is not None)


def test_rows_extents_match_numpy():
    L = _lib.lib()
    L.zn_rows_extents.argtypes = [C.c_void_p] * 4 + [C.c_size_t, C.c_void_p]
    rng = np.random.default_rng(5)
    for n in (1, 7, 64, 1000, 100_000):
        bs = rng.integers(1, 5000, size=n).astype(np.uint64)
        bo = (np.cumsum(bs) - bs + 12345).astype(np.uint64)
        us = rng.integers(0, 300_000, size=n).astype(np.uint64)
        oo = (np.cumsum(us) - us).astype(np.uint64)
        if n > 5:
            oo[3] += 5  # an output offset that is not a multiple of 16
        out = (C.c_uint64 * 8)()
        L.zn_rows_extents(*(a.ctypes.data_as(C.c_void_p) for a in (bo, bs, oo, us)), n, out)
        nblk = int(((us + np.uint64(131071)) >> np.uint64(17)).sum() + (us == 0).sum())
        assert list(out)[:3] == [int(bo.min()), int((bo + bs).max()), int((oo + us).max())]
        assert out[3] == 0 and out[4] == int(us.sum()) and out[5] == nblk and out[6] == int((us > 65536).sum())
        assert (out[7] != 0) == bool((oo & np.uint64(15)).any())
    # wrap-around of blob_offset + blob_size is flagged
    bo = np.array([2**64 - 10], np.uint64); bs = np.array([100], np.uint64); z = np.zeros(1, np.uint64)
    out = (C.c_uint64 * 8)()
    L.zn_rows_extents(*(a.ctypes.data_as(C.c_void_p) for a in (bo, bs, z, z)), 1, out)
    assert out[3] != 0


def test_rows_pack32_recognises_front_to_back_tables():
    """zn_rows_pack32 (csrc/host/extents.cc): a table whose blobs and rows follow one another (what the reference's writer lays
    down: running blob offsets, index.rs / decompress.rs:L135-190 reads them back in order) is sent to the device as its two
    size columns, 32 bits each; anything else — a gap, an overlap, a size of 4 GiB or more — is not."""
    L = _lib.lib()
    L.zn_rows_pack32.argtypes = [C.c_void_p] * 4 + [C.c_size_t, C.c_void_p]
    L.zn_rows_pack32.restype = C.c_int
    rng = np.random.default_rng(6)

    def call(bo, bs, oo, us):
        dst = np.zeros(2 * len(bo), np.uint32)
        rc = L.zn_rows_pack32(*(a.ctypes.data_as(C.c_void_p) for a in (bo, bs, oo, us)), len(bo), dst.ctypes.data_as(C.c_void_p))
        return rc, dst

    for n in (1, 2, 63, 1000, 100_001):
        bs = rng.integers(0, 5000, size=n).astype(np.uint64)
        us = rng.integers(0, 300_000, size=n).astype(np.uint64)
        bo = (np.cumsum(bs) - bs + 777).astype(np.uint64)
        oo = (np.cumsum(us) - us + 5).astype(np.uint64)
        rc, dst = call(bo, bs, oo, us)
        assert rc == 1 and (dst[:n] == bs).all() and (dst[n:] == us).all()
        if n >= 3:
            for col, k in ((bo, n // 2), (oo, n - 1), (bo, 1)):
                old = int(col[k]); col[k] = old + 1
                assert call(bo, bs, oo, us)[0] == 0          # a gap (or an overlap) in either running sum
                col[k] = old
            big = bs.copy(); big[n // 3] = 1 << 32
            bo2 = (np.cumsum(big) - big + 777).astype(np.uint64)
            assert call(bo2, big, oo, us)[0] == 0            # a size that does not fit 32 bits
            assert call(bo, bs, oo, us)[0] == 1
