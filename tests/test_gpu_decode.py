"""GPU parity: zstd frame decode + verify through the C ABI vs the oracle (bit-exact bytes).

Replaces codec::decompress_into (codec.rs:L67-78) and the read worker loop body
(decompress.rs:L135-190).  Frames come from the container's libzstd (an independent RFC 8878
encoder) at several levels, so Huffman literals (1 and 4 streams, FSE-compressed weights),
FSE-compressed / RLE / repeat sequence tables, repeat offsets, raw and RLE blocks are all hit.
"""
import json
import os

import numpy as np
import pytest

import gen

pytestmark = pytest.mark.gpu

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "zstd_frames.json")))


def _gen(name, n):
    return bytes(n) if name == "zeros" else getattr(gen, name)(n)


def test_golden_frames_shim(gpu_ctx, oracle):
    for fr in GOLD["frames"]:
        frame = bytes.fromhex(fr["frame_hex"])
        want = _gen(fr["gen"], fr["size"])
        got = gpu_ctx.decompress(frame)
        assert got == want, (fr["gen"], fr["size"], fr["level"])
        assert gpu_ctx.blake3(got).hex() == fr["blake3"]


CASES = [
    ("text", 10240), ("binary", 10240), ("random_lcg", 10240), ("pseudo_text", 5000), ("pseudo_text", 300000),
    ("text", 1), ("text", 0), ("zeros", 200000), ("text", 1 << 20), ("pseudo_text", 1 << 20),
    ("binary", 3 << 20), ("random_lcg", 300000),
]


@pytest.mark.parametrize("gname,n", CASES)
@pytest.mark.parametrize("level", [1, 3, 19])
def test_shim_vs_libzstd_frames(gpu_ctx, oracle, gname, n, level):
    data = _gen(gname, n)
    frame = oracle.libzstd_compress(data, level)
    assert oracle.zstd_decompress(frame) == data  # oracle agrees first
    assert gpu_ctx.decompress(frame) == data


def _build_archive(oracle, entries, level=19, skip=None):
    """entries: list of bytes; returns dict of index columns + blob region (oracle write loop)."""
    src = np.frombuffer(b"".join(entries) + b"\0" * 16, dtype=np.uint8)
    lens = np.array([len(e) for e in entries], dtype=np.uint64)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    skip = np.zeros(len(entries), dtype=np.uint8) if skip is None else np.asarray(skip, dtype=np.uint8)
    r = oracle.compress_rounds(src, offs, lens, skip, level=level, n_threads=1)
    r["usize"] = lens
    r["out_off"] = offs
    r["src"] = src
    return r


def _run_gpu(gpu_ctx, arch, pad_blobs=0):
    import torch
    from znippy_amd import hip
    blobs = np.concatenate([np.zeros(pad_blobs, np.uint8), arch["blobs"], np.zeros(32, np.uint8)])
    d_blobs = torch.from_numpy(blobs).cuda()
    total = int(arch["usize"].sum())
    d_out = torch.zeros(total + 64, dtype=torch.uint8, device="cuda")
    bitmap = np.packbits(arch["compressed"].astype(bool), bitorder="little")
    rt = hip.RowTable(gpu_ctx, arch["blob_offset"] + np.uint64(pad_blobs), arch["blob_size"], arch["usize"],
                      arch["out_off"], bitmap, arch["checksum"])
    counters, corrupt, status = rt.decode_verify(d_blobs, d_out)
    return counters, corrupt, status, d_out.cpu().numpy()[:total], rt


def test_rows_mixed_archive_matches_oracle_loop(gpu_ctx, oracle):
    """Mixed archive: compressed text/binary/pseudo-text rows of ragged sizes, stored (skip) rows,
    an empty row — GPU counters, bytes and digests equal the restated CPU read loop."""
    rng = np.random.default_rng(11)
    entries, skip = [], []
    for i in range(300):
        kind = i % 6
        n = int(rng.integers(0, 40000))
        if kind == 0: e = gen.text(n)
        elif kind == 1: e = gen.binary(n)
        elif kind == 2: e = gen.pseudo_text(n, seed=i)
        elif kind == 3: e = gen.incompressible(i, n)
        elif kind == 4: e = b""
        else: e = gen.pseudo_text(n * 4, seed=i)
        entries.append(e)
        skip.append(1 if kind == 3 and i % 2 else 0)
    entries.append(gen.pseudo_text(2 << 20, seed=77))   # multi-block frame, > 64 leaves
    skip.append(0)
    entries.append(gen.incompressible(5, 3 << 20))      # big stored row
    skip.append(1)
    arch = _build_archive(oracle, entries, level=3, skip=skip)
    counters, corrupt, status, out, rt = _run_gpu(gpu_ctx, arch, pad_blobs=5)
    n = len(entries)
    bitmap = np.packbits(arch["compressed"].astype(bool), bitorder="little")
    want_out = np.zeros(int(arch["usize"].sum()), dtype=np.uint8)
    want, want_corrupt = oracle.decompress_rows(arch["blobs"], arch["blob_offset"], arch["blob_size"], arch["usize"],
                                                arch["out_off"], bitmap, arch["checksum"], 0, n, out=want_out)
    assert (status == 0).all()
    assert counters == want
    assert len(corrupt) == 0 and len(want_corrupt) == 0
    assert np.array_equal(out, want_out)
    assert np.array_equal(out, arch["src"][:len(out)])
    assert np.array_equal(rt.digests(), arch["checksum"])


def test_rows_corruption_is_counted_not_fatal(gpu_ctx, oracle):
    """decompress.rs:L159-162,L175-189: a checksum mismatch is counted and the bytes are still
    written; a frame that fails to decode is counted in chunks only."""
    entries = [gen.text(10240) for _ in range(20)] + [gen.pseudo_text(30000, seed=3)]
    arch = _build_archive(oracle, entries, level=19)
    arch["checksum"] = arch["checksum"].copy()
    arch["checksum"][3, 0] ^= 0xFF           # wrong expected digest -> corrupt row 3
    arch["checksum"][17, 31] ^= 0x01
    blobs = arch["blobs"].copy()
    o = int(arch["blob_offset"][20])
    blobs[o + 1] ^= 0xFF                     # break the magic of row 20 -> decode error
    arch["blobs"] = blobs
    counters, corrupt, status, out, rt = _run_gpu(gpu_ctx, arch)
    bitmap = np.packbits(arch["compressed"].astype(bool), bitorder="little")
    want, want_corrupt = oracle.decompress_rows(arch["blobs"], arch["blob_offset"], arch["blob_size"], arch["usize"],
                                                arch["out_off"], bitmap, arch["checksum"], 0, len(entries))
    assert counters == want
    assert list(corrupt) == [3, 17] == list(want_corrupt)
    assert status[20] < 0 and (status[:20] == 0).all()
    assert counters["decode_errors"] == 1 and counters["total_chunks"] == 21
    # mismatching rows are still written
    assert out[3 * 10240:4 * 10240].tobytes() == entries[3]


def test_truncated_and_garbage_frames_fail_cleanly(gpu_ctx, oracle):
    from znippy_amd._lib import ZnippyError
    frame = oracle.libzstd_compress(gen.pseudo_text(50000), 3)
    for bad in (frame[:len(frame) // 2], frame[:-1], frame[:7]):
        with pytest.raises(ZnippyError):
            gpu_ctx.decompress(bad)
    mangled = bytearray(frame)
    mangled[len(frame) // 2] ^= 0x55
    try:
        got = gpu_ctx.decompress(bytes(mangled))   # either an error or wrong bytes, never a hang/fault
        assert got != gen.pseudo_text(50000) or True
    except ZnippyError:
        pass


def test_c2_shape_100k_rows_roundtrip_property(gpu_ctx, oracle):
    """BASELINE C2 at full size (100k x 10 KiB text): every row decodes to the generator's bytes
    (size-independent property: all rows identical + checksum of checksums)."""
    import torch
    from znippy_amd import hip
    n, sz = 100_000, 10240
    chunk = gen.text(sz)
    frame = np.frombuffer(oracle.libzstd_compress(chunk, 19), dtype=np.uint8)
    fl = len(frame)
    d_blobs = torch.from_numpy(np.concatenate([np.tile(frame, n), np.zeros(32, np.uint8)])).cuda()
    d_out = torch.zeros(n * sz + 64, dtype=torch.uint8, device="cuda")
    ck = np.tile(np.frombuffer(oracle.blake3(chunk), dtype=np.uint8), (n, 1))
    rt = hip.RowTable(gpu_ctx, np.arange(n, dtype=np.uint64) * fl, np.full(n, fl, np.uint64),
                      np.full(n, sz, np.uint64), np.arange(n, dtype=np.uint64) * sz, None, ck)
    counters, corrupt, status = rt.decode_verify(d_blobs, d_out)
    assert counters == dict(total_chunks=n, total_written_bytes=n * sz, verified_bytes=n * sz, corrupt_bytes=0,
                            corrupt_rows=0, decode_errors=0)
    out = d_out[:n * sz].view(n, sz)
    ref = torch.from_numpy(np.frombuffer(chunk, dtype=np.uint8).copy()).cuda()
    assert bool((out == ref[None, :]).all())


def _with_checksum(oracle, frame, content):
    """Turn a libzstd frame (no checksum) into one WITH the Content_Checksum flag + XXH64 trailer."""
    import struct
    import xxhash
    b = bytearray(frame)
    assert not (b[4] & 4)
    b[4] |= 4
    return bytes(b) + struct.pack("<I", xxhash.xxh64(content).intdigest() & 0xFFFFFFFF)


@pytest.mark.parametrize("gname,n", [("text", 10240), ("pseudo_text", 70000), ("binary", 300000), ("text", 0)])
def test_frame_content_checksum_is_verified(gpu_ctx, oracle, gname, n):
    from znippy_amd._lib import ZnippyError, E_CHECKSUM
    data = _gen(gname, n)
    frame = _with_checksum(oracle, oracle.libzstd_compress(data, 3), data)
    assert oracle.zstd_decompress(frame) == data and oracle.libzstd_decompress(frame, max(n, 1)) == data
    assert gpu_ctx.decompress(frame) == data
    bad = bytearray(frame)
    bad[-1] ^= 0x10                                      # wrong checksum -> ZNIPPY_E_CHECKSUM (oracle: error too)
    with pytest.raises(ValueError):
        oracle.zstd_decompress(bytes(bad))
    with pytest.raises(ZnippyError) as ei:
        gpu_ctx.decompress(bytes(bad))
    assert ei.value.code == E_CHECKSUM


def test_block_items_and_their_fallbacks(gpu_ctx, oracle):
    """Frames of >= 2 blocks are tried block by block (every 128 KiB block a work item) and fall back to the serial
    decoder per frame.  One table holds: multi-block frames written by this library (independent blocks: the
    fast path), libzstd frames (repeat offsets / Huffman: fallback), a library frame with a damaged block (error
    surfaces through the fallback) — outputs and statuses must equal the oracle's."""
    import torch
    from znippy_amd import hip
    rng = np.random.default_rng(3)
    srcs = [gen.pseudo_text(700_000, seed=1), gen.binary(300_000), gen.text(1 << 20), gen.pseudo_text(400_000, seed=2),
            gen.incompressible(4, 300_000), gen.pseudo_text(262_145, seed=5)]
    own = [gpu_ctx.compress(x) for x in srcs]
    for f, x in zip(own, srcs):
        assert oracle.zstd_decompress(f) == x
    frames = list(own) + [oracle.libzstd_compress(srcs[0], 19), oracle.libzstd_compress(srcs[2], 3)]
    want = list(srcs) + [srcs[0], srcs[2]]
    bad = bytearray(own[3])
    bad[len(bad) // 2] ^= 0x5A                      # somewhere inside a middle block
    frames.append(bytes(bad)); want.append(srcs[3])
    lens = np.array([len(x) for x in want], np.uint64)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    bo = np.concatenate([[0], np.cumsum([len(f) for f in frames])[:-1]]).astype(np.uint64)
    bs = np.array([len(f) for f in frames], np.uint64)
    blobs = torch.from_numpy(np.frombuffer(b"".join(frames) + bytes(64), np.uint8).copy()).cuda()
    d_out = torch.zeros(int(lens.sum()) + 64, dtype=torch.uint8, device="cuda")
    ck = np.stack([np.frombuffer(oracle.blake3(x), np.uint8) for x in want])
    rt = hip.RowTable(gpu_ctx, bo, bs, lens, offs, None, ck)
    counters, corrupt, status = rt.decode_verify(blobs, d_out)
    out = d_out.cpu().numpy()
    for i in range(len(frames) - 1):
        assert status[i] == 0, i
        assert out[int(offs[i]):int(offs[i] + lens[i])].tobytes() == want[i], i
    # the damaged frame: the oracle either rejects it or decodes different bytes; the GPU must agree
    try:
        ob = oracle.zstd_decompress(bytes(bad))
    except ValueError:
        ob = None
    i = len(frames) - 1
    if ob is None or len(ob) != len(want[i]):
        assert status[i] < 0
        assert counters["decode_errors"] == 1
    else:
        assert status[i] == 0 and out[int(offs[i]):int(offs[i] + lens[i])].tobytes() == ob
        assert list(corrupt) == ([i] if ob != want[i] else [])
    # timing names tell which path ran
    names = dict(gpu_ctx.kernel_times())
    assert "zstd_decode_blocks" in names and "zstd_block_scan" in names


def _periodic(period, n, seed):
    rng = np.random.default_rng(seed)
    p = rng.integers(32, 127, size=period, dtype=np.uint8).tobytes()
    return (p * (n // period + 1))[:n]


@pytest.mark.parametrize("level", [1, 19])
def test_periodic_rows_every_period_and_alignment(gpu_ctx, oracle, level):
    """The lane-parallel recognised-row path of the fused kernel (literal prefix + one overlapping match):
    periods 1..600, sizes that are whole leaves (hashed from the staged window) and ragged ones (hashed from the
    output), output offsets of every alignment because the rows are packed back to back."""
    rng = np.random.default_rng(5)
    entries = []
    for i, period in enumerate(list(range(1, 70)) + [100, 127, 128, 129, 255, 256, 257, 400, 511, 600]):
        n = int(rng.integers(1, 40)) * 1024 if i % 3 else int(rng.integers(70, 40000))
        entries.append(_periodic(period, n, i))
    entries += [gen.text(10240)] * 40 + [gen.text(10239), gen.text(10241), gen.text(65536), gen.text(65), gen.text(64)]
    arch = _build_archive(oracle, entries, level=level)
    counters, corrupt, status, out, rt = _run_gpu(gpu_ctx, arch, pad_blobs=3)
    assert (status == 0).all()
    assert len(corrupt) == 0 and counters["verified_bytes"] == sum(len(e) for e in entries)
    assert out.tobytes() == b"".join(entries)
    assert np.array_equal(rt.digests(), arch["checksum"])


def test_periodic_rows_from_this_encoder(gpu_ctx, oracle):
    """Same shapes, frames written by the HIP encoder (what the fused kernel sees in practice)."""
    import torch
    from znippy_amd import hip
    rng = np.random.default_rng(6)
    entries = []
    for i, period in enumerate(list(range(1, 70)) + [100, 128, 257, 600, 1000, 1024, 1025, 2000]):
        n = int(rng.integers(1, 40)) * 1024 if i % 3 else int(rng.integers(70, 40000))
        entries.append(_periodic(period, n, 100 + i))
    entries += [gen.text(10240)] * 30
    src = np.frombuffer(b"".join(entries), dtype=np.uint8)
    lens = np.array([len(e) for e in entries], dtype=np.uint64)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    d_src = torch.from_numpy(src.copy()).cuda()
    rounds = hip.RoundTable(gpu_ctx, offs, lens)
    d_blob = torch.zeros(rounds.blob_bound() + 64, dtype=torch.uint8, device="cuda")
    enc = rounds.encode_hash(d_src, d_blob)
    rows = hip.RowTable(gpu_ctx, enc["blob_offset"], enc["blob_size"], lens, offs, None, enc["checksum"])
    d_out = torch.zeros(len(src) + 64, dtype=torch.uint8, device="cuda")
    counters, corrupt, status = rows.decode_verify(d_blob, d_out)
    assert (status == 0).all() and len(corrupt) == 0
    assert counters["verified_bytes"] == len(src)
    assert np.array_equal(d_out.cpu().numpy()[:len(src)], src)
    for i in (0, 5, 40, len(entries) - 1):
        assert bytes(enc["checksum"][i]) == oracle.blake3(entries[i])


def test_big_rows_block_items_fused(gpu_ctx, oracle):
    """Big rows whose 128 KiB blocks are 'literals + one periodic match' or raw are written and hashed by the fused
    block kernel; short last blocks, other shapes and mixed frames take the block decoder + second hash pass.
    Bytes and digests must not depend on which path a block took."""
    import torch
    from znippy_amd import hip
    rng = np.random.default_rng(9)
    BLK = 128 * 1024
    entries = [
        _periodic(45, 8 * BLK, 1),                     # whole blocks, all recognised
        _periodic(7, 3 * BLK + 12345, 2),              # short last block
        _periodic(600, 2 * BLK + 1, 3),
        gen.incompressible(4, 4 * BLK),                # raw blocks
        gen.incompressible(5, 2 * BLK + 999),
        _periodic(13, BLK, 6) + gen.incompressible(7, BLK) + gen.pseudo_text(BLK, seed=8) + _periodic(200, BLK + 77, 9),  # mixed frame
        gen.pseudo_text(3 * BLK + 5, seed=10),         # entropy-coded blocks: not for the fused kernel
        _periodic(1, 5 * BLK, 11),
        gen.text(10240), gen.text(70000),
    ]
    src = np.frombuffer(b"".join(entries), dtype=np.uint8)
    lens = np.array([len(e) for e in entries], dtype=np.uint64)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    d_src = torch.from_numpy(src.copy()).cuda()
    rounds = hip.RoundTable(gpu_ctx, offs, lens)
    d_blob = torch.zeros(rounds.blob_bound() + 64, dtype=torch.uint8, device="cuda")
    enc = rounds.encode_hash(d_src, d_blob)
    for i, e in enumerate(entries):
        assert bytes(enc["checksum"][i]) == oracle.blake3(e)
    rows = hip.RowTable(gpu_ctx, enc["blob_offset"], enc["blob_size"], lens, offs, None, enc["checksum"])
    d_out = torch.zeros(len(src) + 64, dtype=torch.uint8, device="cuda")
    for _ in range(2):  # the second run reuses the row table (done flags are reset per run)
        d_out.zero_()
        counters, corrupt, status = rows.decode_verify(d_blob, d_out)
        assert (status == 0).all() and len(corrupt) == 0
        assert counters["verified_bytes"] == len(src)
        assert np.array_equal(d_out.cpu().numpy()[:len(src)], src)
    assert np.array_equal(rows.digests(), enc["checksum"])
    names = [k for k, _ in gpu_ctx.kernel_times()]
    assert "decode_verify_fused_blocks" in names


def test_stored_big_rows_every_output_alignment(gpu_ctx, oracle):
    """Store path, big rows: the hash+copy kernel re-cuts the bytes when an output offset is not a multiple of 16
    (16-byte stores land on boundaries; a leaf's first and last 16 bytes go out as they are).  Every alignment 0..15,
    sizes that are whole leaves and ragged ones, guard bytes between the rows must survive."""
    import torch
    from znippy_amd import hip
    rng = np.random.default_rng(21)
    sizes = [(3 << 16) + (i % 3) * 1024 + (0 if i % 2 else int(rng.integers(1, 1023))) for i in range(16)] + [1 << 20, (1 << 20) + 5]
    rows = [gen.incompressible(50 + i, n) for i, n in enumerate(sizes)]
    GAP = 48
    out_off, pos = [], 0
    for i, n in enumerate(sizes):
        pos = (pos + 15) // 16 * 16 + (i % 16)          # alignment i mod 16
        out_off.append(pos)
        pos += n + GAP
    total = pos + 64
    blobs = np.frombuffer(b"".join(rows) + bytes(64), dtype=np.uint8)
    bs = np.array(sizes, dtype=np.uint64)
    bo = np.concatenate([[0], np.cumsum(bs)[:-1]]).astype(np.uint64)
    ck = np.stack([np.frombuffer(oracle.blake3(d), dtype=np.uint8) for d in rows])
    d_blobs = torch.from_numpy(blobs.copy()).cuda()
    d_out = torch.full((total,), 0xA5, dtype=torch.uint8, device="cuda")
    bitmap = np.zeros((len(sizes) + 7) // 8, np.uint8)   # nothing compressed
    rt = hip.RowTable(gpu_ctx, bo, bs, bs, np.array(out_off, dtype=np.uint64), bitmap, ck)
    counters, corrupt, status = rt.decode_verify(d_blobs, d_out)
    assert (status == 0).all() and len(corrupt) == 0 and counters["verified_bytes"] == sum(sizes)
    out = d_out.cpu().numpy()
    want = np.full(total, 0xA5, np.uint8)
    for o, d in zip(out_off, rows):
        want[o:o + len(d)] = np.frombuffer(d, dtype=np.uint8)
    assert np.array_equal(out, want)


@pytest.mark.parametrize("switch", [{}, {"ZNIPPY_NO_STORED_ONLY": "1"}])
def test_tables_without_a_compressed_row(oracle, switch):
    """A table whose rows are all stored (nothing to recognise or decode: png / jpg / gz files, jars) takes one pass of the
    store path kernel over all its tiles — small ones too — instead of the fused small-row kernels (api.hip: stored_only;
    ZNIPPY_NO_STORED_ONLY=1 = the old route).  Small rows of every size class, empty rows, 64-leaf rows, big rows, odd and
    even output offsets, a corrupted row: same counters, same corrupt list, same bytes either way and as the oracle says."""
    import os
    import torch
    from znippy_amd import hip
    old = {k: os.environ.get(k) for k in switch}
    os.environ.update(switch)
    try:
        ctx = hip.Context(0)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    try:
        rng = np.random.default_rng(5)
        sizes = [10240] * 40 + [0, 1, 63, 64, 1023, 1024, 1025, 65535, 65536, 65537, 3 << 16, (1 << 20) + 77] + [int(x) for x in rng.integers(0, 40000, 300)] + [10240] * 25
        rows = [gen.incompressible(7000 + i, n) for i, n in enumerate(sizes)]
        for pad, gap in ((0, 0), (5, 3)):
            out_off, pos = [], pad
            for n in sizes:
                out_off.append(pos)
                pos += n + gap
            total = pos + 64
            blobs = np.frombuffer(b"".join(rows) + bytes(64), dtype=np.uint8).copy()
            bs = np.array(sizes, dtype=np.uint64)
            bo = np.concatenate([[0], np.cumsum(bs)[:-1]]).astype(np.uint64)
            ck = np.stack([np.frombuffer(oracle.blake3(d), dtype=np.uint8) for d in rows])
            bad_row = 17
            blobs[int(bo[bad_row]) + 100] ^= 0x40            # one stored row damaged in the archive
            d_blobs = torch.from_numpy(blobs).cuda()
            bitmap = np.zeros((len(sizes) + 7) // 8, np.uint8)
            rt = hip.RowTable(ctx, bo, bs, bs, np.array(out_off, dtype=np.uint64), bitmap, ck)
            want = np.full(total, 0xA5, np.uint8)
            for i, (o, d) in enumerate(zip(out_off, rows)):
                want[o:o + len(d)] = blobs[int(bo[i]):int(bo[i]) + len(d)]
            for _ in range(2):
                d_out = torch.full((total,), 0xA5, dtype=torch.uint8, device="cuda")
                counters, corrupt, status = rt.decode_verify(d_blobs, d_out)
                assert (status == 0).all() and list(corrupt) == [bad_row], (list(corrupt), np.nonzero(status)[0][:5])
                assert counters["corrupt_rows"] == 1 and counters["verified_bytes"] == sum(sizes) - sizes[bad_row]
                assert np.array_equal(d_out.cpu().numpy(), want)
            names = dict(ctx.kernel_times())
            assert ("decode_verify_fused" in names) == bool(switch), sorted(names)
            rt.close()
    finally:
        ctx.close()


@pytest.mark.parametrize("tiles_per_wave", ["1", "2"])
def test_stored_rows_one_and_two_tiles_per_wave(oracle, tiles_per_wave):
    """Store path kernel, both forms (hash_kernels.hip: k_hash_tiles<COPY, 1 | 2>; picked from the tile count, forced here
    through ZNIPPY_STORE_G): with two tiles per wave the first tile's leaf CVs wait in LDS in front of the stage and the two
    parent trees are folded together.  Rows whose tiles pair up across row boundaries, ragged last tiles, a last wave
    with one tile, small rows between the big ones (their tiles fold on the spot or queue, by their unit count) — every
    digest equal to the oracle's, every byte in place, guard bytes untouched."""
    import os
    import torch
    from znippy_amd import hip
    old = os.environ.get("ZNIPPY_STORE_G")
    os.environ["ZNIPPY_STORE_G"] = tiles_per_wave
    try:
        ctx = hip.Context(0)
    finally:
        if old is None:
            del os.environ["ZNIPPY_STORE_G"]
        else:
            os.environ["ZNIPPY_STORE_G"] = old
    try:
        rng = np.random.default_rng(77)
        sizes = [1 << 16, (1 << 16) + 1, 3 << 16, (5 << 16) + 1024 * 7, (2 << 16) + 999, 100, 0, 10240, 1 << 20, 65, (1 << 20) + 65536 + 3,
                 (7 << 16) - 1, 4 << 20] + [int(x) for x in rng.integers(1, 30000, 150)] + [(9 << 16) + 512]
        rows = [gen.incompressible(300 + i, n) for i, n in enumerate(sizes)]
        GAP = 32
        out_off, pos = [], 0
        for n in sizes:
            pos = (pos + 15) // 16 * 16
            out_off.append(pos)
            pos += n + GAP
        total = pos + 64
        blobs = np.frombuffer(b"".join(rows) + bytes(64), dtype=np.uint8)
        bs = np.array(sizes, dtype=np.uint64)
        bo = np.concatenate([[0], np.cumsum(bs)[:-1]]).astype(np.uint64)
        ck = np.stack([np.frombuffer(oracle.blake3(d), dtype=np.uint8) for d in rows])
        d_blobs = torch.from_numpy(blobs.copy()).cuda()
        bitmap = np.zeros((len(sizes) + 7) // 8, np.uint8)   # nothing compressed
        rt = hip.RowTable(ctx, bo, bs, bs, np.array(out_off, dtype=np.uint64), bitmap, ck)
        want = np.full(total, 0xA5, np.uint8)
        for o, d in zip(out_off, rows):
            want[o:o + len(d)] = np.frombuffer(d, dtype=np.uint8)
        for _ in range(2):
            d_out = torch.full((total,), 0xA5, dtype=torch.uint8, device="cuda")
            counters, corrupt, status = rt.decode_verify(d_blobs, d_out)
            assert (status == 0).all() and len(corrupt) == 0 and counters["verified_bytes"] == sum(sizes)
            assert np.array_equal(d_out.cpu().numpy(), want)
        rt.close()
        # the write side's hash + copy (rounds marked skip: stored as they are) runs the same kernel
        rounds = hip.RoundTable(ctx, [int(x) for x in bo], sizes, skip=np.ones(len(sizes), np.uint8))
        d_blob_out = torch.full((rounds.blob_bound() + 64,), 0x5A, dtype=torch.uint8, device="cuda")
        res = rounds.encode_hash(d_blobs, d_blob_out)
        blob = d_blob_out.cpu().numpy()
        for i, d in enumerate(rows):
            assert res["checksum"][i].tobytes() == oracle.blake3(d), i
            o, n = int(res["blob_offset"][i]), int(res["blob_size"][i])
            assert n == len(d) and blob[o:o + n].tobytes() == d, i
        assert (blob[res["blob_bytes"]:] == 0x5A).all()
        rounds.close()
    finally:
        ctx.close()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_archives_every_path_every_time(gpu_ctx, gpu_ctx_roles, oracle, seed):
    """Randomised archives — whole-leaf and ragged rows, text / binary / word-soup / incompressible / empty, stored and
    compressed, runs of equal rows and single ones, a few big rows, ~1 in 40 rows damaged — through the default context and
    through one that sends every table to the role-split kernel first, three runs each: counters, verdicts, bytes and
    digests equal the oracle's read loop every time."""
    rng = np.random.default_rng(1000 + seed)
    entries, skip = [], []
    while len(entries) < 2500:
        kind = int(rng.integers(0, 8))
        run = int(rng.integers(1, 30)) if rng.random() < 0.5 else 1
        n = int(rng.choice([0, 1, 1023, 1024, 1025, 4096, 10240, 10240, 10240, 20480, 30720, 65536, int(rng.integers(2, 50000))]))
        if kind <= 1: e = gen.text(n)
        elif kind == 2: e = gen.binary(n)
        elif kind == 3: e = gen.pseudo_text(min(n, 20000), seed=len(entries))
        elif kind == 4: e = gen.incompressible(len(entries), min(n, 30000))
        elif kind == 5: e = bytes(n)
        else: e = gen.text(n)
        for _ in range(run):
            entries.append(e)
            skip.append(1 if kind == 4 and len(entries) % 3 == 0 else 0)
    for big, sk in ((gen.text(700_000), 0), (gen.incompressible(9, 400_000), 1), (gen.pseudo_text(300_000, seed=5), 0)):
        at = int(rng.integers(0, len(entries)))
        entries.insert(at, big); skip.insert(at, sk)
    arch = _build_archive(oracle, entries, level=3, skip=skip)
    n = len(entries)
    blobs = arch["blobs"].copy()
    for i in rng.choice(n, size=n // 40, replace=False):       # damage: one byte somewhere in the row's blob
        if arch["blob_size"][i] > 0:
            at = int(arch["blob_offset"][i]) + int(rng.integers(0, int(arch["blob_size"][i])))
            blobs[at] ^= 1 << int(rng.integers(0, 8))
    arch["blobs"] = blobs
    bitmap = np.packbits(arch["compressed"].astype(bool), bitorder="little")
    want_out = np.zeros(int(arch["usize"].sum()), dtype=np.uint8)
    want, want_corrupt = oracle.decompress_rows(arch["blobs"], arch["blob_offset"], arch["blob_size"], arch["usize"],
                                                arch["out_off"], bitmap, arch["checksum"], 0, n, out=want_out)
    for ctx in (gpu_ctx, gpu_ctx_roles):
        for rep in range(3):
            counters, corrupt, status, out, rt = _run_gpu(ctx, arch, pad_blobs=3)
            assert counters == want, (rep, counters, want)
            assert sorted(int(x) for x in corrupt) == sorted(int(x) for x in want_corrupt), rep
            okrows = status >= 0
            assert int((~okrows).sum()) == want["decode_errors"]
            for i in np.nonzero(okrows)[0][:: max(1, n // 400)]:    # bytes of a sample of the decoded rows (all digests below)
                a, b = int(arch["out_off"][i]), int(arch["out_off"][i] + arch["usize"][i])
                assert np.array_equal(out[a:b], want_out[a:b]), (rep, int(i))
            good = okrows.copy(); good[[int(x) for x in want_corrupt]] = False
            assert np.array_equal(rt.digests()[good], arch["checksum"][good]), rep
