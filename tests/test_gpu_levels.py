"""GPU tests for the encoder's effort tiers: CompressCtx::new(compression_level) (znippy-common/src/codec.rs:L16-28,
CONFIG.compression_level = 19, common_config.rs:L37) -> znippy_ctx_set_level.

Levels 1-3 are the fast tier, 4-22 the higher effort one.  Both write plain RFC 8878 frames: every frame here is decoded
by libzstd, by the oracle restatement and by the GPU decoder.  What the higher tier adds is checked on the bytes it
writes (per-block entropy tables, Huffman literals for byte alphabets, repeat offsets, the closing mark of multi-block
frames) and on how its frames come back through the read path (block items, not the foreign-frame path).
"""
import numpy as np
import pytest

import gen

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctxs():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from znippy_amd import hip
    fast, high = hip.Context(0), hip.Context(0)
    fast.set_level(1)
    high.set_level(19)
    yield fast, high
    fast.close()
    high.close()


def _blocks(frame):
    """[(type, size, last, payload offset)] of a single-segment frame as this encoder writes it."""
    fhd = frame[4]
    assert (fhd >> 5) & 1
    p = 5 + {0: 1, 1: 2, 2: 4, 3: 8}[fhd >> 6]
    out = []
    while True:
        bh = frame[p] | (frame[p + 1] << 8) | (frame[p + 2] << 16)
        typ, size, last = (bh >> 1) & 3, bh >> 3, bh & 1
        out.append((typ, size, last, p + 3))
        p += 3 + (1 if typ == 1 else size)
        if last:
            assert p == len(frame)
            return out


def _sections(frame, off):
    """(literals type, Compression_Modes byte or None) of the compressed block whose payload starts at off."""
    b0 = frame[off]
    lt, sf = b0 & 3, (b0 >> 2) & 3
    h = int.from_bytes(frame[off:off + 5], "little")
    if lt <= 1:
        lh = 1 if (sf & 1) == 0 else (2 if sf == 1 else 3)
        regen = (h & 0xFF) >> 3 if (sf & 1) == 0 else ((h & 0xFFFF) >> 4 if sf == 1 else (h & 0xFFFFFF) >> 4)
        q = off + lh + (regen if lt == 0 else 1)
    else:
        lh, nb = (3, 10) if sf <= 1 else ((4, 14) if sf == 2 else (5, 18))
        q = off + lh + ((h >> (4 + nb)) & ((1 << nb) - 1))
    s0 = frame[q]
    hl = 0 if s0 == 0 else (1 if s0 < 128 else (2 if s0 < 255 else 3))
    return lt, (frame[q + hl] if hl else None), off + (lh if lt >= 2 else 0)


def _three_decoders(oracle, ctx, frame, data):
    assert oracle.libzstd_decompress(frame, max(len(data), 1)) == data
    assert oracle.zstd_decompress(frame) == data
    assert ctx.decompress(frame) == data


def _byte_skew(n, seed, symbols=200):
    """Bytes over an alphabet of `symbols` values reaching beyond 127, geometric-ish frequencies, no repeats to match."""
    rng = np.random.default_rng(seed)
    w = 1.0 / np.arange(1, symbols + 1) ** 1.2
    vals = (np.arange(symbols) * 37 + 11) % 256
    assert len(set(vals.tolist())) == symbols
    return vals.astype(np.uint8)[rng.choice(symbols, size=n, p=w / w.sum())].tobytes()


def _records(n_rec, seed):
    """6 fresh bytes + 10 bytes copied from 1024 bytes back, over and over (the fresh bytes are kept from extending the
    copy at either end): one match length for the whole block, and behind the first sequence one repeated offset."""
    rng = np.random.default_rng(seed)
    buf = bytearray(rng.integers(0, 256, 1024, dtype=np.uint8).tobytes())
    for _ in range(n_rec):
        fresh = bytearray(rng.integers(0, 256, 6, dtype=np.uint8).tobytes())
        src = len(buf) + 6 - 1024
        if len(buf) >= 1040 and fresh[0] == buf[len(buf) - 1024]:
            fresh[0] ^= 1        # would lengthen the previous record's copy
        if fresh[5] == buf[src - 1]:
            fresh[5] ^= 1        # would be taken over by this record's copy
        buf += fresh
        buf += buf[src:src + 10]
    return bytes(buf)


def _rle_records(n_rec, seed):
    """Records of 100 fresh bytes + the block's first 8 bytes again: every sequence has literal length 100 and match
    length 8 — one code each, which the sequences section writes in RLE mode.  The first and last fresh byte of a record
    differ from those of the ~250 records around it (more than the match finder remembers), so that no candidate copy
    is a byte longer at either end."""
    rng = np.random.default_rng(seed)
    buf = bytearray()
    for r in range(n_rec):
        fresh = bytearray(rng.integers(0, 256, 100, dtype=np.uint8).tobytes())
        fresh[0], fresh[99] = r % 251, (r * 7 + 3) % 253
        buf += fresh
        buf += buf[0:8]
    return bytes(buf)


def _varied(n_rec, seed, max_fresh=40, max_copy=60, alphabet=256):
    """Records with literal runs of 0..max_fresh bytes and copies of 4..max_copy bytes from random earlier places: many
    literal-length, match-length and offset codes with small counts each — what the normalisation of the per-block
    tables has to squeeze into 64..256 cells — over a byte alphabet of the given size."""
    rng = np.random.default_rng(seed)
    buf = bytearray(rng.integers(0, alphabet, 64, dtype=np.uint8).tobytes())
    for _ in range(n_rec):
        buf += rng.integers(0, alphabet, int(rng.integers(0, max_fresh + 1)), dtype=np.uint8).tobytes()
        ln = int(rng.integers(4, max_copy + 1))
        src = int(rng.integers(0, max(1, len(buf) - ln)))
        buf += buf[src:src + ln]
    return bytes(buf)


@pytest.mark.parametrize("seed", range(12))
def test_many_codes_with_small_counts(ctxs, oracle, seed):
    """Table normalisation under pressure (more symbols than a 64-cell table has room to round generously), Huffman
    trees of every alphabet size, both on the higher tier; three decoders judge every frame."""
    fast, high = ctxs
    rng = np.random.default_rng(1000 + seed)
    n_rec = int(rng.choice([130, 200, 400, 700, 1500, 4000, 12000]))
    alphabet = int(rng.choice([2, 5, 17, 64, 127, 129, 200, 256]))
    data = _varied(n_rec, seed, max_fresh=int(rng.choice([3, 20, 40, 90])), max_copy=int(rng.choice([8, 30, 60, 200])), alphabet=alphabet)
    for ctx in (high, fast):
        frame = ctx.compress(data)
        _three_decoders(oracle, ctx, frame, data)
    assert len(high.compress(data)) <= len(fast.compress(data)) + 16


def test_level_is_part_of_the_context(ctxs):
    from znippy_amd import hip
    from znippy_amd._lib import ZnippyError
    ctx = hip.Context(0)
    try:
        assert ctx.level == 19                      # CONFIG.compression_level, common_config.rs:L37
        for bad in (0, -1, 23, 100):
            with pytest.raises(ZnippyError):
                ctx.set_level(bad)
        assert ctx.level == 19
        for lv in (1, 3, 4, 22):
            ctx.set_level(lv)
            assert ctx.level == lv
        data = gen.pseudo_text(200_000, seed=5)
        ctx.set_level(3)
        f3 = ctx.compress(data)
        ctx.set_level(4)
        f4 = ctx.compress(data)
        assert f3 == ctxs[0].compress(data) and f4 == ctxs[1].compress(data)   # two tiers: 1-3 and 4-22
        assert len(f4) < len(f3)
    finally:
        ctx.close()


def test_codec_mirror_applies_its_level(ctxs, oracle):
    from znippy_amd import codec
    fast, high = ctxs
    data = gen.pseudo_text(150_000, seed=9)
    want1, want19 = fast.compress(data), high.compress(data)
    shared = high
    a, b = codec.CompressCtx(1, ctx=shared), codec.CompressCtx(19, ctx=shared)   # one HIP context, two CompressCtx
    fa, fb, fa2 = a.compress(data), b.compress(data), a.compress(data)
    shared.set_level(19)                                                         # (the fixture's level again)
    assert fa == fa2 == want1 and fb == want19 and len(fb) < len(fa)
    with pytest.raises(codec.ZnippyError):
        codec.CompressCtx(0, ctx=shared).compress(data)
    assert shared.level == 19
    _three_decoders(oracle, shared, fb, data)


@pytest.mark.parametrize("name,n", [("pseudo_text", 300_000), ("binary_skew", 200_000), ("records", 6000), ("rle_records", 400), ("text", 1 << 20),
                                    ("pseudo_text", 131_072 + 3000), ("pseudo_text", 2 * 131_072)])
def test_what_the_higher_tier_writes(ctxs, oracle, name, n):
    fast, high = ctxs
    data = {"binary_skew": lambda: _byte_skew(n, 3), "records": lambda: _records(n, 4), "rle_records": lambda: _rle_records(n, 6)}.get(name, lambda: getattr(gen, name)(n))()
    f1, f19 = fast.compress(data), high.compress(data)
    _three_decoders(oracle, fast, f1, data)
    _three_decoders(oracle, high, f19, data)
    _three_decoders(oracle, fast, f19, data)            # the level belongs to the encoder: any context reads any frame
    assert len(f19) <= len(f1)
    b1, b19 = _blocks(f1), _blocks(f19)
    n_real = (len(data) + 131071) // 131072
    # fast tier: one block per 128 KiB, the last one flagged; sequences on the predefined tables
    assert len(b1) == n_real and all(_sections(f1, off)[1] in (None, 0) for typ, _, _, off in b1 if typ == 2)
    if n_real > 1:   # the closing mark: an empty raw block behind the last real one
        assert len(b19) == n_real + 1 and b19[-1][:3] == (0, 0, 1) and all(b[2] == 0 for b in b19[:-1])
    else:
        assert len(b19) == 1
    if name in ("pseudo_text", "binary_skew", "records"):
        assert b19[0][0] == 2
    lt, modes, lit_off = _sections(f19, b19[0][3]) if b19[0][0] == 2 else (None, None, None)
    if name == "pseudo_text":
        assert lt == 2 and modes is not None and modes != 0 and len(f19) < 0.9 * len(f1)   # Huffman literals, per-block tables
    if name == "binary_skew":
        assert (b1[0][0] != 2 or _sections(f1, b1[0][3])[0] == 0) and lt == 2   # raw at level 1 (alphabet beyond 128), Huffman at 19
        assert f19[lit_off] < 128                                  # ... described by FSE-compressed weights
        assert len(f19) < 0.95 * len(f1)
    if name == "records":
        assert modes is not None and modes != 0 and len(f19) < 0.8 * len(f1)   # offsets go out as repeat codes
    if name == "rle_records":
        assert b19[0][0] == 2 and modes is not None and (modes >> 6) == 1 and ((modes >> 2) & 3) == 1, modes   # LL and ML: RLE_Mode


def test_own_frames_of_both_tiers_come_back_as_block_items(ctxs, oracle):
    """Multi-block frames of either tier are decoded block by block (zstd_decode_blocks), not by the path for
    foreign frames: the higher tier's blocks carry their own entropy tables and repeat codes, and still stand alone."""
    import torch
    from znippy_amd import hip
    ents = [gen.pseudo_text(300_000, seed=1), gen.binary(1 << 20), gen.pseudo_text(131_072 + 5000, seed=2), _byte_skew(400_000, 5),
            _records(20_000, 6), gen.text(700_000), gen.pseudo_text(131_072 * 3, seed=3)]
    lens = np.array([len(e) for e in ents], np.uint64)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    total = int(lens.sum())
    d_src = torch.from_numpy(np.frombuffer(b"".join(ents) + bytes(64), np.uint8).copy()).cuda()
    sizes = {}
    for label, ctx in zip(("fast", "high"), ctxs):
        rt = hip.RoundTable(ctx, offs, lens)
        d_blob = torch.zeros(rt.blob_bound() + 64, dtype=torch.uint8, device="cuda")
        enc = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in rt.encode_hash(d_src, d_blob).items()}
        sizes[label] = int(enc["blob_size"].sum())
        for i, e in enumerate(ents):
            assert bytes(enc["checksum"][i]) == oracle.blake3(e)
        rows = hip.RowTable(ctx, enc["blob_offset"], enc["blob_size"], lens, offs, None, enc["checksum"])
        d_out = torch.zeros(total + 64, dtype=torch.uint8, device="cuda")
        for _ in range(2):
            d_out.zero_()
            c, corrupt, status = rows.decode_verify(d_blob, d_out)
            assert c["corrupt_rows"] == 0 and c["decode_errors"] == 0 and c["verified_bytes"] == total
            assert bool((d_out[:total] == d_src[:total]).all())
            assert rows.foreign_stats()["frames"] == 0, (label, rows.foreign_stats())
            kt = dict(ctx.kernel_times())
            assert kt.get("zstd_decode_blocks", 0) > 0.05 and kt.get("zstd_decode_fallback", 0) < 0.05, kt
        rows.close()
        rt.close()
    assert sizes["high"] < sizes["fast"]


def test_repeat_codes_that_need_the_block_before_leave_the_block_decoder(ctxs, oracle):
    """The block decoder starts every block with an undefined offset history.  A frame whose second block opens with a
    repeat code (libzstd writes such frames) cannot be decoded block by block; it goes to the serial paths and comes
    back right."""
    import torch
    import workloads
    from znippy_amd import hip
    fast, high = ctxs
    rec = _records(30_000, 8)                       # every sequence behind the first repeats the offset 1024
    frame = workloads.libzstd_compress(rec, 3)
    assert len(_blocks_any(frame)) > 1
    d_blob = torch.from_numpy(np.frombuffer(frame + bytes(64), np.uint8).copy()).cuda()
    rows = hip.RowTable(high, [0], [len(frame)], [len(rec)], [0], None, np.frombuffer(oracle.blake3(rec), np.uint8).reshape(1, 32))
    d_out = torch.zeros(len(rec) + 64, dtype=torch.uint8, device="cuda")
    c, corrupt, status = rows.decode_verify(d_blob, d_out)
    assert c["corrupt_rows"] == 0 and c["decode_errors"] == 0
    assert d_out[:len(rec)].cpu().numpy().tobytes() == rec
    rows.close()


def _blocks_any(frame):
    """Block list of any zstd frame (window descriptor / content size as the header says)."""
    fhd = frame[4]
    single, fcs_flag, did = (fhd >> 5) & 1, fhd >> 6, fhd & 3
    p = 5 + (0 if single else 1) + (0, 1, 2, 4)[did] + ((1 if single else 0) if fcs_flag == 0 else 1 << fcs_flag)
    out = []
    while True:
        bh = frame[p] | (frame[p + 1] << 8) | (frame[p + 2] << 16)
        typ, size, last = (bh >> 1) & 3, bh >> 3, bh & 1
        out.append((typ, size, last, p + 3))
        p += 3 + (1 if typ == 1 else size)
        if last:
            return out


@pytest.mark.parametrize("kind,cap", [("text", 4e6), ("binary", 12e6)])
def test_real_files_of_the_image_through_both_tiers(ctxs, oracle, kind, cap):
    """Files found in the image (Python / C++ sources; shared objects in 8 MiB rounds), one Round each, through both
    tiers: every frame is decoded by libzstd and compared, the table comes back through the GPU read path with every
    digest verified, and the higher tier is the smaller one."""
    import torch
    import workloads
    from znippy_amd import hip
    ents = workloads.image_corpus(kind, cap, whole_files=False)
    assert len(ents) >= 2
    lens = np.array([len(e) for e in ents], np.uint64)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    total = int(lens.sum())
    d_src = torch.from_numpy(np.frombuffer(b"".join(ents) + bytes(64), np.uint8).copy()).cuda()
    sizes = {}
    for label, ctx in zip(("fast", "high"), ctxs):
        rt = hip.RoundTable(ctx, offs, lens)
        d_blob = torch.zeros(rt.blob_bound() + 64, dtype=torch.uint8, device="cuda")
        enc = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in rt.encode_hash(d_src, d_blob).items()}
        sizes[label] = int(enc["blob_size"].sum())
        hb = d_blob.cpu().numpy()
        for i, e in enumerate(ents):
            f = hb[int(enc["blob_offset"][i]):int(enc["blob_offset"][i] + enc["blob_size"][i])].tobytes()
            assert oracle.libzstd_decompress(f, max(len(e), 1)) == e, (label, i, len(e))
            assert bytes(enc["checksum"][i]) == oracle.blake3(e)
        rows = hip.RowTable(ctx, enc["blob_offset"], enc["blob_size"], lens, offs, None, enc["checksum"])
        d_out = torch.zeros(total + 64, dtype=torch.uint8, device="cuda")
        c, corrupt, status = rows.decode_verify(d_blob, d_out)
        assert c["corrupt_rows"] == 0 and c["decode_errors"] == 0 and c["verified_bytes"] == total
        assert bool((d_out[:total] == d_src[:total]).all())
        rows.close()
        rt.close()
    assert sizes["high"] < sizes["fast"] < total
