"""GPU parity for the write side: BLAKE3 + zstd encode of Rounds through the C ABI.

Replaces the barrel/writer bodies (stream_packer.rs:L217-284, slot_packer.rs:L551-609) and
CompressCtx::compress_into (codec.rs:L43-55).  Compressed BYTES cannot equal the reference's
(OpenZL framing is unpinned); what is bit-exact is everything the format pins:
decoded bytes (checked with three independent decoders: libzstd, the oracle, the GPU decoder),
per-chunk BLAKE3, the compressed flag, uncompressed_size, and blob packing (offsets are the
running sum of on_disk_len from 0 with no gaps, stream_packer.rs:L258).
"""
import numpy as np
import pytest

import gen

pytestmark = pytest.mark.gpu


CASES = [("text", 10240), ("binary", 10240), ("random_lcg", 10240), ("pseudo_text", 5000), ("pseudo_text", 300000),
         ("text", 1), ("text", 0), ("text", 7), ("text", 8), ("zeros", 200000), ("text", 1 << 20),
         ("pseudo_text", (1 << 20) + 12345), ("binary", 3 << 20), ("random_lcg", 300000), ("text", 131072),
         ("text", 131073), ("zeros", 131072 * 2)]


@pytest.fixture(scope="module", params=[1, 19], ids=["level1", "level19"])
def gpu_ctx(request):
    """Every test of this file runs on both effort tiers of the encoder (znippy_ctx_set_level: 1-3 fast, 4-22 high)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from znippy_amd import hip
    ctx = hip.Context(0)
    ctx.set_level(request.param)
    assert ctx.level == request.param
    yield ctx
    ctx.close()


def _gen(name, n):
    return bytes(n) if name == "zeros" else getattr(gen, name)(n)


@pytest.mark.parametrize("gname,n", CASES)
def test_shim_roundtrip_three_decoders(gpu_ctx, oracle, gname, n):
    data = _gen(gname, n)
    frame = gpu_ctx.compress(data)
    assert len(frame) <= gpu_ctx.compress_bound(n)
    assert oracle.zstd_decompressed_size(frame) == n
    assert oracle.libzstd_decompress(frame, n) == data      # independent RFC 8878 decoder
    assert oracle.zstd_decompress(frame) == data            # oracle restatement
    assert gpu_ctx.decompress(frame) == data                # GPU decoder


def test_compressible_inputs_actually_shrink(gpu_ctx):
    assert len(gpu_ctx.compress(gen.text(10240))) < 200
    assert len(gpu_ctx.compress(gen.binary(10240))) < 400
    assert len(gpu_ctx.compress(bytes(1 << 20))) < 1000
    assert len(gpu_ctx.compress(gen.pseudo_text(300000))) < 300000 * 0.9  # raw literals, greedy parse
    n = len(gpu_ctx.compress(gen.random_lcg(300000)))
    assert 300000 < n <= 300000 + 3 * 3 + 16    # incompressible -> raw blocks, tiny overhead


def test_rounds_batch_matches_write_loop_metadata(gpu_ctx, oracle):
    """A mixed batch of Rounds incl. skip (store) rounds, an empty round and a multi-block round."""
    import torch
    from znippy_amd import hip
    rng = np.random.default_rng(3)
    entries, skip = [], []
    for i in range(200):
        kind = i % 5
        n = int(rng.integers(0, 50000))
        e = [gen.text(n), gen.binary(n), gen.pseudo_text(n, seed=i), gen.incompressible(i, n), b""][kind]
        entries.append(e)
        skip.append(1 if kind == 3 else 0)
    entries += [gen.pseudo_text(700000, seed=5), gen.incompressible(9, 2_500_000), gen.text(400000)]
    skip += [0, 1, 0]
    lens = np.array([len(e) for e in entries], dtype=np.uint64)
    # ragged, unaligned placement inside the staging buffer
    offs, pos = [], 1
    for L in lens:
        offs.append(pos)
        pos += int(L) + int(rng.integers(0, 5))
    buf = np.zeros(pos + 64, dtype=np.uint8)
    for o, e in zip(offs, entries):
        buf[o:o + len(e)] = np.frombuffer(e, dtype=np.uint8)
    d_src = torch.from_numpy(buf).cuda()
    rt = hip.RoundTable(gpu_ctx, offs, lens, skip)
    d_blob = torch.zeros(rt.blob_bound() + 64, dtype=torch.uint8, device="cuda")
    res = rt.encode_hash(d_src, d_blob)
    want = oracle.compress_rounds(buf, np.array(offs, dtype=np.uint64), lens, np.array(skip, dtype=np.uint8),
                                  level=3, n_threads=1)
    assert np.array_equal(res["checksum"], want["checksum"])          # BLAKE3 of the ORIGINAL bytes
    assert np.array_equal(res["compressed"], want["compressed"])      # skip -> compressed=false
    # blobs packed back-to-back from 0 in round order
    assert int(res["blob_offset"][0]) == 0
    assert np.array_equal(res["blob_offset"][1:], np.cumsum(res["blob_size"])[:-1])
    assert res["blob_bytes"] == int(res["blob_size"].sum())
    blob = d_blob[:res["blob_bytes"]].cpu().numpy()
    for i, e in enumerate(entries):
        b = blob[int(res["blob_offset"][i]):int(res["blob_offset"][i] + res["blob_size"][i])].tobytes()
        if skip[i]:
            assert b == e
        else:
            assert oracle.libzstd_decompress(b, len(e)) == e, i
    # and the GPU read path consumes the GPU-written archive: counters equal the oracle loop's
    out_off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    bitmap = np.packbits(res["compressed"].astype(bool), bitorder="little")
    rows = hip.RowTable(gpu_ctx, res["blob_offset"], res["blob_size"], lens, out_off, bitmap, res["checksum"])
    d_out = torch.zeros(int(lens.sum()) + 64, dtype=torch.uint8, device="cuda")
    counters, corrupt, status = rows.decode_verify(d_blob, d_out)
    wantc, _ = oracle.decompress_rows(blob, res["blob_offset"], res["blob_size"], lens, out_off, bitmap,
                                      res["checksum"], 0, len(entries))
    assert counters == wantc and (status == 0).all() and len(corrupt) == 0
    assert d_out[:int(lens.sum())].cpu().numpy().tobytes() == b"".join(entries)


def test_blob_cap_too_small_is_reported(gpu_ctx):
    import torch
    from znippy_amd import hip
    from znippy_amd._lib import ZnippyError, E_DST_SMALL
    data = np.frombuffer(gen.random_lcg(100000), dtype=np.uint8)
    d_src = torch.from_numpy(data.copy()).cuda()
    rt = hip.RoundTable(gpu_ctx, [0], [100000])
    d_blob = torch.zeros(50000, dtype=torch.uint8, device="cuda")
    with pytest.raises(ZnippyError) as ei:
        rt.encode_hash(d_src, d_blob)
    assert ei.value.code == E_DST_SMALL


def test_c2_shape_100k_rounds(gpu_ctx, oracle):
    """BASELINE C2 at full size on the write side: 100k identical 10 KiB text chunks."""
    import torch
    from znippy_amd import hip
    n, sz = 100_000, 10240
    chunk = gen.text(sz)
    d_src = torch.from_numpy(np.tile(np.frombuffer(chunk, dtype=np.uint8), n)).cuda()
    rt = hip.RoundTable(gpu_ctx, np.arange(n, dtype=np.uint64) * sz, np.full(n, sz, np.uint64))
    d_blob = torch.zeros(rt.blob_bound() + 64, dtype=torch.uint8, device="cuda")
    res = rt.encode_hash(d_src, d_blob)
    want = np.frombuffer(oracle.blake3(chunk), dtype=np.uint8)
    assert (res["checksum"] == want[None, :]).all()
    assert (res["blob_size"] == res["blob_size"][0]).all() and int(res["blob_size"][0]) < 200
    assert np.array_equal(res["blob_offset"], np.arange(n, dtype=np.uint64) * res["blob_size"][0])
    blob = d_blob[:res["blob_bytes"]].cpu().numpy()
    fl = int(res["blob_size"][0])
    frames = blob.reshape(n, fl)
    assert (frames == frames[0][None, :]).all()            # identical inputs -> identical frames
    assert oracle.libzstd_decompress(frames[0].tobytes(), sz) == chunk


def test_store_if_incompressible_opt_in(gpu_ctx, oracle):
    """with the opt-in flag, incompressible rounds are stored raw (compressed=0) and the
    read path passes them through; default behaviour (flag off) keeps them in raw-block frames."""
    import torch
    from znippy_amd import hip
    entries = [gen.random_lcg(300000), gen.text(20000), gen.incompressible(4, 5000), b"", gen.pseudo_text(9000, 2)]
    lens = np.array([len(e) for e in entries], dtype=np.uint64)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    d_src = torch.from_numpy(np.frombuffer(b"".join(entries) + bytes(64), dtype=np.uint8).copy()).cuda()
    for flag, want_comp in ((False, [1, 1, 1, 1, 1]), (True, [0, 1, 0, 1, 1])):
        rt = hip.RoundTable(gpu_ctx, offs, lens)
        if flag:
            rt.set_store_incompressible(True)
        d_blob = torch.zeros(rt.blob_bound() + 64, dtype=torch.uint8, device="cuda")
        res = rt.encode_hash(d_src, d_blob)
        res = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in res.items()}
        assert list(res["compressed"]) == want_comp
        assert np.array_equal(res["blob_offset"][1:], np.cumsum(res["blob_size"])[:-1])
        blob = d_blob[:res["blob_bytes"]].cpu().numpy()
        for i, e in enumerate(entries):
            b = blob[int(res["blob_offset"][i]):int(res["blob_offset"][i] + res["blob_size"][i])].tobytes()
            if res["compressed"][i]:
                assert oracle.libzstd_decompress(b, max(len(e), 1)) == e
            else:
                assert b == e
        bitmap = np.packbits(res["compressed"].astype(bool), bitorder="little")
        rows = hip.RowTable(gpu_ctx, res["blob_offset"], res["blob_size"], lens, offs, bitmap, res["checksum"])
        d_out = torch.zeros(int(lens.sum()) + 64, dtype=torch.uint8, device="cuda")
        counters, corrupt, status = rows.decode_verify(d_blob, d_out)
        assert counters["verified_bytes"] == int(lens.sum()) and len(corrupt) == 0
        assert d_out[:int(lens.sum())].cpu().numpy().tobytes() == b"".join(entries)


def _first_block_literals_type(frame):
    """Literals_Block_Type of the frame's first block (single-segment frame as this encoder writes it)."""
    fhd = frame[4]
    fcs = {0: 1, 1: 2, 2: 4, 3: 8}[fhd >> 6] if (fhd >> 5) & 1 else (0 if fhd >> 6 == 0 else 1 << (fhd >> 6))
    p = 5 + (0 if (fhd >> 5) & 1 else 1) + fcs
    bh = frame[p] | (frame[p + 1] << 8) | (frame[p + 2] << 16)
    return (bh >> 1) & 3, frame[p + 3] & 3


def _skewed(n, seed, alphabet=b"etaoinshrdlucmfw .,;\n"):
    """Letters drawn with a skewed distribution: few 4-byte repeats, low entropy per byte."""
    rng = np.random.default_rng(seed)
    w = 1.0 / np.arange(1, len(alphabet) + 1)
    idx = rng.choice(len(alphabet), size=n, p=w / w.sum())
    return np.frombuffer(alphabet, np.uint8)[idx].tobytes()


@pytest.mark.parametrize("n", [40_000, 131_072, 300_001, 900])
def test_huffman_literals_and_dense_sequences(gpu_ctx, oracle, n):
    """Wide-variant entropy stages: Huffman-coded literals (direct weights, 1 or 4 streams), the wave-parallel
    sequences bitstream, a block of literals only — checked by libzstd, the oracle and the GPU decoder."""
    for data in (gen.pseudo_text(n, seed=n), _skewed(n, n)):
        frame = gpu_ctx.compress(data)
        assert oracle.libzstd_decompress(frame, n) == data
        assert oracle.zstd_decompress(frame) == data
        assert gpu_ctx.decompress(frame) == data
        if n >= 40_000:
            btype, ltype = _first_block_literals_type(frame)
            assert btype == 2 and ltype == 2, (btype, ltype)   # compressed block with Huffman literals
            assert len(frame) < 0.75 * n


def test_small_real_rounds_are_handed_to_the_wide_variant(gpu_ctx, oracle):
    """Rounds <= 16 KiB of non-periodic text exhaust the small variant's sequence budget and are re-encoded by the
    wide variant; periodic rounds of the same size are not.  Either way the frames decode everywhere and the frame
    of a round does not depend on its neighbours in the batch."""
    import torch
    from znippy_amd import hip
    ents = [gen.pseudo_text(6000 + 37 * i, seed=i) for i in range(40)] + [gen.text(10240)] * 8 + [_skewed(9000, 3)]
    lens = np.array([len(e) for e in ents], np.uint64)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    d_src = torch.from_numpy(np.frombuffer(b"".join(ents) + bytes(64), np.uint8).copy()).cuda()
    rt = hip.RoundTable(gpu_ctx, offs, lens)
    d_blob = torch.zeros(rt.blob_bound() + 64, dtype=torch.uint8, device="cuda")
    enc = rt.encode_hash(d_src, d_blob)
    blob = d_blob.cpu().numpy()
    for i, e in enumerate(ents):
        f = blob[int(enc["blob_offset"][i]):int(enc["blob_offset"][i] + enc["blob_size"][i])].tobytes()
        assert oracle.libzstd_decompress(f, len(e)) == e, i
        assert f == gpu_ctx.compress(e), i                     # same bytes alone and inside the batch
        assert enc["checksum"][i].tobytes() == oracle.blake3(e)
    sizes = enc["blob_size"]
    assert sizes[:40].sum() < 0.62 * lens[:40].sum()            # the n/40 budget alone leaves ~0.9 here
    assert (sizes[40:48] < 200).all()


def test_gather_of_small_rounds_every_piece_size(gpu_ctx, oracle):
    """Tables of rounds of at most 16 KiB are gathered a lane per piece (zstd_encode.hip, k_gather): pieces of up to 512 bytes
    by their own lanes, longer ones by the whole wave, four at a time (the first KiB of each of the four loaded before any
    of it is stored, the rest piece by piece).  Incompressible rounds make frames a few bytes longer than their input, so
    the piece sizes here sit on every boundary of that code: 0, around 16 / 512 / 1024 / 1040, odd lengths, 16 KiB, in an
    order that mixes them inside a wave — the blob region must be the frames back to back, every frame must decode."""
    import torch
    from znippy_amd import hip
    rng = np.random.default_rng(11)
    edge = [0, 1, 5, 15, 16, 17, 480, 495, 500, 503, 505, 511, 512, 513, 1000, 1005, 1010, 1015, 1023, 1024, 1025, 1039, 1040, 1041,
            2047, 2048, 2049, 4095, 5000, 8191, 16000, 16384]
    sizes = edge + [int(x) for x in rng.integers(0, 16385, 380)] + edge[::-1]
    ents = [gen.incompressible(900 + i, n) if i % 3 else gen.pseudo_text(n, seed=i) for i, n in enumerate(sizes)]
    lens = np.array([len(e) for e in ents], np.uint64)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    d_src = torch.from_numpy(np.frombuffer(b"".join(ents) + bytes(64), np.uint8).copy()).cuda()
    rt = hip.RoundTable(gpu_ctx, offs, lens)
    d_blob = torch.full((rt.blob_bound() + 64,), 0x3C, dtype=torch.uint8, device="cuda")
    enc = rt.encode_hash(d_src, d_blob)
    assert "gather" in dict(gpu_ctx.kernel_times())
    blob = d_blob.cpu().numpy()
    pos = 0
    for i, e in enumerate(ents):
        o, n = int(enc["blob_offset"][i]), int(enc["blob_size"][i])
        assert o == pos, i                                     # packed without gaps, in round order
        pos += n
        assert oracle.libzstd_decompress(blob[o:o + n].tobytes(), len(e)) == e, (i, len(e), n)
        assert enc["checksum"][i].tobytes() == oracle.blake3(e)
    assert pos == enc["blob_bytes"] and (blob[pos:] == 0x3C).all()
    rt.close()


@pytest.mark.parametrize("sizes", [[5 << 20], [1 << 20, 3 << 20, 160, 70000 * 16, (2 << 20) + 7],
                                   [1 << 20, (1 << 20) + 3, 1 << 20], [100, 1 << 20]])
def test_store_path_tables_blob_is_the_input(gpu_ctx, oracle, sizes):
    """Tables of skip rounds only (the one-big-jar case): with every blob offset a multiple of 16 the hash kernel
    copies the rounds while it hashes them (no gather pass over the stored bytes); with an odd length in the middle
    the two-pass form runs.  Either way the blob region is the input, packed without gaps, and the digests are right."""
    import torch
    from znippy_amd import hip
    data = [gen.incompressible(i + 1, n) for i, n in enumerate(sizes)]
    src = np.frombuffer(b"".join(data), dtype=np.uint8)
    lens = np.array(sizes, dtype=np.uint64)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    d_src = torch.from_numpy(src.copy()).cuda()
    rounds = hip.RoundTable(gpu_ctx, offs, lens, np.ones(len(sizes), np.uint8))
    d_blob = torch.zeros(rounds.blob_bound() + 64, dtype=torch.uint8, device="cuda")
    for _ in range(2):
        d_blob.zero_()
        enc = rounds.encode_hash(d_src, d_blob)
        assert list(enc["blob_offset"]) == list(offs) and list(enc["blob_size"]) == sizes
        assert not enc["compressed"].any() and enc["blob_bytes"] == len(src)
        assert np.array_equal(d_blob.cpu().numpy()[:len(src)], src)
        for i, d in enumerate(data):
            assert bytes(enc["checksum"][i]) == oracle.blake3(d)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_round_tables_encode_decode_and_repeat(gpu_ctx, oracle, seed):
    """Randomised Round tables (every kind of content, block-boundary sizes, runs of equal rounds, some on the store path):
    every frame decodes with libzstd to its input, every digest is the oracle's, a second run writes the same blob region,
    and this library's read side gets every byte back."""
    import torch
    from znippy_amd import hip
    rng = np.random.default_rng(seed)
    entries, skip = [], []
    while len(entries) < 500:
        kind = int(rng.integers(0, 7))
        run = int(rng.integers(1, 12)) if rng.random() < 0.4 else 1
        n = int(rng.choice([0, 1, 63, 1024, 4096, 10240, 10240, 16384, 16385, 20480, 131072, 131073, int(rng.integers(2, 300000))]))
        if kind <= 1: e = gen.text(n)
        elif kind == 2: e = gen.binary(n)
        elif kind == 3: e = gen.pseudo_text(min(n, 60000), seed=len(entries) + seed * 1000)
        elif kind == 4: e = gen.incompressible(len(entries) + seed, min(n, 200000))
        elif kind == 5: e = bytes(n)
        else: e = (gen.pseudo_text(min(n, 30000) // 2 + 1, seed=seed) + gen.incompressible(seed, min(n, 30000) // 2))[:n]
        for _ in range(run):
            entries.append(e); skip.append(1 if kind == 4 and len(entries) % 4 == 0 else 0)
    lens = np.array([len(e) for e in entries], np.uint64)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    total = int(lens.sum())
    d_src = torch.from_numpy(np.frombuffer(b"".join(entries) + bytes(64), np.uint8).copy()).cuda()
    rt = hip.RoundTable(gpu_ctx, offs, lens, np.array(skip, np.uint8))
    d_blob = torch.zeros(rt.blob_bound() + 64, dtype=torch.uint8, device="cuda")
    first = None
    for rep in range(2):
        d_blob.zero_()
        enc = rt.encode_hash(d_src, d_blob)
        hb = d_blob.cpu().numpy()
        got = (enc["blob_offset"].copy(), enc["blob_size"].copy(), enc["checksum"].copy(), hb[:int(enc["blob_bytes"])].copy())
        if first is None:
            first = got
            for i, e in enumerate(entries):
                f = hb[int(enc["blob_offset"][i]):int(enc["blob_offset"][i] + enc["blob_size"][i])].tobytes()
                assert enc["checksum"][i].tobytes() == oracle.blake3(e), i
                if enc["compressed"][i]:
                    assert oracle.libzstd_decompress(f, max(len(e), 1)) == e, (i, len(e))
                else:
                    assert f == e, i
        else:
            assert all((a == b).all() for a, b in zip(got, first)), "the second run wrote something else"
    rows = hip.RowTable(gpu_ctx, enc["blob_offset"], enc["blob_size"], lens, offs,
                        np.packbits(enc["compressed"].astype(bool), bitorder="little"), enc["checksum"])
    d_out = torch.zeros(total + 64, dtype=torch.uint8, device="cuda")
    c, corrupt, st = rows.decode_verify(d_blob, d_out)
    assert c["corrupt_rows"] == 0 and c["decode_errors"] == 0 and c["verified_bytes"] == total
    assert torch.equal(d_out[:total], d_src[:total])
