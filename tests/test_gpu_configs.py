"""BASELINE.json configs at full size on one MI355X, through the batch C ABI, checked with the
size-independent properties SURVEY §8d names: compress -> decompress round trip is the identity,
every row verifies against the write side's checksum, per-chunk digests equal the oracle's on a
sample of rows ("checksum of checksums"), counters add up.
  C3  single 2 GiB text file at the reference's 8 MiB slices (256 chunks)   perf_bench.rs:L185-194
  C4  500 MiB LCG blob, store path (.jar) and codec path (.bin)               perf_bench.rs:L83-92,L125
  C5  mixed jar-style archive, scaled to ~1.2 GB (3.5k small xml + jars of 100 KiB..2 MiB + big jars)
"""
import numpy as np
import pytest

import gen

pytestmark = pytest.mark.gpu


def _roundtrip(gpu_ctx, oracle, d_src, lens, skip, sample_rows):
    import torch
    from znippy_amd import hip
    lens = np.asarray(lens, dtype=np.uint64)
    n = len(lens)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    rounds = hip.RoundTable(gpu_ctx, offs, lens, skip)
    d_blob = torch.empty(rounds.blob_bound() + 64, dtype=torch.uint8, device="cuda")
    enc = rounds.encode_hash(d_src, d_blob)
    enc = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in enc.items()}
    assert int(enc["blob_offset"][0]) == 0
    assert np.array_equal(enc["blob_offset"][1:], np.cumsum(enc["blob_size"])[:-1])
    want_comp = np.ones(n, np.uint8) if skip is None else 1 - np.asarray(skip, dtype=np.uint8)
    assert np.array_equal(enc["compressed"], want_comp)
    # digests of sampled rows equal the oracle's BLAKE3 of the source bytes
    for r in sample_rows:
        src = d_src[int(offs[r]):int(offs[r] + lens[r])].cpu().numpy()
        assert enc["checksum"][r].tobytes() == oracle.blake3(src), r
    rounds.close()
    total = int(lens.sum())
    d_out = torch.zeros(total + 64, dtype=torch.uint8, device="cuda")
    bitmap = np.packbits(enc["compressed"].astype(bool), bitorder="little")
    rows = hip.RowTable(gpu_ctx, enc["blob_offset"], enc["blob_size"], lens, offs, bitmap, enc["checksum"])
    counters, corrupt, status = rows.decode_verify(d_blob, d_out)
    assert (status == 0).all() and len(corrupt) == 0
    assert counters == dict(total_chunks=n, total_written_bytes=total, verified_bytes=total, corrupt_bytes=0,
                            corrupt_rows=0, decode_errors=0)
    assert torch.equal(d_out[:total], d_src[:total])
    ratio = total / max(enc["blob_bytes"], 1)
    rows.close()
    return enc, ratio


def test_generators_on_device_match_reference_generators():
    import gen_gpu
    for n in (1000, 70000, 200001):
        assert gen_gpu.text(n).cpu().numpy().tobytes() == gen.text(n)
        assert gen_gpu.binary(n).cpu().numpy().tobytes() == gen.binary(n)
        assert gen_gpu.random_lcg(n).cpu().numpy().tobytes() == gen.random_lcg(n)
        assert gen_gpu.incompressible(7, n).cpu().numpy().tobytes() == gen.incompressible(7, n)


def test_c3_single_2gib_text_file_8mib_slices(gpu_ctx, oracle):
    import gen_gpu
    size, slice_ = 2 << 30, 8 << 20
    d_src = gen_gpu.text(size)
    lens = [slice_] * (size // slice_)
    enc, ratio = _roundtrip(gpu_ctx, oracle, d_src, lens, None, sample_rows=[0, 1, 255])
    assert len(lens) == 256 and ratio > 500   # each 128 KiB block is one literal run + one long match


def test_c4_500mib_random_store_path_and_codec_path(gpu_ctx, oracle):
    import gen_gpu
    size, slice_ = 500 << 20, 8 << 20
    d_src = gen_gpu.random_lcg(size)
    lens = [slice_] * (size // slice_) + ([size % slice_] if size % slice_ else [])
    # random.jar: skip extension -> stored as-is (blake3 + copy only)
    enc, ratio = _roundtrip(gpu_ctx, oracle, d_src, lens, np.ones(len(lens), np.uint8), sample_rows=[0, 62])
    assert abs(ratio - 1.0) < 1e-9
    # random.bin: goes through the codec; incompressible -> raw blocks, ~3 bytes per 128 KiB overhead
    enc, ratio = _roundtrip(gpu_ctx, oracle, d_src, lens, None, sample_rows=[0, 62])
    assert 0.999 < ratio <= 1.0


def test_c5_mixed_jar_style_archive(gpu_ctx, oracle):
    """Scaled stand-in for the 'real jars' corpus: mostly store path, skewed sizes."""
    import torch
    import gen_gpu
    parts, lens, skip = [], [], []
    for i in range(3500):                                # small .xml text files 1..8 KiB
        n = 1024 + (i % 8) * 1024
        parts.append(("text", n)); lens.append(n); skip.append(0)
    for i in range(700):                                 # .jar 100 KiB + (i%20)*100 KiB, incompressible
        n = 100 * 1024 + (i % 20) * 100 * 1024
        parts.append(("inc", i, n)); lens.append(n); skip.append(1)
    for i in range(10):                                  # big jars 20 MiB + i*0.5 MiB, cut into 8 MiB slices
        n = (20 << 20) + i * (512 << 10)
        off = 0
        while off < n:
            l = min(8 << 20, n - off)
            parts.append(("inc_slice", 1000 + i, off, l)); lens.append(l); skip.append(1)
            off += l
    total = sum(lens)
    d_src = torch.empty(total + 64, dtype=torch.uint8, device="cuda")
    text_max = gen_gpu.text(8 * 1024 + 1024)
    pos, cache = 0, {}
    for p, l in zip(parts, lens):
        if p[0] == "text":
            d_src[pos:pos + l] = text_max[:l]
        elif p[0] == "inc":
            d_src[pos:pos + l] = gen_gpu.incompressible(p[1], l)
        else:
            key = p[1]
            if key not in cache:
                cache.clear()
                cache[key] = gen_gpu.incompressible(key, (20 << 20) + (key - 1000) * (512 << 10))
            d_src[pos:pos + l] = cache[key][p[2]:p[2] + l]
        pos += l
    enc, ratio = _roundtrip(gpu_ctx, oracle, d_src, lens, np.array(skip, np.uint8), sample_rows=[0, 7, 3499, 3500, 4199, 4200, len(lens) - 1])
    assert total > 10**9 and 1.0 < ratio < 1.1
    # the row-cursor split for 8 ranks is balanced by bytes although row counts are wildly skewed
    from znippy_amd.sharding import split_rows
    parts8 = split_rows(lens, 8)
    loads = [sum(lens[a:b]) for a, b in parts8]
    assert max(loads) < 1.5 * total / 8


def test_c3_single_2gib_text_file_200mib_slices(gpu_ctx, oracle):
    """BASELINE configs[2] as worded there: 'split into 200MB Magazine slices' = 11 rounds of <= 200 MiB (in the
    reference 200 MiB is the slot size, slot_packer.rs:L30; a round this large is format-legal)."""
    import workloads
    import torch
    wl = workloads.build("c3slot", torch)
    assert len(wl["lens"]) == 11 and int(wl["lens"].sum()) == 2 << 30 and int(wl["lens"].max()) == 200 << 20
    enc, ratio = _roundtrip(gpu_ctx, oracle, wl["d_src"], wl["lens"], None, sample_rows=[10])
    assert ratio > 500


def test_c5_full_size_mixed_archive_as_8_row_ranges(gpu_ctx, oracle):
    """BASELINE configs[4] at its stated size (~6.2 GB, 5,000 files) on ONE card: the archive is
    decoded once as a whole and once as the 8 contiguous row ranges `split_rows` hands to 8 ranks — each range its
    own row table over only its own blobs (blob_base = its first blob), as decompress_archive does per rank.  The
    summed counters equal the single-table run (= what the RCCL all-reduce of the counters yields), both outputs
    equal the source, sampled digests equal the oracle's."""
    import torch
    import workloads
    from znippy_amd import hip
    from znippy_amd.sharding import split_rows
    wl = workloads.build("c5", torch)
    d_src, lens, skip = wl["d_src"], wl["lens"], wl["skip"]
    n, total = len(lens), int(lens.sum())
    assert total > 5 * 10**9 and n > 5000
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    rounds = hip.RoundTable(gpu_ctx, offs, lens, skip)
    d_blob = torch.empty(rounds.blob_bound() + 64, dtype=torch.uint8, device="cuda")
    enc = rounds.encode_hash(d_src, d_blob)
    enc = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in enc.items()}
    rounds.close()
    bo, bs, ck = enc["blob_offset"], enc["blob_size"], enc["checksum"]
    assert int(bo[0]) == 0 and np.array_equal(bo[1:], np.cumsum(bs)[:-1])
    assert np.array_equal(enc["compressed"], 1 - skip)
    # checksum of checksums on a sample: small xml rows, jar rows, 8 MiB slices of the big jars, the last row
    for r in (0, 7, 3499, 3500, 3519, 4899, 4900, 4903, n - 1):
        src = d_src[int(offs[r]):int(offs[r] + lens[r])].cpu().numpy()
        assert ck[r].tobytes() == oracle.blake3(src), r
    bitmap = np.packbits(enc["compressed"].astype(bool), bitorder="little")
    want = dict(total_chunks=n, total_written_bytes=total, verified_bytes=total, corrupt_bytes=0, corrupt_rows=0,
                decode_errors=0)
    d_out = torch.zeros(total + 64, dtype=torch.uint8, device="cuda")
    whole = hip.RowTable(gpu_ctx, bo, bs, lens, offs, bitmap, ck)
    counters, corrupt, status = whole.decode_verify(d_blob, d_out)
    whole.close()
    assert counters == want and (status == 0).all() and len(corrupt) == 0
    assert torch.equal(d_out[:total], d_src[:total])
    # 8 ranges
    d_out.zero_()
    parts = split_rows(lens, 8)
    loads = [int(lens[a:b].sum()) for a, b in parts]
    assert max(loads) < 1.5 * total / 8 and min(b - a for a, b in parts) > 0
    summed = {k: 0 for k in want}
    for a, b in parts:
        lo, hi = int(bo[a]), int(bo[b - 1] + bs[b - 1])
        rt = hip.RowTable(gpu_ctx, bo, bs, lens, offs, bitmap, ck, row_begin=a, row_end=b)
        c, corrupt, status = rt.decode_verify(d_blob[lo:hi + 64], d_out, blob_base=lo, blob_cap=hi - lo)   # this rank's blobs only
        rt.close()
        assert (status == 0).all() and len(corrupt) == 0 and c["total_chunks"] == b - a
        for k in summed:
            summed[k] += c[k]
    assert summed == want
    assert torch.equal(d_out[:total], d_src[:total])
