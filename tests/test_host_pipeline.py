"""Host-side pipelines, written to read like the reference's own integration tests
(tests/tests/integration_test.rs; line numbers cited per test).  Each test runs twice:
  * backend "oracle": CPU checker double (tests/oracle_backend.py)      -> -m "not gpu"
  * backend "hip":    the product path through libznippy_hip.so          -> -m gpu
"""
import os
import struct

import numpy as np
import pytest

import gen
from znippy_amd import index as ix
from znippy_amd.archive import ZnippyArchive
from znippy_amd.decompress import decompress_archive, verify_archive_integrity
from znippy_amd.stream_packer import ArchiveEntry, compress_stream, plan_rounds, SLICE_SIZE


@pytest.fixture(params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def backend(request, oracle):
    if request.param == "oracle":
        from oracle_backend import OracleBackend
        return OracleBackend()
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from znippy_amd.backend import default_backend
    return default_backend()


def decompress_to_map(archive_path, backend, tmp_path_factory):
    out_dir = tmp_path_factory.mktemp("out")
    report = decompress_archive(archive_path, True, out_dir, backend=backend)
    assert report.corrupt_files == 0, "Corrupt files detected during decompression"
    result = {}
    for root, _, files in os.walk(out_dir):
        for f in files:
            full = os.path.join(root, f)
            result[os.path.relpath(full, out_dir)] = open(full, "rb").read()
    return result


def test_stream_compress_single_small_file(backend, tmp_path, tmp_path_factory):  # L39-68
    archive_path = tmp_path / "test.znippy"
    content = b"Hello, Znippy! This is a small test file."
    c = compress_stream(archive_path, False, backend=backend)
    c.sender().send(ArchiveEntry("hello.txt", content))
    report = c.finish()
    assert report.total_files == 1
    assert archive_path.exists()
    assert not archive_path.with_suffix(".zdata").exists()
    files = decompress_to_map(archive_path, backend, tmp_path_factory)
    assert files == {"hello.txt": content}


def test_stream_compress_multiple_files(backend, tmp_path, tmp_path_factory):  # L70-109
    archive_path = tmp_path / "multi.znippy"
    files_in = [("file1.txt", b"Content of file 1"), ("subdir/file2.txt", b"Content of file 2 in subdir"),
                ("binary.bin", bytes(x % 256 for x in range(4096)))]
    c = compress_stream(archive_path, False, backend=backend)
    for p, d in files_in:
        c.sender().send(ArchiveEntry(p, d))
    assert c.finish().total_files == 3
    files_out = decompress_to_map(archive_path, backend, tmp_path_factory)
    assert files_out == dict(files_in)


def test_stream_compress_empty_file(backend, tmp_path):  # L111-131: empty file still gets one row
    archive_path = tmp_path / "empty.znippy"
    c = compress_stream(archive_path, False, backend=backend)
    c.sender().send(ArchiveEntry("empty.txt", b""))
    assert c.finish().total_files == 1
    _, batches = ix.read_znippy_index(str(archive_path))
    assert batches[0].num_rows == 1


def test_stream_compress_large_file_multi_chunk(backend, tmp_path, tmp_path_factory):  # L134-158
    archive_path = tmp_path / "large.znippy"
    data = gen.binary(12 * 1024 * 1024)
    c = compress_stream(archive_path, False, backend=backend)
    c.sender().send(ArchiveEntry("large.bin", data))
    report = c.finish()
    assert report.total_files == 1
    assert report.chunks >= 2
    assert decompress_to_map(archive_path, backend, tmp_path_factory)["large.bin"] == data


def test_stream_compress_already_compressed_file_skipped(backend, tmp_path, tmp_path_factory):  # L161-185
    archive_path = tmp_path / "skip.znippy"
    data = b"\xAA" * 1024
    c = compress_stream(archive_path, False, backend=backend)
    c.sender().send(ArchiveEntry("image.png", data))
    report = c.finish()
    assert report.total_files == 1 and report.uncompressed_files == 1
    assert decompress_to_map(archive_path, backend, tmp_path_factory)["image.png"] == data
    _, batches = ix.read_znippy_index(str(archive_path))
    assert batches[0].column(3).to_pylist() == [False]  # stored as-is: compressed=false, blob = raw bytes
    assert batches[0].column(6).to_pylist() == [1024]


def test_stream_compress_no_skip_forces_compression(backend, tmp_path, tmp_path_factory):  # L187-211
    archive_path = tmp_path / "noskip.znippy"
    data = b"\xBB" * 2048
    c = compress_stream(archive_path, True, backend=backend)
    c.sender().send(ArchiveEntry("image.png", data))
    report = c.finish()
    assert report.compressed_files == 1 and report.uncompressed_files == 0
    assert decompress_to_map(archive_path, backend, tmp_path_factory)["image.png"] == data


def test_stream_compress_empty_archive(backend, tmp_path):  # L213-224
    c = compress_stream(tmp_path / "none.znippy", False, backend=backend)
    assert c.finish().total_files == 0
    schema, batches = ix.read_znippy_index(str(tmp_path / "none.znippy"))
    assert sum(b.num_rows for b in batches) == 0


def test_output_extension_forced(backend, tmp_path):  # stream_packer.rs:L132
    c = compress_stream(tmp_path / "archive.tmp", False, backend=backend)
    c.sender().send(ArchiveEntry("a", b"a"))
    c.finish()
    assert (tmp_path / "archive.znippy").exists()


def test_index_schema_fields():  # L359-377
    assert ix.znippy_index_schema().names == ["relative_path", "chunk_seq", "fdata_offset", "compressed",
                                              "uncompressed_size", "blob_offset", "blob_size", "checksum"]


def test_read_znippy_index_after_compress(backend, tmp_path):  # L380-410
    p = tmp_path / "idx.znippy"
    c = compress_stream(p, False, backend=backend)
    c.sender().send(ArchiveEntry("test.txt", b"test data for index read"))
    c.finish()
    schema, batches = ix.read_znippy_index(str(p))
    assert batches and batches[0].num_rows == 1
    md = {k.decode(): v.decode() for k, v in schema.metadata.items()}
    assert "compression_level" in md and "checksum_group_0" not in md
    assert md["znippy_format_version"] == "3" and len(md) == 9


def test_verify_via_decompress(backend, tmp_path):  # L415-443
    p = tmp_path / "verify.znippy"
    data = bytes(i % 127 for i in range(10000))
    c = compress_stream(p, False, backend=backend)
    c.sender().send(ArchiveEntry("check.bin", data))
    c.finish()
    out = tmp_path / "o"
    report = decompress_archive(p, True, out, backend=backend)
    assert (report.corrupt_files, report.total_files, report.verified_files) == (0, 1, 1)
    assert report.verified_bytes > 0
    assert (out / "check.bin").read_bytes() == data
    v = verify_archive_integrity(p, backend=backend)   # save_data=false path (index.rs:L550-553)
    assert (v.total_files, v.corrupt_files, v.total_bytes, v.chunks) == (1, 0, 10000, 1)


def test_list_archive_contents(backend, tmp_path):  # L445-470
    p = tmp_path / "list.znippy"
    c = compress_stream(p, False, backend=backend)
    c.sender().send(ArchiveEntry("foo.txt", b"foo"))
    c.sender().send(ArchiveEntry("bar.txt", b"bar"))
    c.finish()
    assert ix.list_archive_contents(str(p)) == [("foo.txt", 3), ("bar.txt", 3)]


def test_manifest_roundtrip():  # L474-497
    entries = [ix.ManifestEntry(1, "central", "maven", 0, 1024, 42), ix.ManifestEntry(2, "crates-io", "cargo", 1024, 512, 17)]
    assert ix.read_manifest_bytes(ix.write_manifest_bytes(entries)) == entries


def test_manifest_empty_roundtrip():  # L499-505
    assert ix.read_manifest_bytes(ix.write_manifest_bytes([])) == []


def test_interpret_footer_single_v06():  # L509-517
    assert ix.interpret_footer(struct.pack("<Q", 12345)) == ("single", 12345)


def test_interpret_footer_multi_v07():  # L520-531
    assert ix.interpret_footer(ix.MULTI_INDEX_MAGIC + struct.pack("<Q", 99999)) == ("multi", 99999)


def test_multi_index_write_read_roundtrip(backend, tmp_path, tmp_path_factory):  # L534-581
    p = tmp_path / "multi.znippy"
    c = compress_stream(p, False, backend=backend)
    c.sender().send(ArchiveEntry("pom.xml", b"<project/>", 1, "maven"))
    c.sender().send(ArchiveEntry("lib.jar", b"JAR_CONTENT", 1, "maven"))
    c.sender().send(ArchiveEntry("Cargo.toml", b"[package]", 2, "cargo"))
    c.finish()
    manifest = ix.read_znippy_manifest(str(p))
    assert len(manifest) == 2
    maven = next(e for e in manifest if e.repo == "maven")
    cargo = next(e for e in manifest if e.repo == "cargo")
    assert (maven.pkg_type, maven.row_count, cargo.pkg_type, cargo.row_count) == (1, 2, 2, 1)
    files = decompress_to_map(p, backend, tmp_path_factory)
    assert files == {"pom.xml": b"<project/>", "lib.jar": b"JAR_CONTENT", "Cargo.toml": b"[package]"}
    # container tail bytes: [..manifest]["ZNPYMIDX"][LE u64 manifest_offset] (meta_sink.rs:L103-118)
    raw = p.read_bytes()
    assert raw[-16:-8] == b"ZNPYMIDX"
    moff = struct.unpack("<Q", raw[-8:])[0]
    assert moff == max(e.index_offset + e.index_len for e in manifest)
    assert min(e.index_offset for e in manifest) == ix.blob_region_end(str(p))


def test_single_group_writes_v07(backend, tmp_path, tmp_path_factory):  # L584-614
    p = tmp_path / "single.znippy"
    c = compress_stream(p, False, backend=backend)
    c.sender().send(ArchiveEntry("a.txt", b"aaa", 1, "r1"))
    c.sender().send(ArchiveEntry("b.txt", b"bbb", 1, "r1"))
    c.finish()
    manifest = ix.read_znippy_manifest(str(p))
    assert len(manifest) == 1 and (manifest[0].pkg_type, manifest[0].repo, manifest[0].row_count) == (1, "r1", 2)
    assert len(decompress_to_map(p, backend, tmp_path_factory)) == 2


def test_znippy_archive_extract_file_multi_chunk(backend, tmp_path):  # L617-642
    p = tmp_path / "extract.znippy"
    data = gen.binary(12 * 1024 * 1024)
    c = compress_stream(p, False, backend=backend)
    c.sender().send(ArchiveEntry("big.bin", data))
    assert c.finish().chunks >= 2
    a = ZnippyArchive.open(p, backend=backend)
    got = a.extract_file("big.bin")
    assert len(got) == len(data) and got == data
    assert a.contains("big.bin") and not a.contains("nope") and a.file_size("big.bin") == len(data)
    with pytest.raises(KeyError):
        a.extract_file("nope")


def test_repro_crate_roundtrip_scaled(backend, tmp_path):  # repro_crate.rs:L19-67 at 500 blobs
    n, sizes = 500, [1_000, 5_000, 10_000, 50_000, 100_000]
    base = tmp_path / "repro_crates"
    c = compress_stream(base, True, backend=backend)   # no_skip = true: incompressible blobs go through the codec
    for i in range(n):
        c.sender().send(ArchiveEntry(f"bench-crate-{i:06}-1.0.0.crate", gen.incompressible(i, sizes[i % 5])))
    report = c.finish()
    assert report.total_files == n and report.compressed_files == n
    archive = tmp_path / "repro_crates.znippy"
    verify = decompress_archive(archive, True, tmp_path / "out", backend=backend)
    assert verify.corrupt_files == 0 and verify.total_files == n
    a = ZnippyArchive.open(archive, backend=backend)
    for i in range(0, n, 137):
        name = f"bench-crate-{i:06}-1.0.0.crate"
        assert a.extract_file(name) == gen.incompressible(i, sizes[i % 5])


def test_corrupt_blob_is_reported_not_fatal(backend, tmp_path):
    """decompress.rs:L175-189: mismatch counted (corrupt_files = corrupt ROWS), bytes still written."""
    p = tmp_path / "c.znippy"
    c = compress_stream(p, False, backend=backend)
    for i in range(5):
        c.sender().send(ArchiveEntry(f"f{i}.png", gen.incompressible(i, 3000)))   # stored rows: flip a payload byte
    c.finish()
    raw = bytearray(p.read_bytes())
    raw[3000 * 2 + 17] ^= 0xFF
    p.write_bytes(bytes(raw))
    rep = decompress_archive(p, True, tmp_path / "o", backend=backend)
    assert (rep.total_files, rep.corrupt_files, rep.verified_files) == (5, 1, 4)
    assert rep.corrupt_bytes == 3000 and rep.verified_bytes == 12000 and rep.corrupt_rows == [2]
    assert len((tmp_path / "o" / "f2.png").read_bytes()) == 3000


def test_chunking_rules():
    """stream_packer.rs:L169-202: empty entry -> one zero-length round; <= 8 MiB -> one round;
    bigger -> 8 MiB rounds with fdata_offset = byte offset and chunk_seq = 0,1,2..."""
    e = [ArchiveEntry("a", b""), ArchiveEntry("b", b"x" * 10), ArchiveEntry("c.jar", b"y" * 5)]
    rounds, (uf, ub, cf, cb) = plan_rounds(e, False)
    assert rounds == [(0, 0, 0, False, 0, 0), (1, 0, 10, False, 0, 0), (2, 0, 5, True, 0, 0)]
    assert (uf, ub, cf, cb) == (1, 5, 2, 10)

    big = ArchiveEntry("big.bin", bytes(2 * SLICE_SIZE + 1))
    rounds, _ = plan_rounds([big], False)
    assert rounds == [(0, 0, SLICE_SIZE, False, 0, 0), (0, SLICE_SIZE, SLICE_SIZE, False, SLICE_SIZE, 1),
                      (0, 2 * SLICE_SIZE, 1, False, 2 * SLICE_SIZE, 2)]
    exact = plan_rounds([ArchiveEntry("e", bytes(SLICE_SIZE))], False)[0]
    assert exact == [(0, 0, SLICE_SIZE, False, 0, 0)]


def test_skip_extension_list():  # index.rs:L470-488
    for name in ("a.jar", "B.PNG", "x/y.tar.gz", "c.crate", "d.znippy", "e.parquet"):
        assert ix.should_skip_compression(name), name
    for name in ("a.txt", "noext", ".gz", "a.gz.txt", "dir.zip/readme", "pom.xml", "random.bin"):
        assert not ix.should_skip_compression(name), name


# ─── Directory compressor tests (integration_test.rs:L228-354) ─────────────────────────────────
def _mk_tree(root, files):
    for rel, data in files.items():
        p = root / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_bytes(data)


def test_compress_dir_basic_and_mixed(backend, tmp_path, tmp_path_factory):  # L228-299
    from znippy_amd.slot_packer import compress_dir
    src = tmp_path / "in"
    files = {"a.txt": b"alpha " * 100, "sub/b.txt": b"beta " * 300, "sub/deep/c.bin": gen.binary(5000),
             "img.png": gen.incompressible(1, 2000), "empty.txt": b""}
    _mk_tree(src, files)
    report = compress_dir(src, tmp_path / "out.znippy", False, backend=backend)
    assert report.total_files == 5 and report.uncompressed_files == 1 and report.compressed_files == 4
    assert report.chunks == 5 and report.total_dirs == 3
    out = decompress_to_map(tmp_path / "out.znippy", backend, tmp_path_factory)
    assert out == files
    _, batches = ix.read_znippy_index(str(tmp_path / "out.znippy"))
    rows = {p: (s, f) for p, s, f in zip(batches[0].column(0).to_pylist(), batches[0].column(1).to_pylist(),
                                         batches[0].column(2).to_pylist())}
    assert all(v == (0, 0) for v in rows.values())      # small files: chunk_seq = fdata_offset = 0 (L499)


def test_compress_dir_slots_round_trip(backend, tmp_path, tmp_path_factory):  # L301-354
    from znippy_amd import index as ix2
    from znippy_amd.slot_packer import compress_dir
    src = tmp_path / "in"
    files = {f"small/f{i:02}.txt": gen.pseudo_text(500 + 37 * i, seed=i) for i in range(50)}
    files["nested/a/b/c.txt"] = b"nested"
    files["archive.gz"] = gen.incompressible(3, 4096)
    files["empty.dat"] = b""
    files["big.bin"] = gen.binary(8 * 1024 * 1024)
    _mk_tree(src, files)
    cfg = ix2.StrategicConfig(max_core_in_flight=64)   # slice_size = 200 MiB / 64 = 3.125 MiB -> big.bin is cut
    report = compress_dir(src, tmp_path / "slots", False, repo="r", backend=backend, config=cfg)
    assert report.total_files == len(files)
    assert report.chunks == len(files) - 1 + 3          # big.bin -> 3 slices
    out = decompress_to_map(tmp_path / "slots.znippy", backend, tmp_path_factory)
    assert out == files
    m = ix.read_znippy_manifest(str(tmp_path / "slots.znippy"))
    assert len(m) == 1 and (m[0].pkg_type, m[0].repo, m[0].row_count) == (0, "r", report.chunks)


def test_pipelines_hand_the_configured_level_to_the_codec(backend, tmp_path):
    """CompressCtx::new(CONFIG.compression_level) (stream_packer.rs:L217, slot_packer.rs:L551): both pipelines set the
    backend's level from the config they run with, and the level is what the archive's metadata records."""
    import copy
    from znippy_amd.slot_packer import compress_dir
    seen = []
    real = backend.set_level
    backend.set_level = lambda lv: (seen.append(lv), real(lv))[1]
    try:
        cfg = copy.copy(ix.CONFIG)
        cfg.compression_level = 3
        sc = compress_stream(str(tmp_path / "lv"), False, backend=backend, config=cfg)
        sc.sender().send(ArchiveEntry("a.txt", gen.pseudo_text(50_000, seed=3)))
        sc.finish()
        schema, _ = ix.read_znippy_index(str(tmp_path / "lv.znippy"))
        assert {k.decode(): v.decode() for k, v in schema.metadata.items()}["compression_level"] == "3"
        src = tmp_path / "in"
        src.mkdir()
        (src / "b.txt").write_bytes(gen.pseudo_text(20_000, seed=4))
        compress_dir(str(src), str(tmp_path / "dir"), backend=backend, config=cfg)
        assert seen == [3, 3]
        out = decompress_archive(str(tmp_path / "lv.znippy"), True, tmp_path / "o", backend=backend)
        assert out.corrupt_files == 0 and (tmp_path / "o" / "a.txt").read_bytes() == gen.pseudo_text(50_000, seed=3)
    finally:
        backend.set_level = real
        backend.set_level(19)
