"""The compiled host layer (csrc/host/*.cpp, include/znippy_host.h).

CPU part (-m "not gpu"): the hand-written Arrow IPC writer/reader against pyarrow — what C++ writes
pyarrow must read back identically (schema, metadata, values) and what pyarrow writes C++ must read.
GPU part: compress_stream / decompress_archive / ZnippyArchive through the C ABI, cross-checked with
the Python mirror (pyarrow index I/O) in both directions.
"""
import os
import re
import struct

import numpy as np
import pyarrow as pa
import pytest

import gen
from znippy_amd import index as ix


def test_header_symbols_exported():
    import __graft_entry__ as g
    g.build()
    from znippy_amd import host
    L = host.lib()
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "znippy_host.h")).read()
    declared = set(re.findall(r"\b(znippy_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(host.HOST_EXPORTS), declared ^ set(host.HOST_EXPORTS)
    for n in declared:
        assert hasattr(L, n), n


def test_manifest_written_by_cpp_is_read_by_pyarrow_and_back():
    from znippy_amd import host
    entries = [ix.ManifestEntry(1, "central", "maven", 0, 1024, 42), ix.ManifestEntry(2, "crates-io", "cargo", 1024, 512, 17),
               ix.ManifestEntry(-3, "", "", 2**40, 7, 0)]
    b = host.write_manifest_bytes(entries)
    assert ix.read_manifest_bytes(b) == entries                      # pyarrow parses the C++ stream
    r = pa.ipc.open_stream(pa.py_buffer(b))
    assert r.schema.equals(ix.manifest_schema())
    assert host.write_manifest_bytes([]) and ix.read_manifest_bytes(host.write_manifest_bytes([])) == []
    # framing per the IPC spec / arrow-rs defaults: continuation marker, 64-byte aligned messages, EOS marker
    assert b[:4] == b"\xff\xff\xff\xff" and b[-8:] == b"\xff\xff\xff\xff\x00\x00\x00\x00"
    meta_len = struct.unpack("<i", b[4:8])[0]
    assert (8 + meta_len) % 64 == 0


def test_interpret_footer_cpp():
    from znippy_amd import host
    assert host.interpret_footer(struct.pack("<Q", 12345)) == ("single", 12345)
    assert host.interpret_footer(ix.MULTI_INDEX_MAGIC + struct.pack("<Q", 99999)) == ("multi", 99999)


def test_cpp_index_reader_reads_pyarrow_written_archive(tmp_path, oracle):
    """Archive written by the Python mirror (pyarrow IPC) with the checker backend, read by the C++ reader."""
    from oracle_backend import OracleBackend
    from znippy_amd import host
    from znippy_amd.stream_packer import ArchiveEntry, compress_stream
    c = compress_stream(tmp_path / "a.znippy", False, backend=OracleBackend())
    ents = [ArchiveEntry("pom.xml", b"<project/>" * 50, 1, "maven"), ArchiveEntry("lib.jar", gen.incompressible(1, 5000), 1, "maven"),
            ArchiveEntry("Cargo.toml", b"[package]", 2, "cargo"), ArchiveEntry("empty", b""),
            ArchiveEntry("big.bin", gen.binary(9 * 1024 * 1024))]
    for e in ents:
        c.sender().send(e)
    c.finish()
    rows, manifest, md = host.read_index(tmp_path / "a.znippy")
    _, batches = ix.read_znippy_index(str(tmp_path / "a.znippy"))
    want = batches[0].to_pylist()
    assert len(rows) == len(want) == 6
    for a, b in zip(rows, want):
        assert a == b
    assert manifest == ix.read_znippy_manifest(str(tmp_path / "a.znippy"))
    assert md["znippy_format_version"] == "3" and "compression_level" in md and "checksum_group_0" not in md


def test_cpp_index_reader_rejects_malformed_containers_with_status_codes(tmp_path, oracle):
    """A file is untrusted input: wrong manifest schema, lengths that do not fit the file, truncated streams and
    absurd sizes all come back as status codes from the C ABI (no fault, no C++ exception across the boundary)."""
    import pyarrow as pa
    from oracle_backend import OracleBackend
    from znippy_amd import host
    from znippy_amd.stream_packer import ArchiveEntry, compress_stream
    c = compress_stream(tmp_path / "a.znippy", False, backend=OracleBackend())
    for i in range(5):
        c.sender().send(ArchiveEntry(f"f{i}.txt", gen.pseudo_text(4000 + i, seed=i)))
    c.finish()
    good = (tmp_path / "a.znippy").read_bytes()
    rows, manifest, _ = host.read_index(tmp_path / "a.znippy")
    assert len(rows) == 5
    moff = struct.unpack("<Q", good[-8:])[0]

    def expect_error(blob, name):
        f = tmp_path / name
        f.write_bytes(blob)
        with pytest.raises(host.HostError):
            host.read_index(f)

    # (1) manifest whose columns have other types (strings where the offsets should be)
    sink = pa.BufferOutputStream()
    bad_schema = pa.schema([pa.field(n, pa.utf8(), nullable=False) for n in
                            ("pkg_type", "repo", "module_name", "index_offset", "index_len", "row_count")])
    with pa.ipc.new_stream(sink, bad_schema) as w:
        w.write_batch(pa.record_batch([pa.array(["x"])] * 6, schema=bad_schema))
    mb = sink.getvalue().to_pybytes()
    expect_error(good[:moff] + mb + ix.MULTI_INDEX_MAGIC + struct.pack("<Q", moff), "badschema.znippy")
    # (2) sub-index length far beyond the file, and offset + length wrapping around 2^64
    for off, ln in ((manifest[0].index_offset, 1 << 60), ((1 << 64) - 8, 64), (len(good) + 5, 1)):
        m2 = [ix.ManifestEntry(manifest[0].pkg_type, manifest[0].repo, manifest[0].module_name, off, ln, manifest[0].row_count)]
        mb = host.write_manifest_bytes(m2)
        expect_error(good[:moff] + mb + ix.MULTI_INDEX_MAGIC + struct.pack("<Q", moff), f"badlen_{ln}.znippy")
    # (3) manifest offset beyond the file / garbage where the manifest should be / file cut in the middle
    expect_error(good[:-8] + struct.pack("<Q", len(good) + 100), "badmoff.znippy")
    expect_error(good[:moff] + b"\x00" * 64 + ix.MULTI_INDEX_MAGIC + struct.pack("<Q", moff), "garbage.znippy")
    expect_error(good[:moff + 40] + ix.MULTI_INDEX_MAGIC + struct.pack("<Q", moff), "cut.znippy")
    expect_error(b"tiny", "tiny.znippy")
    # (4) sub-index bytes damaged: flatbuffer offsets pointing anywhere
    sub0 = manifest[0].index_offset
    rng = np.random.default_rng(5)
    for _ in range(20):
        b = bytearray(good)
        for k in rng.integers(sub0 + 8, sub0 + 300, size=6):
            b[int(k)] = int(rng.integers(0, 256))
        f = tmp_path / "fuzz.znippy"
        f.write_bytes(bytes(b))
        try:
            host.read_index(f)       # either it still parses or it is an error code; never a crash
        except host.HostError:
            pass


gpu = pytest.mark.gpu


@gpu
def test_cpp_compress_stream_roundtrip_and_cross_read(tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from znippy_amd import host
    from znippy_amd.decompress import decompress_archive as py_decompress
    from znippy_amd.stream_packer import ArchiveEntry
    ents = [ArchiveEntry("hello.txt", b"Hello, Znippy! This is a small test file."),
            ArchiveEntry("sub/dir/text.txt", gen.pseudo_text(200000, seed=4)), ArchiveEntry("image.png", gen.incompressible(2, 30000)),
            ArchiveEntry("empty.txt", b""), ArchiveEntry("large.bin", gen.binary(12 * 1024 * 1024)),
            ArchiveEntry("pom.xml", gen.text(10240), 1, "maven"), ArchiveEntry("Cargo.toml", b"[package]", 2, "cargo")]
    c = host.compress_stream(tmp_path / "cpp.tmp", False)
    for e in ents:
        c.sender().send(e)
    rep = c.finish()
    archive = tmp_path / "cpp.znippy"                                      # extension forced (stream_packer.rs:L132)
    assert archive.exists()
    assert (rep.total_files, rep.uncompressed_files, rep.compressed_files) == (7, 1, 6)
    assert rep.chunks == 8 and rep.total_bytes_in == sum(len(e.data) for e in ents)
    assert rep.total_bytes_out == archive.stat().st_size
    # pyarrow reads what C++ wrote: schema, metadata, groups, rows
    schema, batches = ix.read_znippy_index(str(archive))
    assert schema.names == ix.znippy_index_schema().names
    assert all(not f.nullable for f in schema)
    md = {k.decode(): v.decode() for k, v in schema.metadata.items()}
    assert md["znippy_format_version"] == "3" and len(md) == 9 and "compression_level" in md
    manifest = ix.read_znippy_manifest(str(archive))
    assert [(m.pkg_type, m.repo, m.row_count) for m in manifest] == [(0, "", 6), (1, "maven", 1), (2, "cargo", 1)]
    raw = archive.read_bytes()
    assert raw[-16:-8] == b"ZNPYMIDX"
    # the Python mirror (pyarrow + HIP backend) decodes the C++-written archive
    rep_py = py_decompress(archive, True, tmp_path / "out_py")
    assert (rep_py.corrupt_files, rep_py.total_files, rep_py.chunks) == (0, 7, 8)
    for e in ents:
        assert (tmp_path / "out_py" / e.relative_path).read_bytes() == e.data
    # and the C++ reader decodes it too; reports agree
    rep_cpp = host.decompress_archive(archive, True, tmp_path / "out_cpp")
    assert rep_cpp == rep_py
    for e in ents:
        assert (tmp_path / "out_cpp" / e.relative_path).read_bytes() == e.data
    # verify-only path and per-rank split
    v = host.decompress_archive(archive, False, "/dev/null")
    assert (v.total_files, v.corrupt_files, v.total_bytes) == (7, 0, rep.total_bytes_in)
    parts = [host.decompress_archive(archive, False, "/dev/null", rank=r, world=3) for r in range(3)]
    assert sum(p.chunks for p in parts) == 8 and sum(p.total_bytes for p in parts) == rep.total_bytes_in


@gpu
def test_cpp_stream_send_packed_writes_the_same_archive(tmp_path):
    """n entries in ONE call (znippy_stream_send_packed) = the archive n send() calls write, byte for byte; offsets
    that decrease are an error code."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from znippy_amd import host
    from znippy_amd.stream_packer import ArchiveEntry
    datas = [gen.pseudo_text(3000 + 977 * i, seed=i) if i % 5 else b"" for i in range(60)] + [gen.binary(9 * 1024 * 1024 + 17)]
    names = [f"d{i % 4}/f{i:03}.txt" for i in range(60)] + ["big/blob.bin"]
    a = host.compress_stream(tmp_path / "one.tmp", False)
    for nm, d in zip(names, datas):
        a.send(ArchiveEntry(nm, d))
    ra = a.finish()
    b = host.compress_stream(tmp_path / "packed.tmp", False)
    offs = np.concatenate([[0], np.cumsum([len(d) for d in datas])]).astype(np.uint64)
    b.send_packed(names[:20], b"".join(datas), offs[:21])
    b.send_packed(names[20:], b"".join(datas), offs[20:])
    rb = b.finish()
    assert ra == rb
    assert (tmp_path / "one.znippy").read_bytes() == (tmp_path / "packed.znippy").read_bytes()
    v = host.decompress_archive(tmp_path / "packed.znippy", True, tmp_path / "out")
    assert (v.corrupt_files, v.total_files) == (0, len(names))
    for nm, d in zip(names, datas):
        assert (tmp_path / "out" / nm).read_bytes() == d
    c = host.compress_stream(tmp_path / "bad.tmp", False)
    with pytest.raises(Exception):
        c.send_packed(["x", "y"], b"abcdef", np.array([0, 4, 2], np.uint64))
    c.finish()


@gpu
def test_cpp_reads_python_written_archive_and_random_access(tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from znippy_amd import host
    from znippy_amd.stream_packer import ArchiveEntry, compress_stream
    c = compress_stream(tmp_path / "py.znippy", False)
    data = {"big.bin": gen.binary(12 * 1024 * 1024), "a/b.txt": gen.pseudo_text(5000, 2), "z.jar": gen.incompressible(5, 70000),
            "e": b""}
    for k, v in data.items():
        c.sender().send(ArchiveEntry(k, v))
    c.finish()
    rep = host.decompress_archive(tmp_path / "py.znippy", True, tmp_path / "o")
    assert (rep.total_files, rep.corrupt_files, rep.verified_files) == (4, 0, 4)
    for k, v in data.items():
        assert (tmp_path / "o" / k).read_bytes() == v
    a = host.ZnippyArchive.open(tmp_path / "py.znippy")
    assert a.file_count() == 4 and a.contains("big.bin") and not a.contains("nope")
    assert a.extract_file("big.bin") == data["big.bin"]                      # multi-chunk order (integration_test.rs:L617-642)
    assert a.extract_file("z.jar") == data["z.jar"] and a.extract_file("e") == b""
    with pytest.raises(KeyError):
        a.extract_file("nope")
    # corruption is counted, not fatal
    raw = bytearray((tmp_path / "py.znippy").read_bytes())
    i = bytes(raw).find(data["z.jar"][:64])
    raw[i + 9] ^= 0x40
    (tmp_path / "bad.znippy").write_bytes(bytes(raw))
    bad = host.decompress_archive(tmp_path / "bad.znippy", True, tmp_path / "o2")
    assert (bad.corrupt_files, bad.verified_files, bad.corrupt_bytes) == (1, 3, 70000) and len(bad.corrupt_rows) == 1


def _tree(root, files):
    for rel, data in files.items():
        p = root / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_bytes(data)


@gpu
def test_cpp_pipeline_many_slots_ranges_and_files(tmp_path, monkeypatch):
    """Small staging slots / decoded ranges force every pipeline hand-off: rounds spilling across slots, a
    multi-chunk file continuing across ranges (first-touch create before later rows), thousands of output
    files with one cached descriptor per writer, and a duplicated relative path (single-writer fallback)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from znippy_amd import host
    from znippy_amd.stream_packer import ArchiveEntry, compress_stream as py_compress_stream
    monkeypatch.setenv("ZNIPPY_HOST_SLOT_MB", "1")      # slots grow to one 8 MiB round at most
    monkeypatch.setenv("ZNIPPY_HOST_RANGE_MB", "1")
    rng = np.random.default_rng(5)
    data = {f"d{i % 7}/f{i:05}.txt": gen.pseudo_text(int(rng.integers(0, 3000)), seed=i) for i in range(3000)}
    data["big/a.bin"] = gen.binary(20 * 1024 * 1024 + 17)       # 3 rounds of <= 8 MiB
    data["big/b.jar"] = gen.incompressible(3, 9 * 1024 * 1024)  # stored, 2 rounds
    data["zz/last.txt"] = gen.text(10240)
    ents = [ArchiveEntry(k, v) for k, v in data.items()]
    c = host.compress_stream(tmp_path / "a.znippy", False)
    for e in ents:
        c.send(e)
    rep = c.finish()
    cp = py_compress_stream(tmp_path / "p.znippy", False)
    for e in ents:
        cp.sender().send(e)
    rep_py = cp.finish()
    # slot boundaries do not show in the archive: byte-identical blob region + rows vs the one-batch Python mirror
    assert rep.chunks == rep_py.chunks == 3000 + 3 + 2 + 1
    a, p = (tmp_path / "a.znippy").read_bytes(), (tmp_path / "p.znippy").read_bytes()
    ia, ip = host.read_index(tmp_path / "a.znippy")[0], host.read_index(tmp_path / "p.znippy")[0]
    end = max(r["blob_offset"] + r["blob_size"] for r in ia)
    assert a[:end] == p[:end]
    assert ia == ip
    out = tmp_path / "out"
    v = host.decompress_archive(tmp_path / "a.znippy", True, out)
    assert (v.total_files, v.corrupt_files, v.chunks) == (len(data), 0, rep.chunks)
    for k, want in data.items():
        assert (out / k).read_bytes() == want, k
    # pre-existing longer files are truncated on first touch (File::create, decompress.rs:L86)
    (out / "zz/last.txt").write_bytes(b"x" * 50000)
    host.decompress_archive(tmp_path / "a.znippy", True, out)
    assert (out / "zz/last.txt").read_bytes() == data["zz/last.txt"]
    # duplicated relative path: later entry overlays the earlier one, as positioned writes into one file do
    c = host.compress_stream(tmp_path / "dup.znippy", False)
    for k, vbytes in (("x.txt", b"A" * 5000), ("y.txt", b"yy"), ("x.txt", b"B" * 100)):
        c.send(ArchiveEntry(k, vbytes))
    c.finish()
    host.decompress_archive(tmp_path / "dup.znippy", True, tmp_path / "dup")
    assert (tmp_path / "dup/x.txt").read_bytes() == b"B" * 100 + b"A" * 4900


@gpu
def test_cpp_compress_dir_matches_python_mirror(tmp_path, monkeypatch):
    """compress_dir (slot_packer.rs:L55-209) on the compiled host: same rows, blob bytes and report as the
    Python mirror; archive restores the tree."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from znippy_amd import host
    from znippy_amd.slot_packer import compress_dir as py_compress_dir, SLOT_SIZE
    monkeypatch.setenv("ZNIPPY_HOST_SLOT_MB", "4")
    slice_size = SLOT_SIZE // max(ix.CONFIG.max_core_in_flight, 1)
    files = {f"src/m{i % 5}/f{i:04}.rs": gen.pseudo_text(200 + 37 * i, seed=i) for i in range(400)}
    files.update({"README.md": gen.text(10240), "empty.txt": b"", "assets/logo.png": gen.incompressible(1, 40000),
                  "assets/deep/er/data.bin": gen.binary(2 * slice_size + 12345),      # big pass: 3 rounds
                  "assets/pack.zip": gen.incompressible(9, slice_size + 5)})         # big + stored: 2 rounds
    root = tmp_path / "tree"
    _tree(root, files)
    (root / "emptydir").mkdir()
    rep = host.compress_dir(root, tmp_path / "c.znippy", False, repo="myrepo")
    rep_py = py_compress_dir(root, tmp_path / "p.znippy", False, repo="myrepo")
    import dataclasses
    da, dp = dataclasses.asdict(rep), dataclasses.asdict(rep_py)
    for k in da:   # the metadata layer is written by two Arrow IPC writers (64- vs 8-byte buffer alignment); ratio is f32 here
        if k not in ("total_bytes_out", "compression_ratio"):
            assert da[k] == dp[k], k
    assert abs(rep.compression_ratio - rep_py.compression_ratio) < 1e-3 * max(rep_py.compression_ratio, 1)
    assert rep.total_files == len(files) and rep.chunks == len(files) + 2 + 1 and rep.total_dirs == rep_py.total_dirs
    a, p = (tmp_path / "c.znippy").read_bytes(), (tmp_path / "p.znippy").read_bytes()
    assert len(a) == rep.total_bytes_out and len(p) == rep_py.total_bytes_out
    ia, ip = host.read_index(tmp_path / "c.znippy")[0], host.read_index(tmp_path / "p.znippy")[0]
    assert ia == ip
    end = max(r["blob_offset"] + r["blob_size"] for r in ia)
    assert a[:end] == p[:end]
    m = ix.read_znippy_manifest(str(tmp_path / "c.znippy"))
    assert [(e.pkg_type, e.repo, e.row_count) for e in m] == [(0, "myrepo", rep.chunks)]
    # pyarrow sees two record batches (big pass, small pass) in the one sub-index
    sub = a[m[0].index_offset:m[0].index_offset + m[0].index_len]
    assert [b.num_rows for b in pa.ipc.open_stream(sub)] == [1 + 3 + 2, len(files) - 3]
    out = tmp_path / "out"
    v = host.decompress_archive(tmp_path / "c.znippy", True, out)
    assert (v.total_files, v.corrupt_files) == (len(files), 0)
    for k, want in files.items():
        assert (out / k).read_bytes() == want, k
    # an empty directory still yields a readable archive with zero rows
    (tmp_path / "nothing").mkdir()
    r0 = host.compress_dir(tmp_path / "nothing", tmp_path / "n.znippy")
    assert (r0.total_files, r0.chunks, r0.total_dirs) == (0, 0, 1)
    assert host.decompress_archive(tmp_path / "n.znippy", False, "/dev/null").total_files == 0


@gpu
def test_config_c1_plumbing_1000_files(tmp_path):
    """1,000 files x 10,240 B of generate_text_data named d000/file_000000.txt ... through the
    directory entry point: compress -> decompress -> byte compare, corrupt_files == 0 (compiled host and Python
    mirror write interchangeable archives)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from znippy_amd import host
    from znippy_amd.decompress import decompress_archive as py_decompress
    from znippy_amd.slot_packer import compress_dir as py_compress_dir
    chunk = gen.text(10 * 1024)
    root = tmp_path / "in"
    files = {f"d{i // 100:03}/file_{i:06}.txt": chunk for i in range(1000)}
    _tree(root, files)
    rep = host.compress_dir(root, tmp_path / "c1.znippy")
    assert (rep.total_files, rep.chunks, rep.total_dirs, rep.total_bytes_in) == (1000, 1000, 11, 1000 * 10240)
    assert rep.total_bytes_out < 400_000                       # ~85 B of frame + ~130 B of index per file
    v = host.decompress_archive(tmp_path / "c1.znippy", True, tmp_path / "out")
    assert (v.total_files, v.verified_files, v.corrupt_files, v.total_bytes, v.chunks) == (1000, 1000, 0, 1000 * 10240, 1000)
    for rel in files:
        assert (tmp_path / "out" / rel).read_bytes() == chunk, rel
    # the other implementation of the same interface, both directions
    py_compress_dir(root, tmp_path / "c1_py.znippy")
    assert host.decompress_archive(tmp_path / "c1_py.znippy", False, "/dev/null").corrupt_files == 0
    assert py_decompress(tmp_path / "c1.znippy", False, None).corrupt_files == 0
