"""Container / index codec of the Znippy archive format (host side).

Mirrors znippy-common/src/index.rs + meta_sink.rs + meta.rs + common_config.rs:
  file layout   [blobs][Arrow-IPC sub-index ...][manifest stream]["ZNPYMIDX"][LE u64 manifest_offset]
                (index.rs:L232-245, meta_sink.rs:L103-118)
  base schema   8 non-null columns in this order (index.rs:L43-54)
  schema meta   9 config keys, znippy_format_version = "3" (index.rs:L73-85)
  manifest      pkg_type Int8, repo Utf8, module_name Utf8, index_offset/index_len/row_count UInt64
                (index.rs:L279-288)
Arrow IPC (de)serialisation is pyarrow's here; everything else is restated.
"""
import os
import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import pyarrow as pa

MULTI_INDEX_MAGIC = b"ZNPYMIDX"  # index.rs:L245

_SKIP_EXT = {  # is_probably_compressed, index.rs:L470-484 (last extension, case-insensitive)
    "zip", "gz", "bz2", "xz", "lz", "lzma", "7z", "rar", "cab", "jar", "war", "ear", "zst", "sz", "lz4", "tgz",
    "txz", "tbz", "apk", "dmg", "deb", "rpm", "arrow", "mpeg", "mpg", "jpeg", "jpg", "gif", "bmp", "png", "crate",
    "znippy", "zdata", "parquet", "webp", "webm",
}


def is_probably_compressed(path: str) -> bool:
    # std::path::Path::extension(): text after the last '.' of the file name, none if the name
    # starts with '.' and has no other dot
    name = os.path.basename(path.rstrip("/"))
    if name.startswith("."):
        stem = name[1:]
        if "." not in stem:
            return False
    if "." not in name:
        return False
    ext = name.rsplit(".", 1)[1]
    if not ext:
        return False
    return ext.lower() in _SKIP_EXT


def should_skip_compression(path: str) -> bool:  # index.rs:L486-488
    return is_probably_compressed(path)


def base_index_fields() -> List[pa.Field]:  # index.rs:L43-54
    return [
        pa.field("relative_path", pa.utf8(), nullable=False),
        pa.field("chunk_seq", pa.uint32(), nullable=False),
        pa.field("fdata_offset", pa.uint64(), nullable=False),
        pa.field("compressed", pa.bool_(), nullable=False),
        pa.field("uncompressed_size", pa.uint64(), nullable=False),
        pa.field("blob_offset", pa.uint64(), nullable=False),
        pa.field("blob_size", pa.uint64(), nullable=False),
        pa.field("checksum", pa.binary(32), nullable=False),
    ]


def znippy_index_schema() -> pa.Schema:
    return pa.schema(base_index_fields())


def compose_index_schema(ext_fields=()) -> pa.Schema:  # index.rs:L63-70
    fields = base_index_fields()
    if ext_fields:
        fields.append(pa.field("pkg_type", pa.int8(), nullable=True))
        fields.extend(ext_fields)
    return pa.schema(fields)


@dataclass
class StrategicConfig:  # common_config.rs:L11-21
    max_core_allowed: int = 0
    max_core_in_flight: int = 1
    max_core_in_compress: int = 0
    max_mem_allowed: int = 0
    min_free_memory_ratio: float = 0.0
    file_split_block_size: int = 10 * 1024 * 1024
    max_chunks: int = 128
    compression_level: int = 19
    zstd_output_buffer_size: int = 1024 * 1024


def strategic_config() -> StrategicConfig:
    """CONFIG (common_config.rs:L23-78): workers = ceil(0.9 x cores), level 19, legacy fields kept
    only because they are serialised into the schema metadata."""
    import math
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        total_mem = os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES")
    except (ValueError, OSError):
        total_mem = 0
    in_flight = int(math.ceil(cores * 0.90))
    return StrategicConfig(max_core_allowed=cores, max_core_in_flight=in_flight,
                           max_core_in_compress=max(cores - in_flight, 0), max_mem_allowed=total_mem,
                           min_free_memory_ratio=0.0, max_chunks=min(total_mem // (10 * 1024 * 1024), 128))


CONFIG = strategic_config()


def _fmt_f32(x: float) -> str:
    s = repr(float(np.float32(x)))
    return s[:-2] if s.endswith(".0") else s  # Rust prints 0 for 0.0f32


def build_arrow_metadata_for_config(cfg: StrategicConfig) -> Dict[str, str]:  # index.rs:L73-85
    return {
        "znippy_format_version": "3",
        "max_core_in_flight": str(cfg.max_core_in_flight),
        "max_core_in_compress": str(cfg.max_core_in_compress),
        "max_mem_allowed": str(cfg.max_mem_allowed),
        "min_free_memory_ratio": _fmt_f32(cfg.min_free_memory_ratio),
        "file_split_block_size": str(cfg.file_split_block_size),
        "max_chunks": str(cfg.max_chunks),
        "compression_level": str(cfg.compression_level),
        "zstd_output_buffer_size": str(cfg.zstd_output_buffer_size),
    }


@dataclass
class ChunkMeta:  # meta.rs:L4-13
    fdata_offset: int
    file_index: int
    chunk_seq: int
    checksum: bytes
    compressed: bool
    uncompressed_size: int
    compressed_size: int


@dataclass
class BlobMeta:  # meta.rs:L17-21
    chunk_meta: ChunkMeta
    blob_offset: int
    blob_size: int


def build_metadata_batch(blobs: List[BlobMeta], path_resolver) -> pa.RecordBatch:  # index.rs:L131-191 (no ext fields)
    ck = np.frombuffer(b"".join(b.chunk_meta.checksum for b in blobs), dtype=np.uint8) if blobs else np.zeros(0, np.uint8)
    cols = [
        pa.array([path_resolver(b.chunk_meta.file_index) for b in blobs], type=pa.utf8()),
        pa.array([b.chunk_meta.chunk_seq for b in blobs], type=pa.uint32()),
        pa.array([b.chunk_meta.fdata_offset for b in blobs], type=pa.uint64()),
        pa.array([b.chunk_meta.compressed for b in blobs], type=pa.bool_()),
        pa.array([b.chunk_meta.uncompressed_size for b in blobs], type=pa.uint64()),
        pa.array([b.blob_offset for b in blobs], type=pa.uint64()),
        pa.array([b.blob_size for b in blobs], type=pa.uint64()),
        pa.FixedSizeBinaryArray.from_buffers(pa.binary(32), len(blobs), [None, pa.py_buffer(ck.tobytes())]),
    ]
    return pa.RecordBatch.from_arrays(cols, schema=compose_index_schema())


def batch_from_columns(paths, chunk_seq, fdata_offset, compressed, usize, blob_offset, blob_size, checksum):
    """Vectorised twin of build_metadata_batch for large archives (numpy columns)."""
    n = len(paths)
    ck = np.ascontiguousarray(checksum, dtype=np.uint8).reshape(-1)
    cols = [
        pa.array(paths, type=pa.utf8()),
        pa.array(np.asarray(chunk_seq, dtype=np.uint32)),
        pa.array(np.asarray(fdata_offset, dtype=np.uint64)),
        pa.array(np.asarray(compressed, dtype=bool)),
        pa.array(np.asarray(usize, dtype=np.uint64)),
        pa.array(np.asarray(blob_offset, dtype=np.uint64)),
        pa.array(np.asarray(blob_size, dtype=np.uint64)),
        pa.FixedSizeBinaryArray.from_buffers(pa.binary(32), n, [None, pa.py_buffer(ck.tobytes())]),
    ]
    return pa.RecordBatch.from_arrays(cols, schema=compose_index_schema())


# ---- multi-index container (v0.7) ---------------------------------------------------------------
@dataclass
class ManifestEntry:  # index.rs:L248-256
    pkg_type: int
    repo: str
    module_name: str
    index_offset: int
    index_len: int
    row_count: int


def interpret_footer(tail: bytes):  # index.rs:L269-277
    """-> ("multi", manifest_offset) | ("single", index_offset)"""
    n = len(tail)
    offset = struct.unpack("<Q", tail[n - 8:])[0]
    if n >= 16 and tail[n - 16:n - 8] == MULTI_INDEX_MAGIC:
        return ("multi", offset)
    return ("single", offset)


def manifest_schema() -> pa.Schema:  # index.rs:L279-288
    return pa.schema([
        pa.field("pkg_type", pa.int8(), nullable=False),
        pa.field("repo", pa.utf8(), nullable=False),
        pa.field("module_name", pa.utf8(), nullable=False),
        pa.field("index_offset", pa.uint64(), nullable=False),
        pa.field("index_len", pa.uint64(), nullable=False),
        pa.field("row_count", pa.uint64(), nullable=False),
    ])


def _ipc_stream_bytes(schema: pa.Schema, batches) -> bytes:
    sink = pa.BufferOutputStream()
    with pa.ipc.new_stream(sink, schema) as w:  # Arrow IPC *stream* (StreamWriter in the reference)
        for b in batches:
            w.write_batch(b)
    return sink.getvalue().to_pybytes()


def write_manifest_bytes(entries: List[ManifestEntry]) -> bytes:  # index.rs:L291-330
    schema = manifest_schema()
    batch = pa.RecordBatch.from_arrays([
        pa.array([e.pkg_type for e in entries], type=pa.int8()),
        pa.array([e.repo for e in entries], type=pa.utf8()),
        pa.array([e.module_name for e in entries], type=pa.utf8()),
        pa.array([e.index_offset for e in entries], type=pa.uint64()),
        pa.array([e.index_len for e in entries], type=pa.uint64()),
        pa.array([e.row_count for e in entries], type=pa.uint64()),
    ], schema=schema)
    return _ipc_stream_bytes(schema, [batch])


def read_manifest_bytes(data: bytes) -> List[ManifestEntry]:  # index.rs:L333-367
    out = []
    reader = pa.ipc.open_stream(pa.py_buffer(data))
    for batch in reader:
        for name in ("pkg_type", "repo", "module_name", "index_offset", "index_len", "row_count"):
            if name not in batch.schema.names:
                raise ValueError(f"manifest missing column {name}")
        cols = {n: batch.column(batch.schema.names.index(n)).to_pylist() for n in batch.schema.names}
        for i in range(batch.num_rows):
            out.append(ManifestEntry(cols["pkg_type"][i], cols["repo"][i], cols["module_name"][i],
                                     cols["index_offset"][i], cols["index_len"][i], cols["row_count"][i]))
    return out


class ArrowIpcSink:
    """meta_sink.rs:L52-118 — places sub-indexes after the blob region, then manifest + footer."""

    def __init__(self, file, blob_end_offset: int):
        self.file = file
        self.cursor = blob_end_offset
        self.entries: List[ManifestEntry] = []

    def push_subindex(self, schema: pa.Schema, batches, pkg_type: int, repo: str, module_name: str = ""):
        sub = _ipc_stream_bytes(schema, batches)
        start = self.cursor
        os.pwrite(self.file.fileno(), sub, start)
        self.cursor += len(sub)
        self.entries.append(ManifestEntry(pkg_type, repo, module_name, start, len(sub),
                                          sum(b.num_rows for b in batches)))

    def finish(self) -> int:
        manifest_offset = self.cursor
        mb = write_manifest_bytes(self.entries)
        fd = self.file.fileno()
        os.pwrite(fd, mb, manifest_offset)
        after = manifest_offset + len(mb)
        os.pwrite(fd, MULTI_INDEX_MAGIC, after)
        os.pwrite(fd, struct.pack("<Q", manifest_offset), after + 8)
        os.fsync(fd)
        return after + 16


def _read_tail(path: str) -> Tuple[int, bytes]:
    size = os.path.getsize(path)
    if size < 16:
        raise ValueError("file too small to be a v0.7 znippy archive")
    with open(path, "rb") as f:
        f.seek(size - 16)
        return size, f.read(16)


def read_znippy_manifest(path: str) -> List[ManifestEntry]:  # index.rs:L443-468
    size, tail = _read_tail(path)
    kind, off = interpret_footer(tail)
    if kind != "multi":
        raise ValueError("not a v0.7 multi-index archive (no MULTI_INDEX_MAGIC)")
    end = size - 16
    if off > end:
        raise ValueError("corrupt manifest_offset")
    with open(path, "rb") as f:
        f.seek(off)
        return read_manifest_bytes(f.read(end - off))


def read_znippy_index(path: str) -> Tuple[pa.Schema, List[pa.RecordBatch]]:  # index.rs:L374-441
    """Footer -> manifest -> every sub-index, merged into ONE batch (callers stay format-agnostic).
    v0.6 single-index archives are rejected like the reference does (L387-389)."""
    size, tail = _read_tail(path)
    kind, off = interpret_footer(tail)
    if kind != "multi":
        raise ValueError("v0.6 archives are not supported; re-compress with v0.7")
    entries = read_znippy_manifest(path)
    batches, schema = [], None
    with open(path, "rb") as f:
        for e in entries:
            f.seek(e.index_offset)
            reader = pa.ipc.open_stream(pa.py_buffer(f.read(e.index_len)))
            if schema is None:
                schema = reader.schema
            batches.extend(list(reader))
    if schema is None:
        schema = znippy_index_schema()
    if len(batches) > 1:
        tbl = pa.Table.from_batches(batches, schema=schema).combine_chunks()
        batches = tbl.to_batches(max_chunksize=None) if tbl.num_rows else []
        if len(batches) > 1:  # combine_chunks normally leaves one chunk
            batches = [pa.concat_batches(batches)] if hasattr(pa, "concat_batches") else batches
    return schema, batches


def blob_region_end(path: str) -> int:
    """First byte after the blob region = smallest sub-index offset (or the manifest offset)."""
    size, tail = _read_tail(path)
    _, moff = interpret_footer(tail)
    entries = read_znippy_manifest(path)
    return min([e.index_offset for e in entries] + [moff])


@dataclass
class VerifyReport:  # index.rs:L490-499
    total_files: int = 0
    verified_files: int = 0
    corrupt_files: int = 0
    total_bytes: int = 0
    verified_bytes: int = 0
    corrupt_bytes: int = 0
    chunks: int = 0
    corrupt_rows: List[int] = field(default_factory=list)  # extra: the rows behind corrupt_files


@dataclass
class CompressionReport:  # lib.rs:L39-51
    total_files: int = 0
    compressed_files: int = 0
    uncompressed_files: int = 0
    total_dirs: int = 0
    total_bytes_in: int = 0
    total_bytes_out: int = 0
    compressed_bytes: int = 0
    uncompressed_bytes: int = 0
    compression_ratio: float = 0.0
    chunks: int = 0


def list_archive_contents(path: str) -> List[Tuple[str, int]]:  # index.rs:L501-548 (returns instead of printing)
    _, batches = read_znippy_index(path)
    out = []
    for b in batches:
        paths = b.column(b.schema.names.index("relative_path")).to_pylist()
        sizes = b.column(b.schema.names.index("uncompressed_size")).to_pylist()
        seqs = b.column(b.schema.names.index("chunk_seq")).to_pylist()
        for p, s, q in zip(paths, sizes, seqs):
            if q == 0:
                out.append((p, s))
    return out
