"""compress_stream — host-side mirror of znippy-compress/src/stream_packer.rs.

Same surface: compress_stream(output, no_skip) -> StreamCompressor{sender(), finish()},
ArchiveEntry{relative_path, data, pkg_type, repo}.  The reader's chunking rules
(stream_packer.rs:L146-206), the writer's blob placement (L255-284) and the finalizer's
sort/group/index layout (L293-346) are restated; the barrels' body (BLAKE3 + encode, L215-248)
runs on the GPU through the backend in batches (the analogue of the Magazine's staging slots).
"""
import os
from dataclasses import dataclass
from typing import List, Optional

import numpy as np

from . import index as ix

SLICE_SIZE = 8 * 1024 * 1024        # stream_packer.rs:L31
BATCH_BYTES = 1 << 30               # staging bytes handed to the GPU at once (8 x 200 MiB Magazine, rounded)


@dataclass
class ArchiveEntry:  # stream_packer.rs:L34-44
    relative_path: str = ""
    data: bytes = b""
    pkg_type: Optional[int] = None
    repo: Optional[str] = None


class _Sender:
    def __init__(self, owner):
        self._owner = owner

    def send(self, entry: ArchiveEntry):
        if self._owner._finished:
            raise RuntimeError("sender already consumed")
        self._owner._entries.append(entry)


class StreamCompressor:  # stream_packer.rs:L58-76
    def __init__(self, output: str, no_skip: bool, backend=None, config=None):
        self._output = output
        self._no_skip = no_skip
        self._backend = backend
        self._config = config or ix.CONFIG
        self._entries: List[ArchiveEntry] = []
        self._finished = False

    def sender(self):
        return _Sender(self)

    def finish(self) -> ix.CompressionReport:
        if self._finished:
            raise RuntimeError("already finished")
        self._finished = True
        return run_pipeline(self._entries, self._output, self._no_skip, self._backend, self._config)


def compress_stream(output: str, no_skip: bool = False, backend=None, config=None) -> StreamCompressor:
    return StreamCompressor(str(output), no_skip, backend, config)


def with_extension(path: str, ext: str) -> str:
    """std::path::Path::with_extension."""
    d, name = os.path.split(path)
    if "." in name and not (name.startswith(".") and name.count(".") == 1):
        name = name.rsplit(".", 1)[0]
    return os.path.join(d, name + "." + ext)


def plan_rounds(entries: List[ArchiveEntry], no_skip: bool):
    """The reader (stream_packer.rs:L146-206): entry -> Rounds.  Returns per-round columns."""
    rounds = []  # (file_index, start, len, skip, fdata_offset, chunk_seq)
    uf = ub = cf = cb = 0
    for file_index, e in enumerate(entries):
        skip = (not no_skip) and ix.should_skip_compression(e.relative_path)
        total = len(e.data)
        if skip:
            uf += 1; ub += total
        else:
            cf += 1; cb += total
        if total == 0:  # empty entry -> one zero-length round so it appears in the index (L169-183)
            rounds.append((file_index, 0, 0, skip, 0, 0))
            continue
        small = total <= SLICE_SIZE
        off = seq = 0
        while off < total:
            ln = total if small else min(SLICE_SIZE, total - off)
            rounds.append((file_index, off, ln, skip, off, seq))
            off += ln
            seq += 1
    return rounds, (uf, ub, cf, cb)


def encode_round_range(rounds, entries, backend, lo, hi):
    """The barrels' work for rounds [lo, hi): staging batches through the backend.  Returns the per-round columns
    (blob_offset relative to this range's payload region) and the region's bytes."""
    n = hi - lo
    cols = dict(blob_offset=np.zeros(n, np.uint64), blob_size=np.zeros(n, np.uint64),
                checksum=np.zeros((n, 32), np.uint8), compressed=np.zeros(n, np.uint8))
    parts, cursor, i = [], 0, lo
    while i < hi:
        j, nbytes = i, 0
        while j < hi and (j == i or nbytes + rounds[j][2] <= BATCH_BYTES):
            nbytes += rounds[j][2]
            j += 1
        staging = np.empty(nbytes, dtype=np.uint8)
        off = np.zeros(j - i, dtype=np.uint64)
        ln = np.zeros(j - i, dtype=np.uint64)
        sk = np.zeros(j - i, dtype=np.uint8)
        pos = 0
        for k, (fi, start, l, skip, _, _) in enumerate(rounds[i:j]):
            if l:
                staging[pos:pos + l] = np.frombuffer(entries[fi].data, dtype=np.uint8, count=l, offset=start)
            off[k], ln[k], sk[k] = pos, l, 1 if skip else 0
            pos += l
        res, blob = backend.encode_hash(staging, off, ln, sk)
        cols["blob_offset"][i - lo:j - lo] = np.asarray(res["blob_offset"], np.uint64) + np.uint64(cursor)
        cols["blob_size"][i - lo:j - lo] = res["blob_size"]
        cols["checksum"][i - lo:j - lo] = res["checksum"]
        cols["compressed"][i - lo:j - lo] = res["compressed"]
        parts.append(np.asarray(blob, np.uint8).tobytes())
        cursor += len(parts[-1])
        i = j
    return cols, b"".join(parts)


def run_pipeline(entries, output, no_skip, backend=None, config=None) -> ix.CompressionReport:
    """Single process: all rounds.  Inside an initialised torch.distributed group every rank encodes a contiguous
    range of the rounds balanced by bytes (its GPU's share), rank 0 concatenates the payload regions in rank order —
    blob offsets are a running sum, so rebasing a region is one addition — and writes the archive; the
    report is the same on every rank.  Every rank is handed the same entries."""
    from .backend import default_backend
    from .sharding import split_rows
    backend = backend or default_backend()
    config = config or ix.CONFIG
    backend.set_level(config.compression_level)  # CompressCtx::new(CONFIG.compression_level), stream_packer.rs:L217 / slot_packer.rs:L551
    output_path = with_extension(output, "znippy")  # L132
    rounds, (uf, ub, cf, cb) = plan_rounds(entries, no_skip)
    rank, world = 0, 1
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            rank, world = dist.get_rank(), dist.get_world_size()
    except ImportError:
        dist = None
    if world > 1:
        lo, hi = split_rows([r[2] for r in rounds], world)[rank]
        mine = encode_round_range(rounds, entries, backend, lo, hi)
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        report = [None]
        if rank == 0:
            base, col_parts, regions = 0, [], []
            for cols, region in gathered:
                cols = dict(cols)
                cols["blob_offset"] = cols["blob_offset"] + np.uint64(base)
                base += len(region)
                col_parts.append(cols)
                regions.append(region)
            cols = {k: np.concatenate([c[k] for c in col_parts]) for k in col_parts[0]}
            report[0] = _write_archive(output_path, entries, rounds, cols, b"".join(regions), (uf, ub, cf, cb), config)
        dist.broadcast_object_list(report, src=0)
        return report[0]
    cols, region = encode_round_range(rounds, entries, backend, 0, len(rounds))
    return _write_archive(output_path, entries, rounds, cols, region, (uf, ub, cf, cb), config)


def _write_archive(output_path, entries, rounds, cols, region, counts, config) -> ix.CompressionReport:
    uf, ub, cf, cb = counts
    blobs: List[ix.BlobMeta] = []
    with open(output_path, "wb+") as f:
        os.pwrite(f.fileno(), region, 0)  # the writer: payloads from offset 0 in round order (L255-284)
        out_cursor = len(region)
        for k, (fi, start, l, skip, fdata_offset, chunk_seq) in enumerate(rounds):
            bsz = int(cols["blob_size"][k])
            blobs.append(ix.BlobMeta(
                ix.ChunkMeta(fdata_offset=fdata_offset, file_index=fi, chunk_seq=chunk_seq,
                             checksum=cols["checksum"][k].tobytes(), compressed=bool(cols["compressed"][k]),
                             uncompressed_size=l, compressed_size=bsz),
                blob_offset=int(cols["blob_offset"][k]), blob_size=bsz))

        # finalizer (L293-346)
        blobs.sort(key=lambda b: (b.chunk_meta.file_index, b.chunk_meta.chunk_seq))
        blob_bytes = out_cursor
        file_keys = [((e.pkg_type if e.pkg_type is not None else 0), (e.repo if e.repo is not None else ""))
                     for e in entries]
        groups = {}
        for b in blobs:
            groups.setdefault(file_keys[b.chunk_meta.file_index], []).append(b)
        meta_map = ix.build_arrow_metadata_for_config(config)
        schema_with_meta = ix.compose_index_schema().with_metadata(meta_map)
        sink = ix.ArrowIpcSink(f, blob_bytes)
        for key in sorted(groups.keys()):  # BTreeMap<(i8, String)> order
            batch = ix.build_metadata_batch(groups[key], lambda fi: entries[fi].relative_path)
            batch = batch.replace_schema_metadata(meta_map) if hasattr(batch, "replace_schema_metadata") else batch
            sink.push_subindex(schema_with_meta, [batch], key[0], key[1], "")
        total_bytes_out = sink.finish()

    return ix.CompressionReport(
        total_files=uf + cf, compressed_files=cf, uncompressed_files=uf, chunks=len(blobs), total_dirs=0,
        total_bytes_in=cb + ub, total_bytes_out=total_bytes_out, compressed_bytes=cb, uncompressed_bytes=ub,
        compression_ratio=(cb / (total_bytes_out - ub) * 100.0) if (cb > 0 and total_bytes_out > ub) else 0.0)
