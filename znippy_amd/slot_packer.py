"""compress_dir — host-side mirror of znippy-compress/src/slot_packer.rs (directory ingest).

Same surface and rules: walk the directory, partition into big (size > slice_size or empty) and
small files (L92-101), big files are cut into slice_size rounds with fdata_offset/chunk_seq
(L265-280), every small file is ONE round with fdata_offset = 0, chunk_seq = 0 (L499); rows are
NOT sorted (big pass batch then small pass batch, L141-189); one sub-index for both batches with
GroupKey{pkg_type 0, repo}.  slice_size = SLOT_SIZE / num_workers (L30-31,L89).
The io_uring reader and the plugin metadata hooks are out of scope; the worker
body (BLAKE3 + encode, L551-580) runs on the GPU through the backend, one staging batch per "slot".
"""
import os
from typing import Optional

import numpy as np

from . import index as ix
from .stream_packer import with_extension

SLOT_SIZE = 200 * 1024 * 1024  # slot_packer.rs:L30
NUM_SLOTS = 8                  # slot_packer.rs:L31


def compress_dir(input_dir, output, no_skip: bool = False, plugin=None, repo: Optional[str] = None, backend=None,
                 config=None) -> ix.CompressionReport:
    if plugin is not None:
        raise NotImplementedError("metadata plugins are outside the hot path (SURVEY §2 #12)")
    from .backend import default_backend
    backend = backend or default_backend()
    config = config or ix.CONFIG
    backend.set_level(config.compression_level)  # CompressCtx::new(CONFIG.compression_level), stream_packer.rs:L217 / slot_packer.rs:L551
    input_dir = str(input_dir)
    total_dirs = 0
    all_files = []
    for root, dirs, files in os.walk(input_dir):  # WalkDir counts the root too (L63-78)
        total_dirs += 1
        dirs.sort()
        for f in sorted(files):
            full = os.path.join(root, f)
            if os.path.isfile(full) and not os.path.islink(full):
                all_files.append(full)
    num_workers = max(config.max_core_in_flight, 1)
    slice_size = SLOT_SIZE // num_workers
    sizes = [os.path.getsize(p) for p in all_files]
    big = [i for i, s in enumerate(sizes) if s > slice_size or s == 0]
    small = [i for i, s in enumerate(sizes) if not (s > slice_size or s == 0)]

    def rel(i):
        return os.path.relpath(all_files[i], input_dir)

    output_path = with_extension(str(output), "znippy")
    uf = ub = cf = cb = 0
    out_cursor = 0
    batches = []
    with open(output_path, "wb+") as f:
        def run_pass(indices, is_big):
            nonlocal uf, ub, cf, cb, out_cursor
            rounds = []  # (file_index, file_offset, len, skip, fdata_offset, chunk_seq)
            for i in indices:
                skip = (not no_skip) and ix.should_skip_compression(all_files[i])
                if skip:
                    uf += 1; ub += sizes[i]
                else:
                    cf += 1; cb += sizes[i]
                if sizes[i] == 0:
                    rounds.append((i, 0, 0, skip, 0, 0))
                elif is_big:
                    off = seq = 0
                    while off < sizes[i]:
                        l = min(slice_size, sizes[i] - off)
                        rounds.append((i, off, l, skip, off, seq))
                        off += l
                        seq += 1
                else:
                    rounds.append((i, 0, sizes[i], skip, 0, 0))
            blobs = []
            k = 0
            while k < len(rounds):  # one staging batch per Magazine (NUM_SLOTS x SLOT_SIZE)
                j, nbytes = k, 0
                while j < len(rounds) and (j == k or nbytes + rounds[j][2] <= NUM_SLOTS * SLOT_SIZE):
                    nbytes += rounds[j][2]
                    j += 1
                staging = np.empty(nbytes, dtype=np.uint8)
                off = np.zeros(j - k, np.uint64); ln = np.zeros(j - k, np.uint64); sk = np.zeros(j - k, np.uint8)
                pos = 0
                for q, (fi, fo, l, skip, _, _) in enumerate(rounds[k:j]):
                    if l:
                        with open(all_files[fi], "rb") as src:
                            src.seek(fo)
                            staging[pos:pos + l] = np.frombuffer(src.read(l), dtype=np.uint8)
                    off[q], ln[q], sk[q] = pos, l, 1 if skip else 0
                    pos += l
                res, blob = backend.encode_hash(staging, off, ln, sk)
                os.pwrite(f.fileno(), blob.tobytes(), out_cursor)
                for q, (fi, fo, l, skip, fdo, seq) in enumerate(rounds[k:j]):
                    bsz = int(res["blob_size"][q])
                    blobs.append(ix.BlobMeta(
                        ix.ChunkMeta(fdata_offset=fdo, file_index=fi, chunk_seq=seq,
                                     checksum=res["checksum"][q].tobytes(), compressed=bool(res["compressed"][q]),
                                     uncompressed_size=l, compressed_size=bsz),
                        blob_offset=out_cursor + int(res["blob_offset"][q]), blob_size=bsz))
                out_cursor += len(blob)
                k = j
            return blobs

        total_chunks = 0
        for indices, is_big in ((big, True), (small, False)):
            if indices:
                blobs = run_pass(indices, is_big)
                total_chunks += len(blobs)
                batches.append(ix.build_metadata_batch(blobs, rel))
        blob_bytes = out_cursor
        meta_map = ix.build_arrow_metadata_for_config(config)
        schema_with_meta = ix.compose_index_schema().with_metadata(meta_map)
        sink = ix.ArrowIpcSink(f, blob_bytes)
        sink.push_subindex(schema_with_meta, batches, 0, repo or "", "")
        total_bytes_out = sink.finish()
    return ix.CompressionReport(
        total_files=len(all_files), compressed_files=cf, uncompressed_files=uf, chunks=total_chunks,
        total_dirs=total_dirs, total_bytes_in=cb + ub, total_bytes_out=total_bytes_out, compressed_bytes=cb,
        uncompressed_bytes=ub, compression_ratio=(cb / max(blob_bytes, 1) * 100.0) if ub > 0 else 0.0)
