"""decompress_archive — host-side mirror of znippy-common/src/decompress.rs.

decompress_archive(index_path, save_data, out_dir) -> VerifyReport, same semantics:
  * every index row is independent work; rows are handed to the GPU in ranges (the reference's
    atomic row cursor, L104/L136, split into per-rank contiguous ranges when torch.distributed is up)
  * decode error: row logged + skipped, counted in chunks only (L159-162)
  * checksum mismatch: counted, bytes STILL written (L175-189)
  * corrupt_files = number of corrupt ROWS, verified_files = total_files - corrupt_files (L210-211)
"""
import logging
import os

import numpy as np

from . import index as ix
from .sharding import reduce_counters, split_rows

log = logging.getLogger("znippy_amd")
RANGE_BYTES = 4 << 30  # decoded bytes per GPU hand-off


def _columns(batches):
    import pyarrow as pa
    if not batches:
        z64 = np.zeros(0, np.uint64)
        return dict(paths=[], chunk_seq=np.zeros(0, np.uint32), fdata_offset=z64, compressed=np.zeros(0, bool),
                    usize=z64, blob_offset=z64, blob_size=z64, checksum=np.zeros((0, 32), np.uint8))
    b = batches[0] if len(batches) == 1 else pa.Table.from_batches(batches).combine_chunks().to_batches()[0]
    col = lambda n: b.column(b.schema.names.index(n))
    ck = col("checksum")
    ck_np = np.frombuffer(ck.buffers()[1], dtype=np.uint8, count=32 * len(ck), offset=32 * ck.offset).reshape(-1, 32)
    return dict(paths=col("relative_path").to_pylist(),
                chunk_seq=col("chunk_seq").to_numpy(zero_copy_only=False),
                fdata_offset=col("fdata_offset").to_numpy(zero_copy_only=False).astype(np.uint64),
                compressed=col("compressed").to_numpy(zero_copy_only=False).astype(bool),
                usize=col("uncompressed_size").to_numpy(zero_copy_only=False).astype(np.uint64),
                blob_offset=col("blob_offset").to_numpy(zero_copy_only=False).astype(np.uint64),
                blob_size=col("blob_size").to_numpy(zero_copy_only=False).astype(np.uint64),
                checksum=ck_np)


def decompress_archive(index_path, save_data: bool, out_dir, backend=None, group=None) -> ix.VerifyReport:
    from .backend import default_backend
    index_path, out_dir = str(index_path), str(out_dir)
    _, batches = ix.read_znippy_index(index_path)
    c = _columns(batches)
    total_rows = len(c["paths"])
    total_files = len(set(c["paths"]))  # L64-69

    rank, world = 0, 1
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            rank, world = dist.get_rank(group), dist.get_world_size(group)
    except ImportError:
        pass
    r0, r1 = split_rows(c["usize"], world)[rank]

    # output files (L74-101): created/truncated on first touch, one cached descriptor (rows of a file are
    # adjacent) instead of the reference's table of open files
    created, cur = set(), [None, -1]

    def out_fd(p):
        if cur[0] == p:
            return cur[1]
        if cur[1] >= 0:
            os.close(cur[1])
        full = os.path.join(out_dir, p)
        first = p not in created
        if first:
            created.add(p)
            os.makedirs(os.path.dirname(full) or ".", exist_ok=True)
        flags = os.O_CREAT | os.O_WRONLY | (os.O_TRUNC if (first and world == 1) else 0)
        cur[0], cur[1] = p, os.open(full, flags, 0o644)
        return cur[1]

    backend = backend or (default_backend() if r1 > r0 else None)
    counters = dict(total_chunks=0, total_written_bytes=0, verified_bytes=0, corrupt_bytes=0, corrupt_rows=0,
                    decode_errors=0)
    corrupt_all = []
    with open(index_path, "rb") as arc:
        i = r0
        while i < r1:
            j, nbytes = i, 0
            while j < r1 and (j == i or nbytes + int(c["usize"][j]) <= RANGE_BYTES):
                nbytes += int(c["usize"][j])
                j += 1
            bo, bs = c["blob_offset"][i:j], c["blob_size"][i:j]
            lo = int(bo.min()) if j > i else 0
            hi = int((bo + bs).max()) if j > i else 0
            arc.seek(lo)
            blobs = np.frombuffer(arc.read(hi - lo), dtype=np.uint8)  # the preads of L148-153, coalesced
            usz = c["usize"][i:j]
            out_off = np.concatenate([[0], np.cumsum(usz)[:-1]]).astype(np.uint64)
            cnt, corrupt, status, out = backend.decode_verify(blobs, lo, bo, bs, usz, out_off, c["compressed"][i:j],
                                                              c["checksum"][i:j], int(usz.sum()))
            for k in counters:
                counters[k] += int(cnt[k])
            corrupt_all.extend(int(i + r) for r in corrupt)
            for r in corrupt:
                log.error("[verify] MISMATCH row=%d", i + int(r))
            for k in np.nonzero(status < 0)[0]:
                log.error("[decomp] row %d error=%d", i + int(k), int(status[k]))
            if save_data:
                for k in range(j - i):
                    if status[k] < 0:
                        continue  # decode error: nothing is written for the row (L159-162)
                    o, l = int(out_off[k]), int(usz[k])
                    os.pwrite(out_fd(c["paths"][i + k]), out[o:o + l].tobytes(), int(c["fdata_offset"][i + k]))
            i = j
    if cur[1] >= 0:
        os.close(cur[1])

    counters, corrupt_all = reduce_counters(counters, corrupt_all, group)
    corrupt_files = len(set(corrupt_all))
    return ix.VerifyReport(total_files=total_files, verified_files=max(total_files - corrupt_files, 0),
                           corrupt_files=corrupt_files, total_bytes=counters["total_written_bytes"],
                           verified_bytes=counters["verified_bytes"], corrupt_bytes=counters["corrupt_bytes"],
                           chunks=counters["total_chunks"], corrupt_rows=corrupt_all)


def verify_archive_integrity(path, backend=None) -> ix.VerifyReport:  # index.rs:L550-553
    return decompress_archive(path, False, "/dev/null", backend=backend)


def decompress_microchunk(data: bytes) -> bytes:  # decompress.rs:L224-226
    from .codec import decompress_frame
    return decompress_frame(data)
