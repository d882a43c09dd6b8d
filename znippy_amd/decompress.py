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


def row_lengths(c):
    """Bytes each row produces: uncompressed_size, or blob_size for a stored row — the reference hashes and writes
    the blob itself there and never looks at uncompressed_size (decompress.rs:L143-166, `&read_buf`)."""
    return np.where(c["compressed"], c["usize"], c["blob_size"]).astype(np.uint64)


def read_spans(f, file_size, bo, bs, gap=64 << 10):
    """Blobs of the given rows as ONE packed buffer: rows are grouped into spans (neighbours closer than `gap` are
    read together), each span is one positioned read, and every row's offset is rebased into the buffer.  Rows
    that do not lie inside the file keep an offset PAST the buffer, so the device layer reports them
    (ZNIPPY_E_CORRUPT) instead of reading anything.  -> (buffer, rebased blob_offset)"""
    n = len(bo)
    new_bo = np.zeros(n, np.uint64)
    if n == 0:
        return np.zeros(0, np.uint8), new_bo
    ok = (bo <= file_size) & (bs <= file_size - np.minimum(bo, file_size))
    order = np.argsort(bo, kind="stable")
    parts, pos, bad = [], 0, []
    i = 0
    while i < n:
        k = int(order[i])
        if not ok[k]:
            bad.append(k)
            i += 1
            continue
        lo, hi = int(bo[k]), int(bo[k] + bs[k])
        members = [k]
        j = i + 1
        while j < n and ok[int(order[j])] and int(bo[int(order[j])]) <= hi + gap:
            m = int(order[j])
            hi = max(hi, int(bo[m] + bs[m]))
            members.append(m)
            j += 1
        f.seek(lo)
        data = f.read(hi - lo)
        if len(data) != hi - lo:  # truncated underneath us: the rows of this span become out-of-range rows
            bad.extend(members)
        else:
            parts.append(np.frombuffer(data, dtype=np.uint8))
            for m in members:
                new_bo[m] = pos + int(bo[m]) - lo
            pos += hi - lo
        i = j
    buf = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
    for k in bad:
        new_bo[k] = len(buf) + 1
    return buf, new_bo


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
    rlen = row_lengths(c)
    r0, r1 = split_rows(c["usize"], world)[rank]
    file_size = os.path.getsize(index_path)
    final_size = {}
    if world > 1 and save_data:  # several ranks write one file: whoever touches it first sets its final length
        ends = c["fdata_offset"] + rlen
        for p, e in zip(c["paths"], ends):
            e = int(e)
            if final_size.get(p, -1) < e:
                final_size[p] = e

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
        if first and world > 1:
            # no rank may O_TRUNC (the others write their parts in any order); instead each sets the file to its final
            # length at first touch — idempotent, and a longer file left by an earlier run loses its stale tail
            os.ftruncate(cur[1], final_size[p])
        return cur[1]

    backend = backend or (default_backend() if r1 > r0 else None)
    counters = dict(total_chunks=0, total_written_bytes=0, verified_bytes=0, corrupt_bytes=0, corrupt_rows=0,
                    decode_errors=0)
    corrupt_all = []
    with open(index_path, "rb") as arc:
        i = r0
        while i < r1:
            j, nbytes = i, 0
            while j < r1 and (j == i or nbytes + int(rlen[j]) <= RANGE_BYTES):
                nbytes += int(rlen[j])
                j += 1
            bs = c["blob_size"][i:j]
            blobs, bo = read_spans(arc, file_size, c["blob_offset"][i:j], bs)  # the preads of L148-153, coalesced
            usz = rlen[i:j]
            out_off = np.concatenate([[0], np.cumsum(usz)[:-1]]).astype(np.uint64)
            cnt, corrupt, status, out = backend.decode_verify(blobs, 0, bo, bs, usz, out_off, c["compressed"][i:j],
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
