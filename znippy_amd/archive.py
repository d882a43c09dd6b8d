"""ZnippyArchive — host-side mirror of znippy-common/src/archive.rs (random-access reads).

open() loads only the index; extract_file(path) preads that file's blobs, decodes compressed
chunks on the GPU and concatenates them in fdata_offset order.  Like the reference, no checksum
verification happens on this path by default (archive.rs:L144-168); `verify=True` adds the check the
reference lacks: every chunk's BLAKE3 against the index's checksum column."""
import os

import numpy as np

from . import index as ix
from .decompress import _columns, read_spans, row_lengths


class ZnippyArchive:
    def __init__(self, path, backend=None):
        self.path = str(path)
        self._backend = backend
        _, batches = ix.read_znippy_index(self.path)
        c = _columns(batches)
        self._c = c
        self._rlen = row_lengths(c)
        self.file_index = {}
        for row, p in enumerate(c["paths"]):
            self.file_index.setdefault(p, []).append(row)
        for p, rows in self.file_index.items():  # chunks sorted by fdata_offset (L131-133)
            rows.sort(key=lambda r: int(c["fdata_offset"][r]))

    @classmethod
    def open(cls, path, backend=None):
        return cls(path, backend)

    def file_count(self):
        return len(self.file_index)

    def list_files(self):
        return list(self.file_index.keys())

    def contains(self, relative_path):
        return relative_path in self.file_index

    def file_size(self, relative_path):
        rows = self.file_index.get(relative_path)
        return None if rows is None else int(sum(int(self._c["usize"][r]) for r in rows))

    def extract_file(self, relative_path, verify=False) -> bytes:
        out = self.extract_files([relative_path], verify=verify)[0]
        if isinstance(out, Exception):
            raise out
        return out

    def extract_files(self, paths, verify=False):
        """Batch extract: all requested files' chunks go to the GPU as one row set.  Only the requested rows' blobs
        are read (rows that sit next to each other in the archive in one read), not the span between them."""
        from .backend import default_backend
        c = self._c
        rows, spans, results = [], [], []
        for p in paths:
            r = self.file_index.get(p)
            if r is None:
                spans.append(None)
                continue
            spans.append((len(rows), len(rows) + len(r)))
            rows.extend(r)
        if rows:
            rows_np = np.asarray(rows)
            bs, usz = c["blob_size"][rows_np], self._rlen[rows_np]
            with open(self.path, "rb") as f:
                blobs, bo = read_spans(f, os.path.getsize(self.path), c["blob_offset"][rows_np], bs)
            out_off = np.concatenate([[0], np.cumsum(usz)[:-1]]).astype(np.uint64)
            backend = self._backend or default_backend()
            _, corrupt, status, out = backend.decode_verify(blobs, 0, bo, bs, usz, out_off, c["compressed"][rows_np],
                                                            c["checksum"][rows_np] if verify else None, int(usz.sum()))
            corrupt = set(int(x) for x in corrupt)
        for p, sp in zip(paths, spans):
            if sp is None:
                results.append(KeyError(f"file not found in archive: {p}"))
                continue
            a, b = sp
            if (status[a:b] < 0).any():
                results.append(ValueError(f"decompress failed for {p}: status {int(status[a:b].min())}"))
                continue
            if verify and any(k in corrupt for k in range(a, b)):
                results.append(ValueError(f"checksum mismatch in {p}"))
                continue
            start = int(out_off[a])
            end = int(out_off[b - 1] + usz[b - 1])
            results.append(out[start:end].tobytes())
        return results
