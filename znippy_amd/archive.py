"""ZnippyArchive — host-side mirror of znippy-common/src/archive.rs (random-access reads).

open() loads only the index; extract_file(path) preads that file's blobs, decodes compressed
chunks on the GPU and concatenates them in fdata_offset order.  Like the reference, no checksum
verification happens on this path (archive.rs:L144-168)."""
import numpy as np

from . import index as ix
from .decompress import _columns


class ZnippyArchive:
    def __init__(self, path, backend=None):
        self.path = str(path)
        self._backend = backend
        _, batches = ix.read_znippy_index(self.path)
        c = _columns(batches)
        self._c = c
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

    def extract_file(self, relative_path) -> bytes:
        out = self.extract_files([relative_path])[0]
        if isinstance(out, Exception):
            raise out
        return out

    def extract_files(self, paths):
        """Batch extract: all requested files' chunks go to the GPU as one row set."""
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
            bo, bs, usz = c["blob_offset"][rows_np], c["blob_size"][rows_np], c["usize"][rows_np]
            lo, hi = int(bo.min()), int((bo + bs).max())
            with open(self.path, "rb") as f:
                f.seek(lo)
                blobs = np.frombuffer(f.read(hi - lo), dtype=np.uint8)
            out_off = np.concatenate([[0], np.cumsum(usz)[:-1]]).astype(np.uint64)
            backend = self._backend or default_backend()
            _, _, status, out = backend.decode_verify(blobs, lo, bo, bs, usz, out_off, c["compressed"][rows_np], None,
                                                      int(usz.sum()))
        for p, sp in zip(paths, spans):
            if sp is None:
                results.append(KeyError(f"file not found in archive: {p}"))
                continue
            a, b = sp
            if (status[a:b] < 0).any():
                results.append(ValueError(f"decompress failed for {p}: status {int(status[a:b].min())}"))
                continue
            start = int(out_off[a])
            end = int(out_off[b - 1] + usz[b - 1])
            results.append(out[start:end].tobytes())
        return results
