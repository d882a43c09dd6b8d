"""The device backend the host-side pipelines drive: host buffers in, host buffers out, all
compute through libznippy_hip.so (C ABI).  There is no CPU implementation in the product; the
CPU tests inject a checker double from tests/ to exercise the host logic without a GPU."""
import numpy as np


class HipBackend:
    def __init__(self, device=None, ctx=None):
        import torch
        from . import hip
        if not torch.cuda.is_available():
            raise RuntimeError("znippy_amd: no GPU visible — the codec/hash path has no CPU fallback")
        self.torch = torch
        self.hip = hip
        self.device = torch.cuda.current_device() if device is None else device
        self.ctx = ctx or hip.Context(self.device)

    def set_level(self, level):
        """CompressCtx::new(compression_level), codec.rs:L16-28."""
        self.ctx.set_level(level)

    def _to_dev(self, a):
        t = self.torch.from_numpy(np.ascontiguousarray(a))
        return t.to(f"cuda:{self.device}", non_blocking=False)

    # write side: Rounds over one staging buffer -> packed blobs + per-round metadata
    def encode_hash(self, staging, off, length, skip):
        n = len(off)
        if n == 0:
            return dict(blob_offset=np.zeros(0, np.uint64), blob_size=np.zeros(0, np.uint64),
                        checksum=np.zeros((0, 32), np.uint8), compressed=np.zeros(0, np.uint8)), np.zeros(0, np.uint8)
        d_src = self._to_dev(np.concatenate([staging, np.zeros(64, np.uint8)]))
        rt = self.hip.RoundTable(self.ctx, off, length, skip)
        d_blob = self.torch.empty(rt.blob_bound() + 64, dtype=self.torch.uint8, device=f"cuda:{self.device}")
        res = rt.encode_hash(d_src, d_blob)
        res = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in res.items()}  # views die with the table
        blob = d_blob[:res["blob_bytes"]].cpu().numpy()
        rt.close()
        return res, blob

    # read side: rows of the index over a blob region -> decoded bytes + counters
    def decode_verify(self, blobs, blob_base, blob_offset, blob_size, usize, out_offset, compressed, checksum, out_total):
        n = len(blob_offset)
        if n == 0:
            return dict(total_chunks=0, total_written_bytes=0, verified_bytes=0, corrupt_bytes=0, corrupt_rows=0,
                        decode_errors=0), np.zeros(0, np.uint64), np.zeros(0, np.int32), np.zeros(0, np.uint8)
        d_blobs = self._to_dev(np.concatenate([blobs, np.zeros(64, np.uint8)]))
        d_out = self.torch.empty(out_total + 64, dtype=self.torch.uint8, device=f"cuda:{self.device}")
        bitmap = np.packbits(np.asarray(compressed, dtype=bool), bitorder="little")
        rt = self.hip.RowTable(self.ctx, blob_offset, blob_size, usize, out_offset, bitmap, checksum)
        counters, corrupt, status = rt.decode_verify(d_blobs, d_out, blob_base=blob_base, out_cap=out_total,
                                                     blob_cap=len(blobs))
        out = d_out[:out_total].cpu().numpy()
        rt.close()
        return counters, corrupt, status, out


_default = None


def default_backend():
    global _default
    if _default is None:
        _default = HipBackend()
    return _default
