"""TEST DOUBLE (not product code): a backend with the HipBackend interface implemented by the
oracle's restated worker loops, so the host-side pipelines (chunking rules, index layout, report
arithmetic, sharding) can be exercised by `-m "not gpu"` tests and by the world_size-2 gloo tests.
It doubles as the checker the GPU runs are compared with."""
import numpy as np

from oracle import oracle as O


class OracleBackend:
    def __init__(self, level=3, n_threads=2):
        self.level = level
        self.n_threads = n_threads

    def set_level(self, level):  # the double keeps its own (cheap) libzstd level: host logic does not depend on it
        pass

    def encode_hash(self, staging, off, length, skip):
        if len(off) == 0:
            return dict(blob_offset=np.zeros(0, np.uint64), blob_size=np.zeros(0, np.uint64),
                        checksum=np.zeros((0, 32), np.uint8), compressed=np.zeros(0, np.uint8)), np.zeros(0, np.uint8)
        st = np.concatenate([np.asarray(staging, dtype=np.uint8), np.zeros(16, np.uint8)])
        r = O.compress_rounds(st, off, length, skip, level=self.level, n_threads=1)  # 1 thread: round order
        return r, r["blobs"].copy()

    def decode_verify(self, blobs, blob_base, blob_offset, blob_size, usize, out_offset, compressed, checksum, out_total):
        n = len(blob_offset)
        out = np.zeros(max(out_total, 1), dtype=np.uint8)
        bitmap = np.packbits(np.asarray(compressed, dtype=bool), bitorder="little")
        have_ck = checksum is not None
        ck = np.ascontiguousarray(checksum, dtype=np.uint8) if have_ck else np.zeros((n, 32), np.uint8)
        rel = np.asarray(blob_offset, dtype=np.uint64) - np.uint64(blob_base)
        bl = np.concatenate([np.asarray(blobs, dtype=np.uint8), np.zeros(16, np.uint8)])
        # rows pointing outside the blob region: the device layer reports them (ZNIPPY_E_CORRUPT) without reading
        # anything; here they are fed to the loop as empty frames, which fail to decode the same way
        blob_size = np.asarray(blob_size, dtype=np.uint64).copy()
        compressed = np.asarray(compressed, dtype=bool).copy()
        oob = (rel > len(blobs)) | (blob_size > np.uint64(len(blobs)) - np.minimum(rel, np.uint64(len(blobs))))
        rel = np.where(oob, np.uint64(0), rel)
        blob_size[oob] = 0
        compressed[oob] = True
        bitmap = np.packbits(compressed, bitorder="little")
        st, corrupt = O.decompress_rows(bl, rel, blob_size, usize, out_offset, bitmap, ck, 0, n, out=out,
                                        n_threads=self.n_threads, use_libzstd=False, corrupt_cap=max(n, 1))
        status = np.zeros(n, dtype=np.int32)
        if st["decode_errors"]:
            # find which rows failed (the loop only counts them)
            for i in range(n):
                if compressed[i]:
                    try:
                        O.zstd_decompress(bl[int(rel[i]):int(rel[i] + blob_size[i])].tobytes(), cap=int(usize[i]))
                    except ValueError:
                        status[i] = -5
        if not have_ck:
            st.update(verified_bytes=st["total_written_bytes"], corrupt_bytes=0, corrupt_rows=0)
            corrupt = np.zeros(0, np.uint64)
        return st, corrupt, status, out[:out_total]
