"""Soak of the read path on foreign frames (GPU box): many tables of libzstd frames of generated data — alphabets of
every size, Zipf word streams, struct-like binary records, runs, sparse buffers, mixtures — at random levels (fast
negative levels included) and advanced parameters (window log, minimum match, strategy, content checksum), sizes 0 B ..
2 MiB.  Every row is decoded by the default path, by the path with no batch kernels (ZNIPPY_NO_BX) and compared with
its source; any difference prints the seed of the entry.  Input generation and compression run in worker processes
started before the GPU is touched.

    python tools/soak_foreign.py [tables] [entries per table] [first seed]
"""
import ctypes as C
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def _entry(seed):
    rng = np.random.default_rng(seed)
    kind = int(rng.integers(0, 9))
    n = int(rng.choice([0, 1, 2, 3, 17, 100, 1000, 5000, 10240, 40000, 131072, 131073, 200000, 500000, 2 << 20],
                       p=[.02, .02, .02, .02, .04, .08, .12, .15, .15, .12, .08, .05, .06, .04, .03]))
    n = int(rng.integers(max(n // 2, 0), n + 1)) if n > 3 else n
    if kind == 0:      # alphabet of A symbols, uniform
        A = int(rng.choice([1, 2, 3, 5, 16, 64, 200, 256]))
        e = rng.integers(0, A, size=n, dtype=np.uint8)
    elif kind == 1:    # skewed symbols (geometric)
        e = np.minimum(rng.geometric(float(rng.choice([0.02, 0.1, 0.3, 0.7])), size=n), 255).astype(np.uint8)
    elif kind == 2:    # Zipf words
        V = int(rng.choice([20, 300, 5000]))
        words = [bytes(rng.integers(97, 123, size=int(rng.integers(1, 12)), dtype=np.uint8)) for _ in range(V)]
        idx = np.minimum(rng.zipf(1.3, size=n // 4 + 1), V) - 1
        e = np.frombuffer(b" ".join(words[i] for i in idx)[:n].ljust(n, b"."), dtype=np.uint8)
    elif kind == 3:    # records: counter + small fields + noise byte
        r = int(rng.choice([8, 12, 24, 40]))
        m = n // r + 1
        rec = np.zeros((m, r), np.uint8)
        rec[:, 0:4] = np.arange(m, dtype=np.uint32).view(np.uint8).reshape(m, 4)
        rec[:, 4] = rng.integers(0, 4, size=m)
        rec[:, r - 1] = rng.integers(0, 256, size=m)
        e = rec.reshape(-1)[:n]
    elif kind == 4:    # runs of random length
        out = np.empty(n, np.uint8); p = 0
        while p < n:
            L = int(rng.choice([1, 3, 30, 300, 70000])); L = min(int(rng.integers(1, L + 1)), n - p)
            out[p:p + L] = rng.integers(0, 256); p += L
        e = out
    elif kind == 5:    # sparse: zeros with islands
        out = np.zeros(n, np.uint8)
        for _ in range(int(rng.integers(0, 40))):
            if n < 2: break
            p = int(rng.integers(0, n - 1)); L = min(int(rng.integers(1, 200)), n - p)
            out[p:p + L] = rng.integers(0, 256, size=L)
        e = out
    elif kind == 6:    # copy-heavy: earlier pieces re-pasted at random distances (long and short offsets)
        out = rng.integers(0, 256, size=n, dtype=np.uint8); p = min(64, n)
        while p < n:
            L = min(int(rng.integers(3, 400)), n - p); d = int(rng.integers(1, p + 1))
            if rng.random() < 0.7:
                for k in range(L): out[p + k] = out[p + k - d]
            p += L
        e = out
    elif kind == 7:    # mixture: segments of the other kinds
        parts = []
        left = n
        while left > 0:
            L = min(left, int(rng.integers(1, 70000)))
            sub = _entry(int(rng.integers(0, 1 << 30)) * 9 + int(rng.integers(0, 7)))[0][:L]
            if len(sub) == 0: sub = bytes(L)
            parts.append(sub); left -= len(sub)
        e = np.frombuffer(b"".join(parts)[:n], dtype=np.uint8)
    else:              # text-like bytes from a fixed phrase with mutations
        base = np.frombuffer((b"the quick brown fox jumps over the lazy dog; " * (n // 45 + 1))[:n], dtype=np.uint8).copy()
        if n:
            hits = rng.integers(0, n, size=n // int(rng.choice([7, 50, 1000])) + 1)
            base[hits] = rng.integers(32, 127, size=len(hits))
        e = base
    data = bytes(e[:n].tobytes())
    level = int(rng.choice([-5, -1, 1, 2, 3, 5, 7, 9, 12, 15, 17, 19, 20, 22]))
    prm = []
    if rng.random() < 0.25: prm.append((101, int(rng.integers(10, 24))))     # windowLog
    if rng.random() < 0.2: prm.append((105, int(rng.integers(3, 8))))        # minMatch
    if rng.random() < 0.2: prm.append((107, int(rng.integers(1, 10))))       # strategy
    if rng.random() < 0.15: prm.append((201, 1))                             # content checksum
    if rng.random() < 0.1: prm.append((200, 0))                              # no content size in the header
    return data, level, prm


_z = None


def _lib():
    global _z
    if _z is None:
        z = C.CDLL("libzstd.so.1")
        z.ZSTD_createCCtx.restype = C.c_void_p
        z.ZSTD_freeCCtx.argtypes = [C.c_void_p]
        z.ZSTD_CCtx_setParameter.restype = C.c_size_t
        z.ZSTD_CCtx_setParameter.argtypes = [C.c_void_p, C.c_int, C.c_int]
        z.ZSTD_compressBound.restype = C.c_size_t
        z.ZSTD_compressBound.argtypes = [C.c_size_t]
        z.ZSTD_compress2.restype = C.c_size_t
        z.ZSTD_compress2.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        z.ZSTD_isError.restype = C.c_uint
        z.ZSTD_isError.argtypes = [C.c_size_t]
        _z = z
    return _z


def make(seed):
    data, level, prm = _entry(seed)
    z = _lib()
    cctx = z.ZSTD_createCCtx()
    ok = not z.ZSTD_isError(z.ZSTD_CCtx_setParameter(cctx, 100, level))
    used = []
    for p, v in prm:
        if not z.ZSTD_isError(z.ZSTD_CCtx_setParameter(cctx, p, v)):
            used.append((p, v))
    cap = z.ZSTD_compressBound(len(data))
    out = C.create_string_buffer(cap)
    r = z.ZSTD_compress2(cctx, out, cap, data, len(data))
    z.ZSTD_freeCCtx(cctx)
    if not ok or z.ZSTD_isError(r):
        return seed, data, None, level, used
    return seed, data, out.raw[:r], level, used


def main():
    tables = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
    seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    pool = Pool(int(os.environ.get("SOAK_WORKERS", "12")))
    jobs = [pool.map_async(make, range(seed0 + t * per, seed0 + (t + 1) * per), chunksize=16) for t in range(tables)]
    import torch
    from znippy_amd import hip
    from oracle import oracle as O
    O.build()
    bad_total = 0
    for t, job in enumerate(jobs):
        t0 = time.time()
        items = [it for it in job.get() if it[2] is not None]
        if os.environ.get("SOAK_ROWS"):  # diagnosis: only rows [a, b) of the table
            a_, b_ = (int(v) for v in os.environ["SOAK_ROWS"].split(":"))
            items = items[a_:b_]
        entries = [it[1] for it in items]
        frames = [it[2] for it in items]
        bs = np.array([len(f) for f in frames], np.uint64)
        bo = np.concatenate([[0], np.cumsum(bs)[:-1]]).astype(np.uint64)
        us = np.array([len(e) for e in entries], np.uint64)
        oo = np.concatenate([[0], np.cumsum(us)[:-1]]).astype(np.uint64)
        ck = np.stack([np.frombuffer(O.blake3(e), dtype=np.uint8) for e in entries])
        d_blobs = torch.from_numpy(np.frombuffer(b"".join(frames) + bytes(64), dtype=np.uint8).copy()).cuda()
        total = int(us.sum())
        src = np.frombuffer(b"".join(entries), dtype=np.uint8)
        res = {}
        for mode in ("default", "no_bx"):
            if mode == "no_bx": os.environ["ZNIPPY_NO_BX"] = "1"
            else: os.environ.pop("ZNIPPY_NO_BX", None)
            ctx = hip.Context(0)
            rt = hip.RowTable(ctx, bo, bs, us, oo, None, ck)
            d_out = torch.zeros(total + 64, dtype=torch.uint8, device="cuda")
            for rep in range(2):
                d_out.zero_()
                c, corrupt, status = rt.decode_verify(d_blobs, d_out)
                out = d_out[:total].cpu().numpy()
                nofcs = np.array([(200, 0) in it[4] for it in items], bool)   # no content size in the header: refused (-6) by design
                badrows = [i for i in np.nonzero((status != 0) & ~nofcs | nofcs & (status != -6) & (status != 0))[0]]
                if not (out == src).all():  # rows the run called verified with different bytes: must never happen
                    diff = np.nonzero(out != src)[0]
                    rows_ = np.unique(np.searchsorted(oo, diff, side="right") - 1)
                    rows_ = [int(i) for i in rows_ if status[i] == 0]
                    if rows_:
                        print("BYTES DIFFER on rows the run called verified:", mode, rows_[:10])
                        badrows += rows_
                for i in badrows[:20]:
                    print("FAIL table %d mode %s rep %d row %d seed %d status %d size %d frame %d level %d prm %s"
                          % (t, mode, rep, i, items[i][0], int(status[i]), len(entries[i]), len(frames[i]), items[i][3], items[i][4]))
                bad_total += len(badrows)
                res[(mode, rep)] = (dict(c), status.copy())
            st = rt.foreign_stats() if hasattr(rt, "foreign_stats") else {}
            rt.close(); ctx.close()
        os.environ.pop("ZNIPPY_NO_BX", None)
        print("table %d: %d frames, %.1f MB -> %.1f MB, %.1f s, bad so far %d, stats %s"
              % (t, len(frames), int(bs.sum()) / 1e6, total / 1e6, time.time() - t0, bad_total, st), flush=True)
    pool.close()
    print("SOAK DONE bad =", bad_total)
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())
