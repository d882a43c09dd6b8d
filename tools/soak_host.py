"""Soak of the compiled host layer (GPU box): archives of generated entries (the generator of soak_foreign.py: nine kinds of
content, 0 B .. 2 MiB, plus a few files of 9-40 MiB that span several slices) written by compress_stream at a random level,
then: the C++ reader and the Python mirror both verify and extract them, every extracted file is compared with its source,
the two reports agree, and a split over 3 ranks covers every chunk once.

    python tools/soak_host.py [archives] [entries per archive] [first seed]
"""
import os
import shutil
import sys
import tempfile
import time
from multiprocessing import Pool
from pathlib import Path

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import soak_foreign as SF


def entry(seed):
    data = SF._entry(seed)[0]
    rng = np.random.default_rng(seed ^ 0x5EED)
    if rng.random() < 0.01:   # a file of several slices
        n = int(rng.integers(9 << 20, 40 << 20))
        data = (data or b"x") * (n // max(len(data), 1) + 1)
        data = data[:n]
    return data


def main():
    archives = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    pool = Pool(int(os.environ.get("SOAK_WORKERS", "12")))
    jobs = [pool.map_async(entry, range(seed0 + t * per, seed0 + (t + 1) * per), chunksize=8) for t in range(archives)]
    import torch  # noqa: F401  (the workers exist before the GPU is touched)
    from znippy_amd import host
    from znippy_amd.decompress import decompress_archive as py_decompress
    from znippy_amd.stream_packer import ArchiveEntry
    bad = 0
    for t, job in enumerate(jobs):
        t0 = time.time()
        datas = job.get()
        rng = np.random.default_rng(seed0 + 31 * t)
        exts = [".txt", ".bin", ".xml", ".jar", ".png", ".so", "", ".zst", ".gz", ".json"]
        ents = [ArchiveEntry("d%d/f%05d%s" % (i % 7, i, exts[int(rng.integers(0, len(exts)))]), d) for i, d in enumerate(datas)]
        tmp = Path(tempfile.mkdtemp(prefix="soak_host_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None))
        try:
            os.environ["ZNIPPY_LEVEL"] = str(int(rng.choice([1, 3, 9, 19])))
            no_skip = bool(rng.integers(0, 2))
            c = host.compress_stream(tmp / "a.tmp", no_skip)
            s = c.sender()
            for e in ents:
                s.send(e)
            rep = c.finish()
            archive = tmp / "a.znippy"
            nb = 0
            if rep.total_files != len(ents) or rep.total_bytes_in != sum(len(e.data) for e in ents):
                nb += 1; print("FAIL archive %d: write report %s" % (t, rep))
            rc = host.decompress_archive(archive, True, tmp / "out_cpp")
            rp = py_decompress(archive, True, tmp / "out_py")
            if rc != rp or rc.corrupt_files or rc.total_files != len(ents):
                nb += 1; print("FAIL archive %d: reports differ or corrupt: cpp %s py %s" % (t, rc, rp))
            for e in ents:
                for d in ("out_cpp", "out_py"):
                    p = tmp / d / e.relative_path
                    if not p.exists() or p.read_bytes() != e.data:
                        nb += 1
                        if nb < 10: print("FAIL archive %d: %s/%s differs (%d B)" % (t, d, e.relative_path, len(e.data)))
            parts = [host.decompress_archive(archive, False, "/dev/null", rank=r, world=3) for r in range(3)]
            if sum(p.chunks for p in parts) != rc.chunks or sum(p.total_bytes for p in parts) != rep.total_bytes_in or any(p.corrupt_files for p in parts):
                nb += 1; print("FAIL archive %d: rank split %s" % (t, parts))
            bad += nb
            print("host archive %d: %d files, %.1f MB -> %.1f MB, level %s no_skip %d, %d chunks, %.1f s, bad so far %d"
                  % (t, len(ents), rep.total_bytes_in / 1e6, rep.total_bytes_out / 1e6, os.environ["ZNIPPY_LEVEL"], no_skip, rc.chunks, time.time() - t0, bad), flush=True)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    pool.close()
    print("SOAK host DONE bad =", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
