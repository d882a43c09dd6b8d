"""Host-level (PCIe + file I/O inclusive) timings of the compiled host layer on the C2 shape.
Not the bench metric: inputs start in host memory / on disk (tmpfs when available).  Each leg runs twice;
the second run has the HIP runtime, code objects and page cache warm."""
import os, sys, time, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen
from znippy_amd import host
from znippy_amd.stream_packer import ArchiveEntry
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
base = "/dev/shm" if os.path.isdir("/dev/shm") else None
d = tempfile.mkdtemp(dir=base)
mb = n * 10240 / 2**20
try:
    chunk = gen.text(10240)
    ents = [ArchiveEntry(f"files/file_{i:06}.txt", chunk) for i in range(n)]
    for rep_i in range(2):
        t0 = time.perf_counter()
        c = host.compress_stream(os.path.join(d, "c2.znippy"), False)
        for e in ents:
            c.send(e)
        t1 = time.perf_counter()
        rep = c.finish()
        t2 = time.perf_counter()
        print(f"[{rep_i}] compress_stream: open+send {t1-t0:.3f}s  finish {t2-t1:.3f}s -> {mb/(t2-t0):.0f} MB/s end to end, "
              f"archive {rep.total_bytes_out/1e6:.1f} MB, chunks {rep.chunks}", flush=True)
    import numpy as np
    packed = np.frombuffer(chunk * n, dtype=np.uint8)
    offs = np.arange(n + 1, dtype=np.uint64) * len(chunk)
    names = [e.relative_path for e in ents]
    for rep_i in range(2):
        t0 = time.perf_counter()
        c = host.compress_stream(os.path.join(d, "c2p.znippy"), False)
        c.send_packed(names, packed, offs)
        t1 = time.perf_counter()
        rep = c.finish()
        t2 = time.perf_counter()
        print(f"[{rep_i}] compress_stream, ONE packed send of {n} entries: open+send {t1-t0:.3f}s  finish {t2-t1:.3f}s -> {mb/(t2-t0):.0f} MB/s end to end, "
              f"archive {rep.total_bytes_out/1e6:.1f} MB, chunks {rep.chunks}", flush=True)
    same = open(os.path.join(d, "c2.znippy"), "rb").read() == open(os.path.join(d, "c2p.znippy"), "rb").read()
    print("packed archive identical to the per-entry one:", same, flush=True)
    for rep_i in range(2):
        t0 = time.perf_counter(); v = host.decompress_archive(os.path.join(d, "c2.znippy"), False, "/dev/null"); t1 = time.perf_counter()
        print(f"[{rep_i}] verify (save_data=false): {t1-t0:.3f}s -> {mb/(t1-t0):.0f} MB/s, corrupt {v.corrupt_files}, chunks {v.chunks}", flush=True)
    for rep_i in range(2):
        shutil.rmtree(os.path.join(d, "out"), ignore_errors=True)
        t0 = time.perf_counter(); v = host.decompress_archive(os.path.join(d, "c2.znippy"), True, os.path.join(d, "out")); t1 = time.perf_counter()
        print(f"[{rep_i}] decompress_archive (save_data=true, {n} files): {t1-t0:.3f}s -> {mb/(t1-t0):.0f} MB/s", flush=True)
    for rep_i in range(2):
        t0 = time.perf_counter(); r = host.compress_dir(os.path.join(d, "out"), os.path.join(d, "dir.znippy")); t1 = time.perf_counter()
        print(f"[{rep_i}] compress_dir ({r.total_files} files from tmpfs): {t1-t0:.3f}s -> {mb/(t1-t0):.0f} MB/s, chunks {r.chunks}", flush=True)
    v = host.decompress_archive(os.path.join(d, "dir.znippy"), False, "/dev/null")
    print("dir archive verify:", v.total_files, v.corrupt_files, v.total_bytes)
finally:
    shutil.rmtree(d, ignore_errors=True)
