"""Diagnostic: the two-phase foreign-frame path on ONE kind of data at a time (a shared object of the image or python
sources), libzstd frames at the given level.  Usage: python tools/diag_fz.py [kind=binary|text] [MiB=8] [frames=1] [level=19]
With ZNIPPY_DDBG=1 the library prints where the execute kernel's cycles went."""
import glob, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, workloads
from znippy_amd import hip
kind = sys.argv[1] if len(sys.argv) > 1 else "binary"
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nfr = int(sys.argv[3]) if len(sys.argv) > 3 else 1
level = int(sys.argv[4]) if len(sys.argv) > 4 else 19
pats = {"binary": ["/opt/rocm/lib/librocblas.so*", "/opt/rocm/lib/*.so*"], "text": ["/usr/lib/python3.10/*.py", "/usr/lib/python3.10/*/*.py"]}[kind]
data = b""
for pat in pats:
    for f in sorted(glob.glob(pat)):
        if os.path.islink(f) or not os.path.isfile(f):
            continue
        data += open(f, "rb").read(64 << 20)
        if len(data) >= (mib << 20) * min(nfr, 4):
            break
    if len(data) >= (mib << 20) * min(nfr, 4):
        break
sz = mib << 20
distinct = max(1, min(nfr, len(data) // sz))
ents = [data[i * sz:(i + 1) * sz] for i in range(distinct)]
t0 = time.time()
frames = [workloads.libzstd_compress(e, level) for e in ents]
print(f"{kind}: {distinct} distinct x {sz} B, libzstd -{level} ratio {sum(map(len, frames)) / (distinct * sz):.3f} ({time.time() - t0:.1f} s on the CPU)")
ents = [ents[i % distinct] for i in range(nfr)]
frames = [frames[i % distinct] for i in range(nfr)]
bs = np.array([len(f) for f in frames], np.uint64)
bo = np.concatenate([[0], np.cumsum(bs)[:-1]]).astype(np.uint64)
us = np.full(nfr, sz, np.uint64)
oo = np.arange(nfr, dtype=np.uint64) * sz
ctx = hip.Context(0)
ck = np.stack([np.frombuffer(ctx.blake3(e), dtype=np.uint8) for e in ents[:distinct]])
ck = ck[np.arange(nfr) % distinct]
d_blobs = torch.from_numpy(np.frombuffer(b"".join(frames) + bytes(64), dtype=np.uint8).copy()).cuda()
d_out = torch.zeros(nfr * sz + 64, dtype=torch.uint8, device="cuda")
rt = hip.RowTable(ctx, bo, bs, us, oo, None, ck)
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    c, corrupt, st = rt.decode_verify(d_blobs, d_out)
    dt = time.perf_counter() - t0
    kt = dict(ctx.kernel_times())
    print(f"rep {rep}: {dt * 1e3:.2f} ms ({nfr * sz / 2**20 / dt:.0f} MB/s) corrupt={c['corrupt_rows']} errs={c['decode_errors']} tables={kt.get('zstd_batch_tables', 0):.2f} huffman={kt.get('zstd_batch_huffman', 0):.2f} sequences={kt.get('zstd_batch_sequences', 0):.2f}+{kt.get('zstd_batch_sequences_long', 0):.2f} execute={kt.get('zstd_batch_execute', 0):.2f} fallback={kt.get('zstd_decode_fallback', 0):.2f}", rt.foreign_stats() if rep == 0 else "")
