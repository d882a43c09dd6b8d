"""Diagnostic: many small foreign frames (python sources of the image, one libzstd frame per file <= 64 KiB) through the
read step; with ZNIPPY_DDBG=1 the library prints the general decoder's phase shares.  Usage: python tools/diag_small.py [level=19] [max_files=2000]"""
import glob, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, workloads
from znippy_amd import hip
level = int(sys.argv[1]) if len(sys.argv) > 1 else 19
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
ents = []
for f in sorted(glob.glob("/usr/lib/python3.10/**/*.py", recursive=True)):
    if os.path.isfile(f):
        b = open(f, "rb").read()
        if 0 < len(b) <= 65536:
            ents.append(b)
    if len(ents) >= cap:
        break
frames = [workloads.libzstd_compress(e, level) for e in ents]
n = len(ents)
us = np.array([len(e) for e in ents], np.uint64); bs = np.array([len(f) for f in frames], np.uint64)
oo = np.concatenate([[0], np.cumsum(us)[:-1]]).astype(np.uint64); bo = np.concatenate([[0], np.cumsum(bs)[:-1]]).astype(np.uint64)
total = int(us.sum())
print(f"{n} files, {total/1e6:.1f} MB, ratio {bs.sum()/total:.3f}, median {int(np.median(us))} B")
ctx = hip.Context(0)
ck = np.stack([np.frombuffer(ctx.blake3(e), dtype=np.uint8) for e in ents])
d_blobs = torch.from_numpy(np.frombuffer(b"".join(frames) + bytes(64), dtype=np.uint8).copy()).cuda()
d_out = torch.zeros(total + 64, dtype=torch.uint8, device="cuda")
rt = hip.RowTable(ctx, bo, bs, us, oo, None, ck)
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    c, corrupt, st = rt.decode_verify(d_blobs, d_out)
    dt = time.perf_counter() - t0
    print(f"rep {rep}: {dt*1e3:.2f} ms ({total/2**20/dt:.0f} MB/s) corrupt={c['corrupt_rows']} errs={c['decode_errors']}", {k: round(v, 3) for k, v in ctx.kernel_times()})
