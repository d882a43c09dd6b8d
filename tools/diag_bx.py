"""Batch path (k_bx_*) on N slices of real text, libzstd frames: correctness against the source + kernel times.
   python tools/diag_bx.py [n_rows] [slice_bytes] [level] [distinct]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch
import workloads
from znippy_amd import hip
from concurrent.futures import ThreadPoolExecutor

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
sz = int(sys.argv[2]) if len(sys.argv) > 2 else 10240
level = int(sys.argv[3]) if len(sys.argv) > 3 else 19
distinct = int(sys.argv[4]) if len(sys.argv) > 4 else min(n, 2048)
blob = b"".join(workloads.image_corpus("text", distinct * sz + (1 << 20), whole_files=False))
sl = [blob[i * sz:(i + 1) * sz] for i in range(distinct)]
t0 = time.time()
with ThreadPoolExecutor(16) as ex:
    fr = list(ex.map(lambda s: workloads.libzstd_compress(s, level), sl))
print(f"{distinct} frames compressed in {time.time()-t0:.1f}s, mean {np.mean([len(f) for f in fr]):.0f} B", flush=True)
idx = np.arange(n) % distinct
bs = np.array([len(fr[i]) for i in idx], np.uint64)
bo = np.concatenate([[0], np.cumsum(bs)[:-1]]).astype(np.uint64)
us = np.full(n, sz, np.uint64)
oo = (np.arange(n, dtype=np.uint64) * sz)
blobs = np.frombuffer(b"".join(fr[i] for i in idx) + bytes(64), dtype=np.uint8)
src = np.frombuffer(b"".join(sl), dtype=np.uint8)
ctx = hip.Context(0)
dig = np.stack([np.frombuffer(ctx.blake3(s), dtype=np.uint8) for s in sl])
ck = dig[idx]
d_blobs = torch.from_numpy(blobs.copy()).cuda()
d_out = torch.zeros(n * sz + 64, dtype=torch.uint8, device="cuda")
rt = hip.RowTable(ctx, bo, bs, us, oo, None, ck)
for rep in range(4):
    d_out.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    c, corrupt, status = rt.decode_verify(d_blobs, d_out)
    dt = time.perf_counter() - t0
    kt = {k: round(v, 3) for k, v in ctx.kernel_times() if v > 0.005}
    print(f"rep {rep}: {dt*1e3:.2f} ms = {n*sz/dt/1e9:.2f} GB/s  corrupt={c['corrupt_rows']} errs={c['decode_errors']} verified={c['verified_bytes']==n*sz}", kt, rt.foreign_stats() if rep == 0 else "", flush=True)
out = d_out[:n * sz].cpu().numpy().reshape(n, sz)
want = src.reshape(distinct, sz)[idx]
bad = np.nonzero((out != want).any(axis=1))[0]
print("rows with wrong bytes:", len(bad), bad[:10], "status!=0:", np.nonzero(status)[0][:10])
