"""Diagnosis of the resolve path (k_rx_*): a handful of big frames, traced.  python tools/diag_rx.py [which]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import gen, workloads
from oracle import oracle as O
from znippy_amd import hip
O.build()
which = sys.argv[1] if len(sys.argv) > 1 else "all"
rng = np.random.default_rng(4)
p = rng.integers(32, 127, size=45, dtype=np.uint8).tobytes()
periodic = (p * (3 * 128 * 1024 // 45 + 1))[:3 * 128 * 1024]
inc = gen.incompressible(8, 2 * 128 * 1024 + 5000)
text = gen.pseudo_text(600_000, 3)
cases = {"periodic": (periodic, 3), "inc": (inc, 1), "text": (text, 3), "text19": (text, 19)}
entries, frames = [], []
for k, (d, lvl) in cases.items():
    if which in ("all", k):
        entries.append(d); frames.append(workloads.libzstd_compress(d, lvl))
        print(k, len(d), "->", len(frames[-1]), flush=True)
bs = np.array([len(f) for f in frames], np.uint64); bo = np.concatenate([[0], np.cumsum(bs)[:-1]]).astype(np.uint64)
us = np.array([len(e) for e in entries], np.uint64); oo = np.concatenate([[0], np.cumsum(us)[:-1]]).astype(np.uint64)
ck = np.stack([np.frombuffer(O.blake3(e), dtype=np.uint8) for e in entries])
d_blobs = torch.from_numpy(np.frombuffer(b"".join(frames) + bytes(64), dtype=np.uint8).copy()).cuda()
d_out = torch.zeros(int(us.sum()) + 64, dtype=torch.uint8, device="cuda")
ctx = hip.Context(0)
ctx.set_kernel_timing(2)
rt = hip.RowTable(ctx, bo, bs, us, oo, None, ck)
for rep in range(2):
    c, corrupt, status = rt.decode_verify(d_blobs, d_out)
    print(dict(c), status, dict(ctx.kernel_times()), rt.foreign_stats(), flush=True)
out = d_out[:int(us.sum())].cpu().numpy().tobytes()
print("bytes equal:", out == b"".join(entries))
