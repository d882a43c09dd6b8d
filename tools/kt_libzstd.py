import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, gen
import workloads
from znippy_amd import hip
n, sz = 100_000, 10240
ctx = hip.Context(0)
chunk = gen.text(sz)
for lvl in (19, 3):
    frame = np.frombuffer(workloads.libzstd_compress(chunk, lvl), dtype=np.uint8)
    fl = len(frame)
    d_blobs = torch.from_numpy(np.concatenate([np.tile(frame, n), np.zeros(64, np.uint8)])).cuda()
    ck = np.tile(np.frombuffer(ctx.blake3(chunk), dtype=np.uint8), (n, 1))
    rows = hip.RowTable(ctx, np.arange(n, dtype=np.uint64) * fl, np.full(n, fl, np.uint64), np.full(n, sz, np.uint64), np.arange(n, dtype=np.uint64) * sz, None, ck)
    d_out = torch.zeros(n * sz + 64, dtype=torch.uint8, device="cuda")
    acc = {}
    for i in range(12):
        rows.decode_verify_async(d_blobs, d_out)
        c, _, _ = rows.results(want_status=False)
        if i >= 2:
            for k, v in ctx.kernel_times():
                acc.setdefault(k, []).append(v)
    print("libzstd level", lvl, "frame", fl, "B", {k: round(float(np.mean(v)), 4) for k, v in acc.items()}, "verified", c["verified_bytes"] == n * sz)
