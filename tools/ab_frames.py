"""Same-process comparison of the C2 read kernel on the libzstd-19 archive vs this build's own archive (runs interleaved)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, gen, workloads
from znippy_amd import hip
n, sz = 100000, 10240
chunk = gen.text(sz)
ctx = hip.Context(0)
ck = np.tile(np.frombuffer(ctx.blake3(chunk), dtype=np.uint8), (n, 1))
d_out = torch.zeros(n * sz + 64, dtype=torch.uint8, device="cuda")
tabs, blobs = [], []
for kind in ("libzstd19", "own"):
    frame = np.frombuffer(workloads.libzstd_compress(chunk, 19) if kind == "libzstd19" else ctx.compress(chunk), dtype=np.uint8)
    fl = len(frame)
    blobs.append(torch.from_numpy(np.concatenate([np.tile(frame, n), np.zeros(64, np.uint8)])).cuda())
    tabs.append(hip.RowTable(ctx, np.arange(n, dtype=np.uint64) * fl, np.full(n, fl, np.uint64), np.full(n, sz, np.uint64),
                             np.arange(n, dtype=np.uint64) * sz, None, ck))
    print(kind, "frame bytes", fl)
t = [[], []]
for i in range(43):
    for j in (0, 1) if i % 2 == 0 else (1, 0):
        c, _, _ = tabs[j].decode_verify(blobs[j], d_out)
        assert c["corrupt_rows"] == 0
        if i >= 3:
            t[j].append(dict(ctx.kernel_times())["decode_verify_roles"])
for j, k in enumerate(("libzstd19", "own")):
    print(f"{k}: median {np.median(t[j]):.4f} ms min {min(t[j]):.4f}")
