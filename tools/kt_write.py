"""Diagnostic: per-kernel times of the C2 write step (no assertions)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, gen, time
from znippy_amd import hip
n, sz = 100_000, 10240
ctx = hip.Context(0)
chunk = np.frombuffer(gen.text(sz), dtype=np.uint8)
d_src = torch.from_numpy(np.tile(chunk, n)).cuda()
rounds = hip.RoundTable(ctx, np.arange(n, dtype=np.uint64) * sz, np.full(n, sz, np.uint64))
d_blob = torch.zeros(rounds.blob_bound() + 64, dtype=torch.uint8, device="cuda")
acc, wall = {}, []
for i in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rounds.encode_hash_async(d_src, d_blob); r = rounds.results()
    wall.append((time.perf_counter() - t0) * 1e3)
    if i >= 2:
        for k, v in ctx.kernel_times(): acc.setdefault(k, []).append(v)
print("NOHASH" if os.environ.get("ZNIPPY_NOHASH") else "full", "wall", round(float(np.median(wall[2:])), 3), {k: round(float(np.mean(v)), 4) for k, v in acc.items()}, "blob", r["blob_bytes"])
# hash alone (no encoder next to it)
ts = []
for i in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rounds.hash(d_src)
    ts.append((time.perf_counter() - t0) * 1e3)
print("hash_rounds alone wall", round(float(np.median(ts[2:])), 3), dict(ctx.kernel_times()))
