"""Diagnostic: per-kernel times of the C2 read step under ZNIPPY_DBG ablations (no assertions)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, gen
from znippy_amd import hip
n, sz = 100_000, 10240
ctx = hip.Context(0)
chunk = np.frombuffer(gen.text(sz), dtype=np.uint8)
d_src = torch.from_numpy(np.tile(chunk, n)).cuda()
lens = np.full(n, sz, np.uint64)
rounds = hip.RoundTable(ctx, np.arange(n, dtype=np.uint64) * sz, lens)
d_blob = torch.zeros(rounds.blob_bound() + 64, dtype=torch.uint8, device="cuda")
enc = rounds.encode_hash(d_src, d_blob)
rows = hip.RowTable(ctx, enc["blob_offset"], enc["blob_size"], lens, np.arange(n, dtype=np.uint64) * sz, None, enc["checksum"])
d_out = torch.zeros(n * sz + 64, dtype=torch.uint8, device="cuda")
acc = {}
for i in range(12):
    rows.decode_verify_async(d_blob, d_out)
    c, _, _ = rows.results(want_status=False)
    if i >= 2:
        for k, v in ctx.kernel_times():
            acc.setdefault(k, []).append(v)
print("DBG", os.environ.get("ZNIPPY_DBG", "0"), {k: round(float(np.mean(v)), 4) for k, v in acc.items()}, "verified", c["verified_bytes"] == n * sz)
