"""Diagnostic: per-kernel times of the read step on big rows (C3 shape: N x 8 MiB of periodic text, numpy-built so
that the script also runs under rocprofv3 --pmc).  Usage: python tools/kt_big.py [rows=64] [kind=text|random|stored]
(stored = random bytes on the store path: rows not compressed, hash + copy)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, gen
from znippy_amd import hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
kind = sys.argv[2] if len(sys.argv) > 2 else "text"
sz = 8 << 20
ctx = hip.Context(0)
if kind == "text":
    one = np.frombuffer(gen.text(sz), dtype=np.uint8)
    src = np.tile(one, n)
else:
    src = np.random.default_rng(1).integers(0, 256, size=n * sz, dtype=np.uint8)
d_src = torch.from_numpy(src).cuda()
lens = np.full(n, sz, np.uint64)
offs = np.arange(n, dtype=np.uint64) * sz
skip = np.ones(n, np.uint8) if kind == "stored" else None
rounds = hip.RoundTable(ctx, offs, lens, skip)
d_blob = torch.zeros(rounds.blob_bound() + 64, dtype=torch.uint8, device="cuda")
enc = rounds.encode_hash(d_src, d_blob)
bitmap = np.packbits(enc["compressed"].astype(bool), bitorder="little")
rows = hip.RowTable(ctx, enc["blob_offset"], enc["blob_size"], lens, offs, bitmap, enc["checksum"])
d_out = torch.zeros(n * sz + 64, dtype=torch.uint8, device="cuda")
acc = {}
for i in range(8):
    rows.decode_verify_async(d_blob, d_out)
    c, _, _ = rows.results(want_status=False)
    if i >= 2:
        for k, v in ctx.kernel_times():
            acc.setdefault(k, []).append(v)
print(kind, n, "rows x 8 MiB", {k: round(float(np.mean(v)), 4) for k, v in acc.items()}, "verified", c["verified_bytes"] == n * sz,
      "blob", int(enc["blob_size"].sum()))
