"""Where the higher effort tier loses to libzstd -1 on the text corpus: compressed bytes by round size class."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import workloads
from znippy_amd import hip
kind = sys.argv[1] if len(sys.argv) > 1 else "text"
ents = workloads.image_corpus(kind, 80 << 20 if kind == "text" else 200e6)
lens = np.array([len(e) for e in ents], np.uint64)
src = np.frombuffer(b"".join(ents) + bytes(64), np.uint8)
offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
d_src = torch.from_numpy(src.copy()).cuda()
sizes = {}
for level in (1, 19):
    ctx = hip.Context(0); ctx.set_level(level)
    rt = hip.RoundTable(ctx, offs, lens)
    d_blob = torch.zeros(rt.blob_bound() + 64, dtype=torch.uint8, device="cuda")
    enc = rt.encode_hash(d_src, d_blob)
    sizes[level] = enc["blob_size"].astype(np.int64).copy()
    rt.close(); ctx.close()
z1 = np.array([len(workloads.libzstd_compress(e, 1)) for e in ents], np.int64)
z3 = np.array([len(workloads.libzstd_compress(e, 3)) for e in ents], np.int64)
L = lens.astype(np.int64)
print(f"{kind}: {len(ents)} rounds, {L.sum()/1e6:.1f} MB;  level1 {sizes[1].sum()/L.sum():.4f}  level19 {sizes[19].sum()/L.sum():.4f}  libzstd-1 {z1.sum()/L.sum():.4f}  libzstd-3 {z3.sum()/L.sum():.4f}")
for lo, hi in ((0, 1024), (1024, 4096), (4096, 16384), (16384, 65536), (65536, 131072), (131072, 1 << 40)):
    m = (L >= lo) & (L < hi)
    if not m.any(): continue
    print(f"  rounds {lo:>7}..{hi:<13} n={m.sum():5d} bytes {L[m].sum()/1e6:7.1f} MB  level19 {sizes[19][m].sum()/L[m].sum():.4f}  libzstd-1 {z1[m].sum()/L[m].sum():.4f}  "
          f"excess over libzstd-1: {(sizes[19][m].sum()-z1[m].sum())/L.sum()*100:.2f} % of the corpus")
