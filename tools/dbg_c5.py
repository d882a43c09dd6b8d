import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench
from znippy_amd import hip
wl = bench.build_workload("c5", torch)
d_src, lens, skip = wl["d_src"], wl["lens"], wl["skip"]
n = len(lens); src_off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
ctx = hip.Context(0)
rt = hip.RoundTable(ctx, src_off, lens, skip)
print("rounds", n, "total", int(lens.sum()), "bound", rt.blob_bound())
d_blob = torch.zeros(rt.blob_bound() + 64, dtype=torch.uint8, device="cuda")
enc = rt.encode_hash(d_src, d_blob)
bo, bs = enc["blob_offset"].copy(), enc["blob_size"].copy()
print("blob_bytes", enc.get("blob_bytes"), "sum bs", int(bs.sum()), "last end", int(bo[-1] + bs[-1]))
exp = np.concatenate([[0], np.cumsum(bs)[:-1]]).astype(np.uint64)
bad = np.nonzero(bo != exp)[0]
print("offset mismatches", len(bad), bad[:5], bo[bad[:5]], exp[bad[:5]])
nb = 0
for i in range(n):
    if skip[i]:
        a = d_blob[int(bo[i]):int(bo[i] + bs[i])]; b = d_src[int(src_off[i]):int(src_off[i] + lens[i])]
        if bs[i] != lens[i] or not bool((a == b).all()):
            if nb < 5:
                neq = (a != b).nonzero()[:3].flatten().tolist() if bs[i] == lens[i] else None
                print("round", i, "len", lens[i], "bs", bs[i], "bo", bo[i], "src_off", src_off[i], "first diffs", neq)
            nb += 1
print("bad stored rounds", nb)
# detail of the first bad round
i = next(i for i in range(n) if skip[i] and not bool((d_blob[int(bo[i]):int(bo[i] + bs[i])] == d_src[int(src_off[i]):int(src_off[i] + lens[i])]).all()))
a = d_blob[int(bo[i]):int(bo[i] + bs[i])]; b = d_src[int(src_off[i]):int(src_off[i] + lens[i])]
ne = (a != b)
idx = ne.nonzero().flatten()
print("round", i, "n diff bytes", int(ne.sum()), "first", int(idx[0]), "last", int(idx[-1]))
k = int(idx[0])
print("blob bytes", a[k:k + 16].tolist(), "src bytes", b[k:k + 16].tolist(), "zero?", bool((a[idx] == 0).all()))
# runs of differing bytes
d = idx[1:] - idx[:-1]
brk = (d != 1).nonzero().flatten()
print("runs", len(brk) + 1, "first run len", (int(brk[0]) + 1) if len(brk) else len(idx))
# second encode: same bad set?
enc2 = rt.encode_hash(d_src, d_blob)
a2 = d_blob[int(bo[i]):int(bo[i] + bs[i])]
print("after 2nd encode same round ok?", bool((a2 == b).all()))
