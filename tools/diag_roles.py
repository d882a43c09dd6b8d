"""Diagnostic: which rows does the role-split kernel get wrong (digest / bytes), and how are they distributed."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, gen
import workloads
from znippy_amd import hip
n, sz = int(os.environ.get("N", 100000)), 10240
ctx = hip.Context(0)
chunk = gen.text(sz)
frame = np.frombuffer(workloads.libzstd_compress(chunk, 19), dtype=np.uint8)
fl = len(frame)
d_blobs = torch.from_numpy(np.concatenate([np.tile(frame, n), np.zeros(64, np.uint8)])).cuda()
want = np.frombuffer(ctx.blake3(chunk), dtype=np.uint8)
ck = np.tile(want, (n, 1))
rt = hip.RowTable(ctx, np.arange(n, dtype=np.uint64) * fl, np.full(n, fl, np.uint64), np.full(n, sz, np.uint64),
                  np.arange(n, dtype=np.uint64) * sz, None, ck)
d_out = torch.zeros(n * sz + 64, dtype=torch.uint8, device="cuda")
for rep in range(3):
    d_out.zero_()
    c, corrupt, status = rt.decode_verify(d_blobs, d_out)
    dg = rt.digests()
    bad = np.nonzero((dg != want[None, :]).any(axis=1))[0]
    ref = torch.from_numpy(np.frombuffer(chunk, dtype=np.uint8).copy()).cuda()
    bytes_bad = torch.nonzero((d_out[:n * sz].view(n, sz) != ref[None, :]).any(dim=1)).flatten().cpu().numpy()
    print(f"rep {rep}: counters {c['corrupt_rows']} corrupt, digest-bad {len(bad)}, bytes-bad {len(bytes_bad)}, status!=0 {int((status != 0).sum())}")
    if len(bad):
        tiles = bad // 6
        print("  bad rows head:", bad[:24].tolist())
        print("  row%6 hist:", np.bincount(bad % 6, minlength=6).tolist(), " tile%4 hist:", np.bincount(tiles % 4, minlength=4).tolist())
        ut, cnt = np.unique(tiles, return_counts=True)
        print("  bad tiles:", len(ut), " rows-per-bad-tile hist:", np.bincount(cnt, minlength=7).tolist())
        print("  group(=tile//4) count:", len(np.unique(tiles // 4)), " distinct digests among bad:", len(np.unique(dg[bad], axis=0)))
    if len(bytes_bad):
        print("  bytes-bad rows head:", bytes_bad[:24].tolist())
print(ctx.kernel_times())
print("shader GHz during the read kernel:", ctx.last_shader_ghz())
if os.environ.get("UBENCH"):
    print("blake3 pass ns:", ctx.blake3_pass_ns(), "at GHz", ctx.ubench_ghz)
