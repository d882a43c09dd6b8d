"""One frame through the batch path: where do the decoded bytes first differ from the source?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch
import workloads, gen
from test_gpu_foreign import _mixed
from znippy_amd import hip

level = int(sys.argv[1]) if len(sys.argv) > 1 else 1
data = _mixed(900_000, 1)
f = workloads.libzstd_compress(data, level)
# block map
fhd = f[4]; single = (fhd >> 5) & 1; fcsf = fhd >> 6
pos = 5 + (0 if single else 1) + [single, 2, 4, 8][fcsf]
blocks = []; out = 0
while True:
    bh = f[pos] | f[pos + 1] << 8 | f[pos + 2] << 16
    last, t, sz = bh & 1, (bh >> 1) & 3, bh >> 3
    info = dict(pos=pos, type=t, size=sz, out0=out)
    if t == 2:
        b0 = f[pos + 3]; info["lit_type"] = b0 & 3
    blocks.append(info)
    pos += 3 + (1 if t == 1 else sz)
    if t != 2: out += sz
    else: out = None
    if out is None: out = -1
    if last: break
ctx = hip.Context(0)
d_blobs = torch.from_numpy(np.frombuffer(f + bytes(64), dtype=np.uint8).copy()).cuda()
d_out = torch.zeros(len(data) + 64, dtype=torch.uint8, device="cuda")
ck = np.frombuffer(ctx.blake3(data), dtype=np.uint8)[None, :]
rt = hip.RowTable(ctx, np.array([0], np.uint64), np.array([len(f)], np.uint64), np.array([len(data)], np.uint64), np.array([0], np.uint64), None, ck)
c, corrupt, status = rt.decode_verify(d_blobs, d_out)
got = d_out[:len(data)].cpu().numpy(); want = np.frombuffer(data, dtype=np.uint8)
bad = np.nonzero(got != want)[0]
print("counters", c, "status", status, dict(ctx.kernel_times()).keys(), rt.foreign_stats())
print("n blocks", len(blocks), [(b["type"], b.get("lit_type")) for b in blocks])
print("mismatches", len(bad), bad[:20], "first block index by 128K:", bad[0] // 131072 if len(bad) else None)
if len(bad):
    i = int(bad[0]); print("got ", bytes(got[i - 8:i + 24])); print("want", bytes(want[i - 8:i + 24]))
    # runs of mismatches
    d = np.diff(bad); starts = np.concatenate([[bad[0]], bad[1:][d > 1]]); print("runs:", len(starts), starts[:20])
