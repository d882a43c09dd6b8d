"""Diagnostic: libzstd frames of PERIODIC data in 8 MiB rows (every block one long match that starts in the previous block):
the two-phase path vs the serial wide decoder (ZNIPPY_NO_FZ=1).  Usage: python tools/diag_fz_periodic.py [rows=16] [level=3]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, gen, workloads
from znippy_amd import hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
level = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sz = 8 << 20
data = gen.text(sz)
frame = workloads.libzstd_compress(data, level)
print("frame bytes", len(frame))
ctx = hip.Context(0)
ck = np.tile(np.frombuffer(ctx.blake3(data), dtype=np.uint8), (n, 1))
fl = len(frame)
d_blobs = torch.from_numpy(np.concatenate([np.tile(np.frombuffer(frame, np.uint8), n), np.zeros(64, np.uint8)])).cuda()
d_out = torch.zeros(n * sz + 64, dtype=torch.uint8, device="cuda")
rt = hip.RowTable(ctx, np.arange(n, dtype=np.uint64) * fl, np.full(n, fl, np.uint64), np.full(n, sz, np.uint64), np.arange(n, dtype=np.uint64) * sz, None, ck)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    c, corrupt, st = rt.decode_verify(d_blobs, d_out)
    dt = time.perf_counter() - t0
    print(f"rep {rep}: {dt*1e3:.2f} ms ({n*sz/2**20/dt:.0f} MB/s) corrupt={c['corrupt_rows']} errs={c['decode_errors']}", {k: round(v, 3) for k, v in ctx.kernel_times() if v > 0.05}, rt.foreign_stats() if rep == 0 else "")
