"""Same-process A/B of the role-split read kernel on the C2 libzstd-19 archive: two contexts (switches are read at creation),
runs interleaved, medians of the kernel's HIP-event time.  Usage: python tools/ab_tree.py [dbgA=0] [dbgB=524288] [pairs=30]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, gen, workloads
from znippy_amd import hip
dA = sys.argv[1] if len(sys.argv) > 1 else "0"
dB = sys.argv[2] if len(sys.argv) > 2 else "524288"
pairs = int(sys.argv[3]) if len(sys.argv) > 3 else 30
n, sz = 100000, 10240
chunk = gen.text(sz)
frame = np.frombuffer(workloads.libzstd_compress(chunk, 19), dtype=np.uint8)
fl = len(frame)
d_blobs = torch.from_numpy(np.concatenate([np.tile(frame, n), np.zeros(64, np.uint8)])).cuda()
d_out = torch.zeros(n * sz + 64, dtype=torch.uint8, device="cuda")
ctxs, tabs = [], []
from znippy_amd import _lib, _build
for d in (dA, dB):
    os.environ["ZNIPPY_DBG"] = d
    if d is dB and os.environ.get("ZN_LIB_B"):  # second side from another build of the library (compile-time variants)
        _lib._lib = None
        _build.SO = os.path.abspath(os.environ["ZN_LIB_B"])
    c = hip.Context(0)
    ck = np.tile(np.frombuffer(c.blake3(chunk), dtype=np.uint8), (n, 1))
    ctxs.append(c)
    tabs.append(hip.RowTable(c, np.arange(n, dtype=np.uint64) * fl, np.full(n, fl, np.uint64), np.full(n, sz, np.uint64),
                             np.arange(n, dtype=np.uint64) * sz, None, ck))
t = [[], []]
for i in range(pairs + 3):
    for j in (0, 1) if i % 2 == 0 else (1, 0):
        c, _, _ = tabs[j].decode_verify(d_blobs, d_out)
        assert c["corrupt_rows"] == 0 and c["verified_bytes"] == n * sz
        if i >= 3:
            t[j].append(dict(ctxs[j].kernel_times())["decode_verify_roles"])
for j, d in enumerate((dA, dB)):
    a = np.array(t[j])
    print(f"ZNIPPY_DBG={d}: median {np.median(a):.4f} ms  mean {a.mean():.4f}  min {a.min():.4f}  ({len(a)} runs)")
print(f"A/B median ratio {np.median(t[0]) / np.median(t[1]):.4f}")
