"""Diagnostic: wall time of each of the first write steps of the C2 workload (pipelined as bench.py's write leg does:
step k+1 queued before step k's results are read), to see where a cold start spends its time.
Usage: python tools/write_steps.py [read_steps_before]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import workloads
from znippy_amd import hip
wl = workloads.build("c2", torch)
lens = wl["lens"]; d_src = wl["d_src"]
offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
ctx = hip.Context(0)
ctx.set_kernel_timing(1)
rounds = hip.RoundTable(ctx, offs, lens, wl.get("skip"))
d_blob = torch.zeros(rounds.blob_bound() + 64, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
ts = []
pending = 0
t_prev = time.perf_counter()
sub, res = [], []
for i in range(30):
    rounds.encode_hash_async(d_src, d_blob)
    t1 = time.perf_counter()
    if pending:
        rounds.results_lagged(1)
    pending = 1
    now = time.perf_counter(); ts.append((now - t_prev) * 1e3); sub.append((t1 - t_prev) * 1e3); res.append((now - t1) * 1e3); t_prev = now
print("submit ms:", " ".join(f"{t:.2f}" for t in sub[:8]))
print("results ms:", " ".join(f"{t:.2f}" for t in res[:8]))
rounds.results_lagged(0)
print("ms per step:", " ".join(f"{t:.2f}" for t in ts))
