"""Diagnostic: write-side step time, synchronous results vs lagged results (two runs in flight)."""
import os, sys, time
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
def sync_steps(k):
    for _ in range(k):
        rounds.encode_hash_async(d_src, d_blob); rounds.results()
def lag_steps(k):
    rounds.encode_hash_async(d_src, d_blob)
    for _ in range(k - 1):
        rounds.encode_hash_async(d_src, d_blob); rounds.results_lagged(1)
    rounds.results_lagged(0)
def nores_steps(k):
    for _ in range(k):
        rounds.encode_hash_async(d_src, d_blob)
    rounds.results()
for name, f in (("sync", sync_steps), ("lagged", lag_steps), ("no-results", nores_steps), ("sync", sync_steps), ("lagged", lag_steps)):
    f(3); torch.cuda.synchronize()
    t0 = time.perf_counter(); f(20); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name:12s} {dt / 20 * 1e3:.4f} ms/step", {k: round(v, 4) for k, v in ctx.kernel_times()})
