"""Diagnostic: time to build a row table / a round table for the C2 shape (100k entries)."""
import os, sys, time
import numpy as np
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, gen
from znippy_amd import hip
n, sz = 100_000, 10240
ctx = hip.Context(0)
lens = np.full(n, sz, np.uint64)
bo = np.arange(n, dtype=np.uint64) * 85; bs = np.full(n, 85, np.uint64)
oo = np.arange(n, dtype=np.uint64) * sz
ck = np.zeros((n, 32), np.uint8)
torch.cuda.synchronize()
for i in range(5):
    t0 = time.perf_counter()
    rows = hip.RowTable(ctx, bo, bs, lens, oo, None, ck)
    torch.cuda.synchronize()
    print("rows_create ms", round((time.perf_counter() - t0) * 1e3, 3))
    del rows
for i in range(4):
    t0 = time.perf_counter(); rounds = hip.RoundTable(ctx, oo, lens); torch.cuda.synchronize()
    print("rounds_create ms", round((time.perf_counter() - t0) * 1e3, 3))
    del rounds
