"""Diagnostic: where table construction time goes (100k x 10 KiB): RowTable / RoundTable creation, repeated."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from znippy_amd import hip
n, sz, fl = 100000, 10240, 64
ctx = hip.Context(0)
bo = np.arange(n, dtype=np.uint64) * fl; bs = np.full(n, fl, np.uint64); us = np.full(n, sz, np.uint64); oo = np.arange(n, dtype=np.uint64) * sz
ck = np.zeros((n, 32), np.uint8); bm = np.packbits(np.ones(n, bool), bitorder="little")
for rep in range(5):
    t0 = time.perf_counter(); rt = hip.RowTable(ctx, bo, bs, us, oo, bm, ck); t1 = time.perf_counter(); rt.close(); t2 = time.perf_counter()
    t3 = time.perf_counter(); rd = hip.RoundTable(ctx, oo, us); t4 = time.perf_counter(); rd.close(); t5 = time.perf_counter()
    print(f"rep {rep}: RowTable create {1e3*(t1-t0):.3f} ms close {1e3*(t2-t1):.3f} | RoundTable create {1e3*(t4-t3):.3f} ms close {1e3*(t5-t4):.3f}")
