"""Diagnostic: do a decode-only launch and a hash-only launch overlap when submitted to two streams?"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, gen
from znippy_amd import hip
n, sz = 100_000, 10240
ctxA, ctxB = hip.Context(0), hip.Context(0)
chunk = np.frombuffer(gen.text(sz), dtype=np.uint8)
d_src = torch.from_numpy(np.tile(chunk, n)).cuda()
lens = np.full(n, sz, np.uint64)
rounds = hip.RoundTable(ctxA, np.arange(n, dtype=np.uint64) * sz, lens)
d_blob = torch.zeros(rounds.blob_bound() + 64, dtype=torch.uint8, device="cuda")
enc = rounds.encode_hash(d_src, d_blob)
enc = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in enc.items()}
off = np.arange(n, dtype=np.uint64) * sz
rowsA = hip.RowTable(ctxA, enc["blob_offset"], enc["blob_size"], lens, off, None, enc["checksum"])
rowsB = hip.RowTable(ctxB, enc["blob_offset"], enc["blob_size"], lens, off, None, enc["checksum"])
outA = torch.zeros(n * sz + 64, dtype=torch.uint8, device="cuda")
outB = torch.zeros(n * sz + 64, dtype=torch.uint8, device="cuda")
def run(dbgA, dbgB, both=True):
    ts = []
    for i in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        os.environ["ZNIPPY_DBG"] = str(dbgA); rowsA.decode_verify_async(d_blob, outA)
        if both:
            os.environ["ZNIPPY_DBG"] = str(dbgB); rowsB.decode_verify_async(d_blob, outB)
        ctxA.sync(); ctxB.sync()
        ts.append((time.perf_counter() - t0) * 1e3)
    return round(float(np.median(ts[2:])), 3)
print("decode-only alone      ", run(1, 0, both=False))
print("hash-only alone        ", run(2, 0, both=False))
print("full alone             ", run(0, 0, both=False))
print("decode-only || hash-only", run(1, 2))
print("full || full (2x work) ", run(0, 0))
