"""Diagnostic: the read path on libzstd level-19 frames of the image's source text (one frame per file): per-kernel
times, the general decoder's phase stamps (run with ZNIPPY_DDBG=1), size classes.  Usage: diag_text19.py [cap_MB] [level]"""
import os, sys, time
from multiprocessing import Pool
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import workloads

def _c(args):
    return workloads.libzstd_compress(args[0], args[1])

if __name__ == "__main__":
    cap = float(sys.argv[1]) * 1e6 if len(sys.argv) > 1 else 81e6
    level = int(sys.argv[2]) if len(sys.argv) > 2 else 19
    ents = workloads.image_corpus("text", cap)
    with Pool(16) as p:
        frames = p.map(_c, [(e, level) for e in ents], chunksize=8)
    import torch
    from znippy_amd import hip
    lens = np.array([len(e) for e in ents], np.uint64)
    bs = np.array([len(f) for f in frames], np.uint64)
    bo = np.concatenate([[0], np.cumsum(bs)[:-1]]).astype(np.uint64)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    total = int(lens.sum())
    ctx = hip.Context(0)
    ck = np.stack([np.frombuffer(ctx.blake3(e), np.uint8) for e in ents])
    d_blob = torch.from_numpy(np.frombuffer(b"".join(frames) + bytes(64), np.uint8).copy()).cuda()
    d_out = torch.zeros(total + 64, dtype=torch.uint8, device="cuda")
    def run(sel, label):
        rows = hip.RowTable(ctx, bo[sel], bs[sel], lens[sel], offs[sel], None, ck[sel])
        ts = []
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            c, corrupt, st = rows.decode_verify(d_blob, d_out)
            ts.append(time.perf_counter() - t0)
        nb = int(lens[sel].sum())
        kt = {k: round(v, 2) for k, v in dict(ctx.kernel_times()).items() if v > 0.03}
        print(f"{label}: {len(sel)} frames, {nb/1e6:.1f} MB in {min(ts)*1e3:.2f} ms ({nb/2**20/min(ts):.0f} MB/s) corrupt={c['corrupt_rows']} errs={c['decode_errors']} {kt}", flush=True)
        rows.close()
    idx = np.arange(len(ents))
    run(idx, "all")
    for lo, hi in ((0, 4096), (4096, 16384), (16384, 131073), (131073, 1 << 40)):
        sel = idx[(lens >= lo) & (lens < hi)]
        if len(sel): run(sel, f"usize {lo}..{hi}")
