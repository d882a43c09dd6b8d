"""Encoder ratio on the image's own files (same corpus as tests/realdata_report.py), both effort tiers, with the frames
checked by libzstd (sample) and by the GPU decoder (all).  Usage: python tools/ratio.py [binary_cap_MB]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import workloads
from workloads import image_corpus as corpus
from znippy_amd import hip

cap_bin = float(sys.argv[1]) * 1e6 if len(sys.argv) > 1 else 300e6
for kind, cap in (("text", 64e6), ("binary", cap_bin)):
    ents = corpus(kind, cap)
    lens = np.array([len(e) for e in ents], np.uint64)
    total = int(lens.sum())
    src = np.frombuffer(b"".join(ents) + bytes(64), np.uint8)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    d_src = torch.from_numpy(src.copy()).cuda()
    for level in (1, 19):
        ctx = hip.Context(0)
        ctx.set_level(level)
        rt = hip.RoundTable(ctx, offs, lens)
        d_blob = torch.zeros(rt.blob_bound() + 64, dtype=torch.uint8, device="cuda")
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            enc = rt.encode_hash(d_src, d_blob)
            ts.append(time.perf_counter() - t0)
        enc = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in enc.items()}
        hb = d_blob.cpu().numpy()
        bad = 0
        for i in sorted(set(np.argsort(lens)[-4:].tolist() + list(range(0, len(ents), max(1, len(ents) // 60))))):
            f = hb[int(enc["blob_offset"][i]):int(enc["blob_offset"][i] + enc["blob_size"][i])].tobytes()
            try:
                ok = workloads.libzstd_decompress(f, max(int(lens[i]), 1)) == ents[i]
            except Exception as e:
                ok = False
            bad += 0 if ok else 1
        rows = hip.RowTable(ctx, enc["blob_offset"], enc["blob_size"], lens, offs, None, enc["checksum"])
        d_out = torch.zeros(total + 64, dtype=torch.uint8, device="cuda")
        td = []
        for _ in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            c, corrupt, st = rows.decode_verify(d_blob, d_out)
            td.append(time.perf_counter() - t0)
        same = bool((d_out[:total] == d_src[:total]).all())
        kt = {k: round(v, 2) for k, v in dict(ctx.kernel_times()).items() if v > 0.05}
        import hashlib
        dig = hashlib.sha1(hb[:int(enc['blob_size'].sum())].tobytes()).hexdigest()[:16]
        print(f"[{kind}] level {level}: blob sha1 {dig} ratio {int(enc['blob_size'].sum())/total:.4f}  encode {min(ts)*1e3:.1f} ms ({total/2**20/min(ts):.0f} MB/s)  "
              f"libzstd-bad {bad}  gpu decode {min(td)*1e3:.1f} ms same={same} corrupt={c['corrupt_rows']} errs={c['decode_errors']} {kt}", flush=True)
        rows.close(); rt.close(); ctx.close()
