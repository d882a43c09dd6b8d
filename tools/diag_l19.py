"""Diagnostic: which of the higher effort tier's own multi-block frames leave the block decoder for the two-phase path."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import workloads
from znippy_amd import hip
ents = workloads.image_corpus("text", 64e6)
lens = np.array([len(e) for e in ents], np.uint64)
src = np.frombuffer(b"".join(ents) + bytes(64), np.uint8)
offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
d_src = torch.from_numpy(src.copy()).cuda()
ctx = hip.Context(0); ctx.set_level(19)
rt = hip.RoundTable(ctx, offs, lens)
d_blob = torch.zeros(rt.blob_bound() + 64, dtype=torch.uint8, device="cuda")
enc = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in rt.encode_hash(d_src, d_blob).items()}
hb = d_blob.cpu().numpy()
big = [i for i in range(len(ents)) if lens[i] > 131072]
print(len(big), "multi-block rounds of", len(ents))
shown = 0
for i in big:
    rows = hip.RowTable(ctx, enc["blob_offset"][i:i+1], enc["blob_size"][i:i+1], lens[i:i+1], np.zeros(1, np.uint64), None, enc["checksum"][i:i+1])
    d_out = torch.zeros(int(lens[i]) + 64, dtype=torch.uint8, device="cuda")
    c, corrupt, st = rows.decode_verify(d_blob, d_out)
    fs = rows.foreign_stats()
    if fs["frames"] or c["corrupt_rows"] or c["decode_errors"]:
        f = hb[int(enc["blob_offset"][i]):int(enc["blob_offset"][i] + enc["blob_size"][i])]
        print(f"round {i}: {lens[i]} B -> {len(f)} B: foreign {fs} counters {c}")
        if shown < 3:
            shown += 1
            # walk the blocks
            fhd = f[4]; fcs_bytes = [0, 2, 4, 8][fhd >> 6] if (fhd >> 6) else (1 if (fhd >> 5) & 1 else 0)
            pos = 5 + (0 if (fhd >> 5) & 1 else 1) + fcs_bytes
            k = 0
            while pos + 3 <= len(f):
                bh = int(f[pos]) | int(f[pos+1]) << 8 | int(f[pos+2]) << 16
                last, typ, size = bh & 1, (bh >> 1) & 3, bh >> 3
                info = ""
                if typ == 2:
                    b0 = int(f[pos+3]); lt, sf = b0 & 3, (b0 >> 2) & 3
                    info = f"lit_type {lt} sf {sf}"
                print(f"   block {k}: type {typ} size {size} last {last} {info}")
                pos += 3 + (1 if typ == 1 else size); k += 1
                if last: break
            print("   end pos", pos, "of", len(f))
    rows.close()
