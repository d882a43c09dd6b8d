"""(Lives under tests/ because it times the oracle's CPU loops beside the GPU path; not collected by pytest.)
Real-data check (not a BASELINE config): source text + shared objects found in the image, one Round per file
(8 MiB slices), GPU encoder ratio/time vs libzstd, GPU decode+verify time on both kinds of frames."""
import os, sys, time, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import oracle as O
from znippy_amd import hip

from workloads import image_corpus as corpus

def run(kind, cap):
    ents = corpus(kind, cap)
    lens = np.array([len(e) for e in ents], np.uint64)
    total = int(lens.sum())
    src = np.frombuffer(b"".join(ents) + bytes(64), np.uint8)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    d_src = torch.from_numpy(src.copy()).cuda()
    ctx = hip.Context(0)
    rt = hip.RoundTable(ctx, offs, lens)
    d_blob = torch.zeros(rt.blob_bound() + 64, dtype=torch.uint8, device="cuda")
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        enc = rt.encode_hash(d_src, d_blob)
        ts.append(time.perf_counter() - t0)
    enc = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in enc.items()}
    gpu_bytes = int(enc["blob_size"].sum())
    # independent check: libzstd decodes this encoder's frames (largest rounds + a sample)
    hb = d_blob.cpu().numpy()
    for i in sorted(set(np.argsort(lens)[-6:].tolist() + list(range(0, len(ents), max(1, len(ents) // 40))))):
        f = hb[int(enc["blob_offset"][i]):int(enc["blob_offset"][i] + enc["blob_size"][i])].tobytes()
        assert O.libzstd_decompress(f, max(int(lens[i]), 1)) == ents[i], f"libzstd disagrees on round {i}"
    print(f"[{kind}] {len(ents)} rounds, {total/1e6:.1f} MB; median round {int(np.median(lens))} B")
    print(f"[{kind}] GPU encode+hash: {min(ts)*1e3:.2f} ms ({total/2**20/min(ts):.0f} MB/s), ratio {gpu_bytes/total:.3f}", dict(ctx.kernel_times()))
    res = {}
    for lvl in (1, 3, 19):
        t0 = time.perf_counter()
        r = O.compress_rounds(src, offs, lens, np.zeros(len(ents), np.uint8), level=lvl, n_threads=min(64, os.cpu_count()))
        dt = time.perf_counter() - t0
        res[lvl] = r
        print(f"[{kind}] libzstd -{lvl}: ratio {int(r['blob_size'].sum())/total:.3f}  ({total/2**20/dt:.0f} MB/s on {min(64, os.cpu_count())} threads)")
    d_out = torch.zeros(total + 64, dtype=torch.uint8, device="cuda")
    for name, bo, bs, blobs, ck in (("gpu frames", enc["blob_offset"], enc["blob_size"], d_blob, enc["checksum"]),
                                    ("libzstd-19 frames", res[19]["blob_offset"], res[19]["blob_size"],
                                     torch.from_numpy(np.concatenate([res[19]["blobs"], np.zeros(64, np.uint8)])).cuda(), res[19]["checksum"])):
        rows = hip.RowTable(ctx, bo, bs, lens, offs, None, ck)
        ts = []
        for _ in range(4):
            d_out.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
            c, corrupt, st = rows.decode_verify(blobs, d_out)
            ts.append(time.perf_counter() - t0)
        ok = bool((d_out[:total] == d_src[:total]).all())
        print(f"[{kind}] GPU decode+verify of {name}: {min(ts)*1e3:.2f} ms ({total/2**20/min(ts):.0f} MB/s) ok={ok} corrupt={c['corrupt_rows']} errs={c['decode_errors']}",
              dict(ctx.kernel_times()), "foreign path:", rows.foreign_stats())
    # where the time goes: the many small frames vs the single largest one (one workgroup decodes one frame)
    bo, bs, ck = res[19]["blob_offset"], res[19]["blob_size"], res[19]["checksum"]
    blobs19 = torch.from_numpy(np.concatenate([res[19]["blobs"], np.zeros(64, np.uint8)])).cuda()
    for label, sel in (("rounds <= 64 KiB", np.nonzero(lens <= 65536)[0]), ("largest round", np.array([int(np.argmax(lens))]))):
        if len(sel) == 0:
            continue
        rows = hip.RowTable(ctx, bo[sel], bs[sel], lens[sel], offs[sel], None, ck[sel])
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            c, corrupt, st = rows.decode_verify(blobs19, d_out)
            ts.append(time.perf_counter() - t0)
        nb = int(lens[sel].sum())
        print(f"[{kind}]   libzstd-19 frames, {label}: {len(sel)} rows, {nb/1e6:.1f} MB in {min(ts)*1e3:.2f} ms ({nb/2**20/min(ts):.0f} MB/s) errs={c['decode_errors']}")
    t0 = time.perf_counter()
    O.decompress_rows(res[19]["blobs"], res[19]["blob_offset"], res[19]["blob_size"], lens, offs, np.packbits(np.ones(len(ents), bool), bitorder="little"), res[19]["checksum"], 0, len(ents), n_threads=min(64, os.cpu_count()), use_libzstd=True)
    dt = time.perf_counter() - t0
    print(f"[{kind}] CPU oracle read loop on libzstd-19 frames: {total/2**20/dt:.0f} MB/s")

cap = int(sys.argv[1]) << 20 if len(sys.argv) > 1 else 128 << 20
bcap = int(sys.argv[2]) << 20 if len(sys.argv) > 2 else cap
for kind in ("text", "binary"):
    run(kind, cap if kind == "text" else bcap)
