#!/usr/bin/env python3
"""bench.py — BASELINE.json's headline metric on MI355X.

A "step" = one pass of the hot path over one rank's batch of index rows: decode-or-passthrough
+ BLAKE3 + verify of every row (the body of decompress.rs:L135-190 for the whole row range),
inputs (blob region, index columns) resident in HBM when the timed region starts.  A second
timed loop measures the write side (BLAKE3 + zstd encode of every Round, stream_packer.rs:L217-284).

Workload (config.workload): BASELINE configs[1] — 100,000 x 10,240-byte text chunks
(perf_bench.rs:L133-141) PER RANK; with N ranks the archive has N x 100k rows and the row cursor is
split into N contiguous ranges (weak scaling, no data-path collective; counters are summed with
one RCCL all-reduce per step).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def build_workload(name, O, gen):
    """Returns (entries_desc) = dict(frames=[bytes], usize=[int], repeat=n, chunk=bytes...)."""
    if name == "c2":
        chunk = gen.text(10 * 1024)
        return dict(chunk=chunk, n=100_000, name="100k x 10KiB text chunks (BASELINE configs[1])")
    if name == "c2small":
        chunk = gen.text(10 * 1024)
        return dict(chunk=chunk, n=2_000, name="2k x 10KiB text chunks (reduced; NOT the headline config)")
    raise SystemExit(f"unknown workload {name}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import gen
    from oracle import oracle as O  # checker + cpu_baseline leg only
    from znippy_amd import hip

    wl = build_workload(args.workload, O, gen)
    chunk, n = wl["chunk"], wl["n"]
    sz = len(chunk)
    ctx = hip.Context(local_rank)

    # ---- write side inputs: this rank's Rounds over a resident staging buffer ----
    chunk_np = np.frombuffer(chunk, dtype=np.uint8)
    d_src = torch.from_numpy(np.tile(chunk_np, n)).cuda()
    src_off = np.arange(n, dtype=np.uint64) * sz
    lens = np.full(n, sz, dtype=np.uint64)
    rounds = hip.RoundTable(ctx, src_off, lens)
    have_encoder = True
    d_blob = torch.zeros(rounds.blob_bound() + 64, dtype=torch.uint8, device="cuda")
    try:
        enc = rounds.encode_hash(d_src, d_blob)
    except Exception as e:  # encoder not built yet: fall back to CPU-made frames for the read side
        have_encoder = False
        enc_err = str(e)

    want_digest = np.frombuffer(O.blake3(chunk), dtype=np.uint8)
    if have_encoder:
        assert (enc["checksum"] == want_digest[None, :]).all(), "GPU write-side digests != oracle"
        bo, bs, comp = enc["blob_offset"], enc["blob_size"], enc["compressed"]
        ck = enc["checksum"]
        # parity spot check: frames decode with the oracle
        host_blob = d_blob[:enc["blob_bytes"]].cpu().numpy()
        for i in (0, n // 2, n - 1):
            f = host_blob[int(bo[i]):int(bo[i] + bs[i])].tobytes()
            assert O.zstd_decompress(f) == chunk, "GPU frame does not decode on the oracle"
        d_blobs = d_blob
        archive_src = "gpu-encoded (this build's zstd frames)"
    else:
        frame = np.frombuffer(O.libzstd_compress(chunk, 19), dtype=np.uint8)
        fl = len(frame)
        d_blobs = torch.from_numpy(np.concatenate([np.tile(frame, n), np.zeros(64, np.uint8)])).cuda()
        bo = np.arange(n, dtype=np.uint64) * fl
        bs = np.full(n, fl, dtype=np.uint64)
        comp = np.ones(n, dtype=np.uint8)
        ck = np.tile(want_digest, (n, 1))
        archive_src = "cpu libzstd-19 frames (GPU encoder unavailable: %s)" % enc_err

    out_off = np.arange(n, dtype=np.uint64) * sz
    d_out = torch.zeros(n * sz + 64, dtype=torch.uint8, device="cuda")
    rows = hip.RowTable(ctx, bo, bs, lens, out_off, np.packbits(comp.astype(bool), bitorder="little"), ck)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    cvec = torch.zeros(8, dtype=torch.int64, device="cuda")

    def read_step():
        rows.decode_verify_async(d_blobs, d_out)
        counters, corrupt, _ = rows.results(want_status=False)
        if world > 1:  # the only cross-GPU traffic: one small all-reduce of the counters
            cvec[:6] = torch.tensor([counters[k] for k in ("total_chunks", "total_written_bytes", "verified_bytes",
                                                           "corrupt_bytes", "corrupt_rows", "decode_errors")],
                                    dtype=torch.int64, device="cuda")
            dist.all_reduce(cvec)
        return counters

    def write_step():
        rounds.encode_hash_async(d_src, d_blob)
        return rounds.results()

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        barrier()
        t0 = time.perf_counter()
        ktimes = {}
        for _ in range(steps):
            fn()
            for name, ms in ctx.kernel_times():
                ktimes.setdefault(name, []).append(ms)
        barrier()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()), {k: float(np.mean(v)) for k, v in ktimes.items()}

    # ---- read side (the headline) ----
    counters = read_step()
    assert counters["corrupt_rows"] == 0 and counters["decode_errors"] == 0 and \
        counters["verified_bytes"] == n * sz, counters
    ref = torch.from_numpy(chunk_np.copy()).cuda()
    assert bool((d_out[:n * sz].view(n, sz)[:: max(1, n // 997)] == ref[None, :]).all()), "decoded bytes differ"
    dt_read, k_read = timed(read_step, args.steps, args.warmup)
    dt_write, k_write = (timed(write_step, args.steps, args.warmup) if have_encoder else (None, {}))

    total_bytes = n * sz * world
    mbps_read = total_bytes / 2**20 / (dt_read / args.steps)
    mbps_write = total_bytes / 2**20 / (dt_write / args.steps) if dt_write else None

    if rank == 0:
        # roofline of the dominant read-side kernel, algorithmic bytes per launch
        blob_bytes = int(bs.sum())
        alg = {
            "zstd_decode": blob_bytes + n * sz + 57 * n,        # read frame, write output, index columns
            "blake3_tiles": n * sz + 32 * n,                      # read every decoded byte once (unfused pass)
            "decode_verify_fused": blob_bytes + n * sz + 57 * n,  # SURVEY §8d figure for the whole path
        }
        dom = max((k for k in k_read if k in alg), key=lambda k: k_read[k]) if k_read else None
        roofline = None
        if dom:
            ach = alg[dom] / (k_read[dom] * 1e-3) / 1e9
            path_ms = sum(k_read.values())
            roofline = dict(bound="hbm", kernel=dom, achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                            frac=round(ach / HBM_PEAK_GBS, 4), traffic=None,
                            kernel_ms={k: round(v, 4) for k, v in k_read.items()},
                            path_achieved=round(alg["decode_verify_fused"] / (path_ms * 1e-3) / 1e9, 1),
                            path_frac=round(alg["decode_verify_fused"] / (path_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cores = len(os.sched_getaffinity(0))
            threads = max(1, int(np.ceil(0.9 * cores)))  # common_config.rs:L34 rule on what we can see
            host_blobs = d_blobs.cpu().numpy()
            bitmap = np.packbits(comp.astype(bool), bitorder="little")
            host_out = np.zeros(n * sz, dtype=np.uint8)
            t0 = time.perf_counter()
            st, _ = O.decompress_rows(host_blobs, bo, bs, lens, out_off, bitmap, ck, 0, n, out=host_out,
                                      n_threads=threads, use_libzstd=O.have_libzstd())
            dt_cpu = time.perf_counter() - t0
            assert st["verified_bytes"] == n * sz
            cpu = dict(value=round(n * sz / 2**20 / dt_cpu, 1), unit="MB/s", cores=threads, kind="port",
                       sample=f"all {n} rows once: oracle read loop (libzstd decode + scalar C BLAKE3), "
                              f"{threads} threads of {cores} visible cores")
        line = {
            "metric": "decompress MB/s (uncompressed) + compress MB/s, 100k x 10KB archive",
            "value": round(mbps_read, 1), "unit": "MB/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt_read / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": wl["name"], "rows_per_gpu": n, "chunk_bytes": sz, "archive": archive_src,
                       "parallelism": f"row-cursor ranges x{world}"},
            "compress_MBps": round(mbps_write, 1) if mbps_write else None,
            "compress_ms_per_step": round(dt_write / args.steps * 1e3, 4) if dt_write else None,
            "compress_kernel_ms": {k: round(v, 4) for k, v in k_write.items()},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
