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


def measured_traffic(kernel, workload):
    """HBM bytes per launch of `kernel` from the PMC passes recorded under profiles/ (rocprofv3 --pmc
    FETCH_SIZE and --pmc WRITE_SIZE in separate runs, FETCH_SIZE doubled per the gfx950 correction)."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        return d.get(workload, {}).get(kernel)
    except (OSError, ValueError):
        return None


def build_workload(name, torch):
    """-> dict(d_src, lens, skip, name, sample): this rank's Rounds over a resident staging buffer."""
    import gen
    import gen_gpu
    if name in ("c2", "c2small"):
        n = 100_000 if name == "c2" else 2_000
        chunk = np.frombuffer(gen.text(10 * 1024), dtype=np.uint8)
        d_src = torch.from_numpy(np.tile(chunk, n)).cuda()
        label = ("100k x 10KiB text chunks (BASELINE configs[1])" if name == "c2"
                 else "2k x 10KiB text chunks (reduced; NOT the headline config)")
        return dict(d_src=d_src, lens=np.full(n, 10240, np.uint64), skip=None, name=label)
    if name == "c3":
        size, sl = 2 << 30, 8 << 20
        return dict(d_src=gen_gpu.text(size), lens=np.full(size // sl, sl, np.uint64), skip=None,
                    name="single 2 GiB text file, 256 x 8 MiB slices (BASELINE configs[2] at the reference's slice size)")
    if name == "c3slot":
        size, sl = 2 << 30, 200 << 20
        lens = np.array([sl] * (size // sl) + ([size % sl] if size % sl else []), dtype=np.uint64)
        return dict(d_src=gen_gpu.text(size), lens=lens, skip=None,
                    name="single 2 GiB text file, 11 slices of <= 200 MiB (BASELINE configs[2] as worded there)")
    if name == "c5":
        # synthetic stand-in for the 5 GB / 5k-file artifact repo: 3,500 .xml text files of 1-8 KiB,
        # 1,400 .jar of 100 KiB..2 MiB and 100 .jar of 20..69.5 MiB incompressible bytes (store path), stream
        # chunking (8 MiB slices).  Jar bytes are consecutive cuts of one LCG stream.
        sl = 8 << 20
        xml = [1024 + (i % 8) * 1024 for i in range(3500)]
        jars = [100 * 1024 + (i % 20) * 100 * 1024 for i in range(1400)] + [(20 << 20) + i * (1 << 19) for i in range(100)]
        lens, skip = list(xml), [0] * len(xml)
        for j in jars:
            for o in range(0, j, sl):
                lens.append(min(sl, j - o))
                skip.append(1)
        d_src = torch.cat([gen_gpu.text(sum(xml)), gen_gpu.incompressible(7, sum(jars))])
        return dict(d_src=d_src, lens=np.array(lens, np.uint64), skip=np.array(skip, np.uint8),
                    name="mixed artifact repo stand-in: 3,500 xml (1-8 KiB) + 1,500 jars (100 KiB-69.5 MiB, store path), %.2f GB"
                         % ((sum(xml) + sum(jars)) / 1e9))
    if name in ("c4store", "c4codec"):
        size, sl = 500 << 20, 8 << 20
        lens = np.array([sl] * (size // sl) + ([size % sl] if size % sl else []), dtype=np.uint64)
        skip = np.ones(len(lens), np.uint8) if name == "c4store" else None
        return dict(d_src=gen_gpu.random_lcg(size), lens=lens, skip=skip,
                    name="500 MiB LCG blob, 8 MiB slices, " + ("store path (random.jar)" if skip is not None else "codec path (random.bin)"))
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
    # one process per GPU; ZNIPPY_BENCH_BACKEND=gloo lets several ranks share one card to rehearse the N>1 path
    backend = os.environ.get("ZNIPPY_BENCH_BACKEND", "nccl")
    dev = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend)

    import gen  # noqa: F401
    from oracle import oracle as O  # checker + cpu_baseline leg only
    from znippy_amd import hip

    wl = build_workload(args.workload, torch)
    d_src, lens, skip = wl["d_src"], wl["lens"], wl["skip"]
    n = len(lens)
    total_in = int(lens.sum())
    src_off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    ctx = hip.Context(dev)

    # ---- write side: this rank's Rounds over the resident staging buffer ----
    rounds = hip.RoundTable(ctx, src_off, lens, skip)
    d_blob = torch.zeros(rounds.blob_bound() + 64, dtype=torch.uint8, device="cuda")
    enc = rounds.encode_hash(d_src, d_blob)
    enc = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in enc.items()}
    # parity spot checks against the oracle: digests and frames of a few rounds
    host_blob_head = None
    for i in sorted({0, n // 2, n - 1}):
        src_i = d_src[int(src_off[i]):int(src_off[i] + lens[i])].cpu().numpy()
        assert enc["checksum"][i].tobytes() == O.blake3(src_i), "GPU write-side digest != oracle"
        f = d_blob[int(enc["blob_offset"][i]):int(enc["blob_offset"][i] + enc["blob_size"][i])].cpu().numpy()
        if enc["compressed"][i]:
            assert O.zstd_decompress(f.tobytes()) == src_i.tobytes(), "GPU frame does not decode on the oracle"
        else:
            assert f.tobytes() == src_i.tobytes()
    bo, bs, comp, ck = enc["blob_offset"], enc["blob_size"], enc["compressed"], enc["checksum"]
    d_blobs = d_blob
    archive_src = "gpu-encoded (this build's zstd frames)"
    sz = int(lens[0])

    out_off = src_off
    d_out = torch.zeros(total_in + 64, dtype=torch.uint8, device="cuda")
    rows = hip.RowTable(ctx, bo, bs, lens, out_off, np.packbits(comp.astype(bool), bitorder="little"), ck)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    cvec = torch.zeros(8, dtype=torch.int64, device="cuda")
    hvec = torch.zeros(8, dtype=torch.int64).pin_memory() if world > 1 else None
    comm_stream = torch.cuda.Stream() if world > 1 else None  # the reduce never sits in front of the next pass

    def read_step():
        rows.decode_verify_async(d_blobs, d_out)
        counters, corrupt, _ = rows.results(want_status=False)
        if world > 1:  # the only cross-GPU traffic: one small all-reduce of the counters (RCCL over xGMI)
            comm_stream.synchronize()  # the previous step's reduce is done with hvec / cvec
            for i, k in enumerate(("total_chunks", "total_written_bytes", "verified_bytes", "corrupt_bytes",
                                   "corrupt_rows", "decode_errors")):
                hvec[i] = counters[k]
            with torch.cuda.stream(comm_stream):
                cvec.copy_(hvec, non_blocking=True)
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
        for _ in range(steps):
            fn()
        barrier()
        dt = time.perf_counter() - t0
        # per-kernel durations (HIP events the library records around each launch on its own streams): read back in
        # a few extra, untimed steps so that the event queries are not part of the timed region
        ktimes = {}
        for _ in range(min(steps, 10)):
            fn()
            for name, ms in ctx.kernel_times():
                ktimes.setdefault(name, []).append(ms)
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()), {k: float(np.mean(v)) for k, v in ktimes.items()}

    # ---- read side (the headline) ----
    counters = read_step()
    assert counters["corrupt_rows"] == 0 and counters["decode_errors"] == 0 and \
        counters["verified_bytes"] == total_in, counters
    assert torch.equal(d_out[:total_in], d_src[:total_in]), "decoded bytes differ from the source"
    dt_read, k_read = timed(read_step, args.steps, args.warmup)
    dt_write, k_write = timed(write_step, args.steps, args.warmup)

    total_bytes = total_in * world
    mbps_read = total_bytes / 2**20 / (dt_read / args.steps)
    mbps_write = total_bytes / 2**20 / (dt_write / args.steps) if dt_write else None

    if rank == 0:
        # roofline of the dominant read-side kernel, algorithmic bytes per launch
        blob_bytes = int(bs.sum())
        path_alg = blob_bytes + total_in + 57 * n        # read each blob byte, write each output byte, index columns
        alg = {
            "decode_verify_fused": path_alg,             # one launch does the whole path for small rows
            "zstd_decode_general": blob_bytes + total_in + 57 * n,
            "zstd_decode_blocks": blob_bytes + total_in + 57 * n,   # block items: same bytes, one work item per block
            "decode_verify_fused_blocks": blob_bytes + total_in + 57 * n,  # block items written + hashed in one go
            "blake3_second_pass": total_in + 32 * n + (0 if skip is None else total_in),  # read (+ copy on the store path)
        }
        # (a launch that only waits for CUs while the other stream's kernel runs — the general decoder with nothing
        # routed to it — is not a candidate: its interval measures its neighbour)
        idle = {"zstd_decode_general"} if "decode_verify_fused_blocks" in k_read else set()
        dom = max((k for k in k_read if k in alg and k not in idle), key=lambda k: k_read[k]) if k_read else None
        roofline = None
        if dom:
            ach = alg[dom] / (k_read[dom] * 1e-3) / 1e9
            path_ms = sum(k_read.values())
            roofline = dict(bound="hbm", kernel=dom, achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                            frac=round(ach / HBM_PEAK_GBS, 4), traffic=None,
                            kernel_ms={k: round(v, 4) for k, v in k_read.items()},
                            path_achieved=round(path_alg / (path_ms * 1e-3) / 1e9, 1),
                            path_frac=round(path_alg / (path_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
            tr = measured_traffic(dom, args.workload)
            if tr is not None:
                roofline["traffic"] = tr
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cores = len(os.sched_getaffinity(0))
            threads = max(1, int(np.ceil(0.9 * cores)))  # common_config.rs:L34 rule on what we can see
            host_blobs = d_blobs.cpu().numpy()
            bitmap = np.packbits(comp.astype(bool), bitorder="little")
            host_out = np.zeros(total_in, dtype=np.uint8)
            t0 = time.perf_counter()
            st, _ = O.decompress_rows(host_blobs, bo, bs, lens, out_off, bitmap, ck, 0, n, out=host_out,
                                      n_threads=threads, use_libzstd=O.have_libzstd())
            dt_cpu = time.perf_counter() - t0
            assert st["verified_bytes"] == total_in
            cpu = dict(value=round(total_in / 2**20 / dt_cpu, 1), unit="MB/s", cores=threads, kind="port",
                       sample=f"all {n} rows once: oracle read loop (libzstd decode + scalar C BLAKE3), "
                              f"{threads} threads of {cores} visible cores")
        line = {
            "metric": "decompress MB/s (uncompressed) + compress MB/s, 100k x 10KB archive",
            "value": round(mbps_read, 1), "unit": "MB/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt_read / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": wl["name"], "rows_per_gpu": n, "chunk_bytes": sz, "bytes_per_gpu": total_in,
                       "blob_bytes_per_gpu": int(bs.sum()), "archive": archive_src,
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
