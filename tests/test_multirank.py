"""N>1 path on CPU: world_size-2 gloo group, per-rank row ranges, counters reduced with one
all-reduce.  Same host code as the GPU path (backend nccl = RCCL there)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import gen
from znippy_amd.decompress import decompress_archive
from znippy_amd.sharding import split_rows
from znippy_amd.stream_packer import ArchiveEntry, compress_stream

HERE = os.path.dirname(os.path.abspath(__file__))


def test_split_rows_is_a_partition_balanced_by_bytes():
    rng = np.random.default_rng(0)
    for world in (1, 2, 3, 8):
        for w in ([], [5], [0, 0, 0], list(rng.integers(0, 10**6, 1000)), [10**9] + [1] * 100, [1] * 7):
            parts = split_rows(w, world)
            assert len(parts) == world
            assert parts[0][0] == 0 and parts[-1][1] == len(w)
            for (a, b), (c, d) in zip(parts, parts[1:]):
                assert b == c and a <= b and c <= d
    # skewed archive (C5 shape): ranges are balanced by bytes, not by row count
    w = [2000] * 3500 + [10**6] * 1400 + [4 * 10**7] * 100
    parts = split_rows(w, 8)
    loads = [sum(w[a:b]) for a, b in parts]
    assert max(loads) <= 1.35 * (sum(w) / 8) + 4 * 10**7
    rows = [b - a for a, b in parts]
    assert max(rows) > 10 * max(min(rows), 1)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_world2_gloo_matches_single_process(tmp_path, oracle):
    from oracle_backend import OracleBackend
    entries = [ArchiveEntry(f"d{i % 3}/f{i:03}.txt", gen.pseudo_text(3000 + 977 * i, seed=i)) for i in range(24)]
    entries += [ArchiveEntry("big.bin", gen.binary(20 * 1024 * 1024)),            # 3 chunks, may straddle ranks
                ArchiveEntry("stored.jar", gen.incompressible(1, 300000)), ArchiveEntry("empty.txt", b"")]
    c = compress_stream(tmp_path / "a.znippy", False, backend=OracleBackend())
    for e in entries:
        c.sender().send(e)
    c.finish()
    archive = tmp_path / "a.znippy"
    # corrupt one stored payload byte so the reduced report has something to carry
    raw = bytearray(archive.read_bytes())
    idx = raw.find(gen.incompressible(1, 300000)[:64])
    raw[idx + 5] ^= 1
    archive.write_bytes(bytes(raw))

    single = decompress_archive(archive, True, tmp_path / "single", backend=OracleBackend())
    # the multi-rank run extracts OVER files left by an earlier, longer extraction: no rank may O_TRUNC (the others
    # write their parts in any order), so the first rank to touch a file sets its final length — stale tails must go
    for e in entries:
        stale = tmp_path / "multi" / e.relative_path
        stale.parent.mkdir(parents=True, exist_ok=True)
        stale.write_bytes(b"\xEE" * (len(e.data) + 4321))
    port = _free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_mr_worker.py"), str(archive),
                                       str(tmp_path / "multi"), str(tmp_path / "rep.json")], env=e))
    for p in procs:
        assert p.wait(timeout=300) == 0
    multi = json.load(open(tmp_path / "rep.json"))
    assert multi == single.__dict__
    assert single.corrupt_files == 1 and single.total_files == len(entries)
    for e in entries:
        a = (tmp_path / "single" / e.relative_path).read_bytes()
        b = (tmp_path / "multi" / e.relative_path).read_bytes()
        assert a == b
        if e.relative_path != "stored.jar":
            assert a == e.data


def test_world2_gloo_write_side_equals_single_process(tmp_path, oracle):
    """SURVEY 8e, write side: each rank encodes a contiguous range of the Rounds, rank 0 concatenates the payload
    regions (blob offsets rebased by a running sum) and writes the archive — byte-identical to the one-process
    archive, same report on every rank."""
    import _mr_write_worker as w
    from oracle_backend import OracleBackend
    c = compress_stream(tmp_path / "single.znippy", False, backend=OracleBackend())
    for e in w.entries():
        c.sender().send(e)
    single = c.finish()
    port = _free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_mr_write_worker.py"), str(tmp_path / "multi.znippy"),
                                       str(tmp_path / "rep.json")], env=e))
    for p in procs:
        assert p.wait(timeout=300) == 0
    assert (tmp_path / "multi.znippy").read_bytes() == (tmp_path / "single.znippy").read_bytes()
    for r in range(2):
        assert json.load(open(f"{tmp_path / 'rep.json'}.{r}")) == single.__dict__
    v = decompress_archive(tmp_path / "multi.znippy", False, None, backend=OracleBackend())
    assert (v.corrupt_files, v.total_files) == (0, len(w.entries()))


@pytest.mark.gpu
@pytest.mark.parametrize("scaling,workload", [("weak", "c2small"), ("strong", "c2small"), ("strong", "c5small")])
def test_bench_n2_path_rehearsed_on_one_gpu(scaling, workload):
    """bench.py's N>1 code path (rendezvous, per-step counter all-reduce on its own stream, barrier + MAX-over-ranks
    timing, rank-0 JSON) with two ranks sharing the one card over gloo; the driver runs the real thing over RCCL.
    weak: every rank owns a copy of the workload; strong: ONE archive, the row cursor split into per-rank ranges and
    every rank builds only its own share of the staging buffer.  c5small = BASELINE configs[4]'s mixed archive (xml text
    + stored jars, skewed sizes) at 1/16 of the files: the shape the driver's 8-rank run of `--workload c5 --scaling
    strong` exercises."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ZNIPPY_BENCH_BACKEND="gloo")
    port = {"weak": 29533, "strong": 29534}[scaling] + (2 if workload == "c5small" else 0)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--workload", workload, "--scaling", scaling],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == scaling and line["value"] > 0
    assert line["config"]["parallelism"].endswith("x2") and line["cpu_baseline"] is None
    if workload == "c2small":
        assert abs(line["config"]["rows_per_gpu"] - (2000 if scaling == "weak" else 1000)) <= 1
        assert "libzstd level-19" in line["config"]["archive"] and line["read_own_archive"]["MBps"] > 0
    else:
        import workloads
        lay = workloads.layout("c5small")
        (a0, a1), _ = split_rows(lay["lens"], 2)
        assert line["config"]["rows_per_gpu"] == a1 - a0                     # rank 0's range, balanced by bytes not rows
        assert line["config"]["bytes_per_gpu"] == int(lay["lens"][a0:a1].sum()) < 0.6 * int(lay["lens"].sum())
