"""Worker for tests/test_multirank.py: one rank of a gloo group running the sharded WRITE path on CPU with the
checker double (on GPUs the same code runs with the HIP backend, one rank per card)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch.distributed as dist  # noqa: E402

import gen  # noqa: E402
from oracle_backend import OracleBackend  # noqa: E402
from znippy_amd.stream_packer import ArchiveEntry, compress_stream  # noqa: E402


def entries():
    ents = [ArchiveEntry(f"d{i % 3}/f{i:03}.txt", gen.pseudo_text(3000 + 977 * i, seed=i)) for i in range(24)]
    ents += [ArchiveEntry("big.bin", gen.binary(20 * 1024 * 1024)), ArchiveEntry("stored.jar", gen.incompressible(1, 300000)),
             ArchiveEntry("empty.txt", b""), ArchiveEntry("pom.xml", gen.text(10240), 1, "maven")]
    return ents


def main():
    output, result = sys.argv[1:3]
    dist.init_process_group("gloo")
    c = compress_stream(output, False, backend=OracleBackend(n_threads=1))
    for e in entries():
        c.sender().send(e)
    rep = c.finish()
    json.dump(rep.__dict__, open(f"{result}.{dist.get_rank()}", "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
