"""Worker for tests/test_multirank.py: one rank of a gloo group running the sharded read path
on CPU with the checker double (the GPU path uses the same code with the nccl backend)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch.distributed as dist  # noqa: E402

from oracle_backend import OracleBackend  # noqa: E402
from znippy_amd.decompress import decompress_archive  # noqa: E402


def main():
    archive, out_dir, result = sys.argv[1:4]
    dist.init_process_group("gloo")
    rep = decompress_archive(archive, True, out_dir, backend=OracleBackend(n_threads=1))
    if dist.get_rank() == 0:
        json.dump(rep.__dict__, open(result, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
