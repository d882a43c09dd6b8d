"""Multi-GPU sharding of the row cursor.

Every index row is an independent unit (own frame, own checksum, own output offset), so the
reference's single atomic cursor (decompress.rs:L104,L136) becomes R contiguous row ranges, one
per rank, balanced by sum(uncompressed_size) rather than by row count.  No payload crosses GPUs;
the only collective is a sum of the report counters (+ a gather of the corrupt row ids).
"""
from typing import List, Sequence, Tuple

import numpy as np

COUNTER_KEYS = ("total_chunks", "total_written_bytes", "verified_bytes", "corrupt_bytes", "corrupt_rows",
                "decode_errors")


def split_rows(weights: Sequence[int], world: int) -> List[Tuple[int, int]]:
    """Contiguous [begin,end) per rank with near-equal sum(weights); every row in exactly one range."""
    w = np.asarray(weights, dtype=np.float64)
    n = len(w)
    if world <= 1 or n == 0:
        return [(0, n)] + [(n, n)] * (max(world, 1) - 1)
    # give empty rows a tiny weight so long runs of them still spread
    c = np.cumsum(np.maximum(w, 1.0))
    total = c[-1]
    cuts = [0]
    for r in range(1, world):
        cuts.append(int(np.searchsorted(c, total * r / world, side="left")))
    cuts.append(n)
    for i in range(1, len(cuts)):
        cuts[i] = max(cuts[i], cuts[i - 1])
    return [(cuts[i], cuts[i + 1]) for i in range(world)]


def reduce_counters(counters: dict, corrupt_rows, group=None):
    """Sum the counters over ranks and gather the corrupt row ids (works on gloo and nccl)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return dict(counters), sorted(int(x) for x in corrupt_rows)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor([int(counters[k]) for k in COUNTER_KEYS], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    out = {k: int(v) for k, v in zip(COUNTER_KEYS, t.tolist())}
    gathered = [None] * dist.get_world_size(group)
    dist.all_gather_object(gathered, [int(x) for x in corrupt_rows], group=group)
    return out, sorted(x for part in gathered for x in part)
