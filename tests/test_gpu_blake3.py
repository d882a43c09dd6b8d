"""GPU parity: BLAKE3 through the C ABI vs the oracle (bit-exact, 32-byte digests).

Reference call sites replaced: stream_packer.rs:L219, slot_packer.rs:L553, decompress.rs:L172.
Edge cases follow the reference's tests: empty chunk (integration_test.rs:L111-131), single
small file, multi-chunk big file, many small files.
"""
import json
import os

import numpy as np
import pytest

import gen

pytestmark = pytest.mark.gpu

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "blake3_kat.json")))


def test_shim_known_answers(gpu_ctx):
    assert gpu_ctx.blake3(b"").hex() == KAT["empty"]
    assert gpu_ctx.blake3(b"abc").hex() == KAT["abc"]
    for n, h in KAT["pattern251"].items():
        assert gpu_ctx.blake3(gen.binary(int(n))).hex() == h, n


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 1023, 1024, 1025, 2047, 2048, 2049, 3 * 1024, 10240, 65535, 65536,
                               65537, 100000, 1 << 20, (1 << 20) + 1, 8 << 20, (8 << 20) + 777, 200 * 1000 * 1000 // 29])
def test_shim_vs_oracle_sizes(gpu_ctx, oracle, n):
    data = gen.incompressible(n, n)
    assert gpu_ctx.blake3(data) == oracle.blake3(data)


def test_rounds_mixed_sizes_unaligned(gpu_ctx, oracle):
    """Many rounds of ragged sizes at unaligned offsets in one staging buffer: exercises tile
    packing (several chunks per wave), big-unit slices and the merge kernel."""
    import torch
    from znippy_amd import hip
    rng = np.random.default_rng(5)
    sizes = [0, 1, 5, 64, 1000, 1024, 1025, 4096, 10240, 10240, 10240, 33333, 65536, 65537, 70000, 300000,
             0, 2, 1 << 20, 3, 999, 5 << 20, 17, 64 * 1024 * 3 + 5] + [int(x) for x in rng.integers(0, 20000, 400)]
    offs, pos = [], 3
    for s in sizes:
        offs.append(pos)
        pos += s + int(rng.integers(0, 7))
    buf = np.frombuffer(gen.incompressible(99, pos + 16), dtype=np.uint8)
    d = torch.from_numpy(buf.copy()).cuda()
    rt = hip.RoundTable(gpu_ctx, offs, sizes)
    got = rt.hash(d)
    for i, (o, s) in enumerate(zip(offs, sizes)):
        assert got[i].tobytes() == oracle.blake3(buf[o:o + s]), (i, o, s)


def test_rounds_around_the_one_launch_merge_limit(gpu_ctx, oracle):
    """Units of more than 64 tile CVs (rounds above 4 MiB): up to 4 groups of 64 (16 MiB) the groups and the unit's top
    are folded by one launch (k_merge_units: a workgroup per unit, a wave per group); a table holding a bigger round takes
    the two launches (k_merge_groups, k_merge_big).  Sizes with a ragged last group, a last group of one CV, exactly 4
    groups, and tables with a round beyond the limit — every digest equal to the oracle's."""
    import torch
    import gen_gpu
    from znippy_amd import hip
    d = gen_gpu.incompressible(4242, (300 << 20) + 16)
    h = d.cpu().numpy()

    def check(sizes, first=0):
        offs = (first + np.concatenate([[0], np.cumsum(sizes)[:-1]])).astype(np.uint64)
        rt = hip.RoundTable(gpu_ctx, offs, sizes)
        got = rt.hash(d)
        for i, (o, n) in enumerate(zip(offs, sizes)):
            assert got[i].tobytes() == oracle.blake3(h[int(o):int(o) + n]), (i, n)
        rt.close()

    check([(4 << 20) + 1, (4 << 20) + 65536 + 7, 8 << 20, (12 << 20) - 5, 16 << 20, 3, (12 << 20) + 65536 + 1, 70000])   # one launch
    check([(16 << 20) + 1, 8 << 20, 100], first=3)                      # five groups: the two launches
    check([(256 << 20) + (4 << 20) + 1000, 10 << 20], first=5)          # 65 groups and a bit


def test_100k_small_chunks_checksum_of_checksums(gpu_ctx, oracle):
    """BASELINE C2 shape (100k x 10 KiB identical text chunks): all digests equal the oracle's."""
    import torch
    from znippy_amd import hip
    n, sz = 100_000, 10240
    chunk = np.frombuffer(gen.text(sz), dtype=np.uint8)
    d = torch.from_numpy(np.tile(chunk, n)).cuda()
    rt = hip.RoundTable(gpu_ctx, np.arange(n, dtype=np.uint64) * sz, np.full(n, sz, dtype=np.uint64))
    got = rt.hash(d)
    want = np.frombuffer(oracle.blake3(chunk), dtype=np.uint8)
    assert (got == want[None, :]).all()
