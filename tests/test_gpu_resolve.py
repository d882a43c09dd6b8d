"""GPU parity: the resolve path (zstd_batch.hip, k_rx_*) — big foreign frames are not executed sequence by sequence; every
output byte gets a word (its value, or the word it copies from), rounds of pointer jumping turn all words into values, a
last pass stores them.  Checked against the source bytes, against the same table decoded with the path switched off
(ZNIPPY_NO_RX: one wave per frame), and — for damaged frames — against the oracle's verdicts.  The reference side is
codec::decompress_into on a whole chunk (codec.rs:L67-78): whatever route a frame takes, the row's bytes are the same."""
import os

import numpy as np
import pytest

import gen
import workloads
from test_gpu_batch import _table
from test_gpu_foreign import _py_corpus, _mixed

pytestmark = pytest.mark.gpu


def _decode_with(env, A, out_pad=0, reps=2):
    """decode table A in a fresh context created under `env`; rows land `out_pad` bytes into the output buffer"""
    import torch
    from znippy_amd import hip
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        ctx = hip.Context(0)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    try:
        d_blobs = torch.from_numpy(A["blobs"].copy()).cuda()
        oo = A["oo"] + np.uint64(out_pad)
        total = int(A["us"].sum())
        d_out = torch.zeros(total + out_pad + 64, dtype=torch.uint8, device="cuda")
        rt = hip.RowTable(ctx, A["bo"], A["bs"], A["us"], oo, None, A["ck"])
        res = None
        for _ in range(reps):
            d_out.fill_(0xEE)
            torch.cuda.synchronize()
            c, corrupt, status = rt.decode_verify(d_blobs, d_out)
            got = (dict(c), status.copy(), d_out.cpu().numpy().copy(), dict(ctx.kernel_times()), rt.foreign_stats())
            if res is not None:
                assert got[0] == res[0] and (got[1] == res[1]).all() and (got[2] == res[2]).all()
            res = got
        rt.close()
        return res
    finally:
        ctx.close()


def _check(entries, frames, oracle, out_pad=0, want_resolve=True):
    A = _table(oracle, entries, frames)
    c, status, out, kt, st = _decode_with({}, A, out_pad)
    total = int(A["us"].sum())
    assert c["corrupt_rows"] == 0 and c["decode_errors"] == 0, (c, np.nonzero(status)[0][:10])
    assert out[out_pad:out_pad + total].tobytes() == b"".join(entries)
    assert (out[:out_pad] == 0xEE).all() and (out[out_pad + total:out_pad + total + 60] == 0xEE).all()   # nothing around the rows is written
    if want_resolve:
        assert "zstd_resolve_expand" in kt, sorted(kt)
    c2, status2, out2, kt2, st2 = _decode_with({"ZNIPPY_NO_RX": "1"}, A, out_pad, reps=1)
    assert "zstd_resolve_expand" not in kt2
    assert c2 == c and (status2 == status).all() and (out2 == out).all()


def test_big_text_and_binary_frames(oracle):
    """Frames of 64 KB .. 3 MB of real text, shared-object bytes and a mixture, libzstd levels 1 / 3 / 19: repeat offsets across
    blocks, treeless literals, matches reaching far back — every row equal to its source, same result with the path off."""
    data = _py_corpus(6 << 20)
    so = b"".join(workloads.image_corpus("binary", 4 << 20, whole_files=False))[:4 << 20]
    entries = [data[:300_000], data[100_000:100_000 + 1_234_567], so[:3_000_001], so[1 << 20:(1 << 20) + 262_144], _mixed(700_000, 5),
               data[2 << 20:(2 << 20) + 65_537]]
    frames = [workloads.libzstd_compress(e, (19, 3, 19, 1, 3, 19)[i]) for i, e in enumerate(entries)]
    _check(entries, frames, oracle)


def test_long_chains_and_odd_addresses(oracle):
    """What pointer jumping has to get right: chains of a million hops (a two-byte period copied forward through 2 MiB; runs),
    a frame cut into 1 KiB blocks (window log 10: 512 blocks), sizes around the thresholds — and rows that start at odd output
    addresses (the values leave as aligned dwords whatever the row's address)."""
    rng = np.random.default_rng(3)
    runs = np.repeat(rng.integers(0, 256, size=3000, dtype=np.uint8), rng.integers(1, 1500, size=3000))[:2_000_000].tobytes()
    entries = [b"ab" * (1 << 20), runs, (b"0123456789abcdef" * 8 + bytes(rng.integers(0, 256, size=37, dtype=np.uint8))) * 3000,
               gen.pseudo_text(524_288, 9), gen.pseudo_text(262_143, 10), gen.pseudo_text(262_145, 11), gen.pseudo_text(65_536, 12),
               gen.pseudo_text(1_048_577, 13)]
    frames = [workloads.libzstd_compress(e, 3) for e in entries]
    frames[3] = workloads.libzstd_compress_adv(entries[3], level=3, window_log=10)
    for pad in (0, 1, 2, 3):
        _check(entries, frames, oracle, out_pad=pad)


def test_damaged_big_frames_agree_with_oracle(gpu_ctx, oracle):
    """Big frames damaged anywhere — sequence sections, literals, block headers: the words are written from whatever the
    entropy stages decoded, so offsets are checked when they are written; the oracle's verdict is the GPU's, nothing is
    reported verified with other bytes, and the run ends."""
    from test_gpu_fuzz import _run as fuzz_run
    data = _py_corpus(2 << 20)
    bases = [(data[:400_000], 3), (data[300_000:300_000 + 280_000], 19), (_mixed(300_000, 8), 1)]
    fuzz_run(gpu_ctx, oracle, bases, 40, 99, 5, 40)


@pytest.mark.parametrize("tiles_per_wave", ["1", "2"])
def test_a_64_leaf_row_beside_a_big_rows_first_slice(oracle, tiles_per_wave):
    """The second hash pass with two tiles per wave (hash_kernels.hip; forced here, by itself from 7,680 tiles on): a row of
    exactly 64 KiB is one unit of 64 leaves — a root —, the first slice of a bigger row is 64 leaves too — a tile CV for
    the merge.  Queued by one wave they are the same node count and different kinds: they must not take the fold that
    assumes one kind (the slice's CV got the ROOT flag and landed in the digest column: a verified row reported corrupt).
    Foreign frames, so that both rows reach the second pass; every order of the two kinds, with ragged neighbours."""
    data = _py_corpus(3 << 20)
    entries = [data[:65_536], data[100_000:100_000 + 200_000], data[400_000:400_000 + 65_536], data[500_000:500_000 + 65_536],
               data[600_000:600_000 + 65_537], data[700_000:700_000 + 64 * 1024 * 3], data[1_000_000:1_000_000 + 65_536], data[1_100_000:1_100_000 + 1_048_577]]
    frames = [workloads.libzstd_compress(e, 3) for e in entries]
    A = _table(oracle, entries, frames)
    for pad in (0, 3):
        c, status, out, kt, st = _decode_with({"ZNIPPY_STORE_G": tiles_per_wave}, A, pad, reps=1)
        total = int(A["us"].sum())
        assert c["corrupt_rows"] == 0 and c["decode_errors"] == 0, (pad, c)
        assert out[pad:pad + total].tobytes() == b"".join(entries)
