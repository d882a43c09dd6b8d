"""GPU parity: MANY foreign frames at once through the batch path (zstd_batch.hip, k_bx_*) — table descriptions, Huffman
streams and the FSE sequence bitstream decoded with lane = block, execution wave = frame — against the source bytes and
the oracle's digests, with NO serial decoder behind it (ZNIPPY_FZ_ONLY): a frame the batch kernels did not finish shows
up as a corrupt row.  This is codec::decompress_into (codec.rs:L67-78) as the read worker loop calls it for every row of
an archive of small chunks (decompress.rs:L135-190; BASELINE configs[1] with real text in place of the 45-byte phrase)."""
import os

import numpy as np
import pytest

import gen
import workloads
from test_gpu_foreign import _py_corpus, _mixed

pytestmark = pytest.mark.gpu


def _table(oracle, entries, frames):
    bs = np.array([len(f) for f in frames], np.uint64)
    bo = np.concatenate([[0], np.cumsum(bs)[:-1]]).astype(np.uint64)
    us = np.array([len(e) for e in entries], np.uint64)
    oo = np.concatenate([[0], np.cumsum(us)[:-1]]).astype(np.uint64)
    dig = {}
    for e in entries:
        if e not in dig:
            dig[e] = np.frombuffer(oracle.blake3(e), dtype=np.uint8)
    ck = np.stack([dig[e] for e in entries])
    blobs = np.frombuffer(b"".join(frames) + bytes(64), dtype=np.uint8)
    return dict(blobs=blobs, bo=bo, bs=bs, us=us, oo=oo, ck=ck)


def _decode(ctx, A, reps=2):
    import torch
    from znippy_amd import hip
    d_blobs = torch.from_numpy(A["blobs"].copy()).cuda()
    total = int(A["us"].sum())
    d_out = torch.zeros(total + 64, dtype=torch.uint8, device="cuda")
    rt = hip.RowTable(ctx, A["bo"], A["bs"], A["us"], A["oo"], None, A["ck"])
    outs = []
    for _ in range(reps):  # work lists are filled in a different order every run: every run must agree
        d_out.zero_()
        c, corrupt, status = rt.decode_verify(d_blobs, d_out)
        outs.append((dict(c), status.copy(), d_out[:total].cpu().numpy().copy()))
    for o in outs[1:]:
        assert o[0] == outs[0][0] and (o[1] == outs[0][1]).all() and (o[2] == outs[0][2]).all()
    return outs[0] + (rt.foreign_stats(), dict(ctx.kernel_times()))


@pytest.mark.parametrize("level", [1, 3, 19])
def test_many_small_text_frames_batch_only(gpu_ctx_fz_only, oracle, level):
    """3,000 slices of real text, 1 B .. 40 KiB (ragged: empty and one-byte rows included), libzstd frames: Huffman
    literals with direct and FSE-compressed weights, described / predefined / RLE sequence tables, single streams."""
    data = _py_corpus(12 << 20)
    rng = np.random.default_rng(level)
    entries, pos = [], 0
    for i in range(3000):
        n = int(rng.choice([0, 1, 7, 63, 200, 1000, 4096, 10240, 10240, 10240, 20000, 40960]))
        if pos + n > len(data):
            pos = 0
        entries.append(data[pos:pos + n])
        pos += n
    frames = [workloads.libzstd_compress(e, level) for e in entries]
    A = _table(oracle, entries, frames)
    c, status, out, st, kt = _decode(gpu_ctx_fz_only, A)
    assert c["corrupt_rows"] == 0 and c["decode_errors"] == 0, (c, np.nonzero(status)[0][:10])
    assert out.tobytes() == b"".join(entries)
    assert st["blocks_given_up"] == 0 and "zstd_batch_sequences" in kt


def test_kinds_of_small_frames_batch_only(gpu_ctx_fz_only, oracle):
    """Random bytes (raw blocks / raw literals), constant runs (RLE blocks, RLE tables), the periodic BASELINE chunks, a
    Zipf word stream and shared-object bytes, 200 B .. 128 KiB, at four levels — one table."""
    rng = np.random.default_rng(7)
    so = b"".join(workloads.image_corpus("binary", 4 << 20, whole_files=False))[:4 << 20]
    entries = []
    for i in range(600):
        n = int(rng.integers(200, 131072)) if i % 5 else int(rng.integers(200, 3000))
        k = i % 6
        if k == 0: e = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
        elif k == 1: e = bytes([i & 255]) * n
        elif k == 2: e = gen.text(n) if i % 12 == 2 else gen.binary(n)
        elif k == 3: e = gen.pseudo_text(n, seed=i)
        elif k == 4: o = int(rng.integers(0, len(so) - n)); e = so[o:o + n]
        else: e = _mixed(n, i)
        entries.append(e)
    frames = [workloads.libzstd_compress(e, (1, 3, 9, 19)[i % 4]) for i, e in enumerate(entries)]
    A = _table(oracle, entries, frames)
    c, status, out, st, kt = _decode(gpu_ctx_fz_only, A)
    assert c["corrupt_rows"] == 0 and c["decode_errors"] == 0, (c, np.nonzero(status)[0][:10])
    assert out.tobytes() == b"".join(entries)


def test_lane_and_wave_sequence_decoders_agree(oracle):
    """The sequence stage has two decoders (lane = block, and wave = block for long chains): the same table through a
    context that sends every block to the first (ZNIPPY_BX_BIG huge) and one that sends every block to the second
    (ZNIPPY_BX_BIG=1) must give the same bytes, statuses and counters."""
    from znippy_amd import hip
    data = _py_corpus(4 << 20)
    entries = [data[i * 30000:(i + 1) * 30000 + (i % 7) * 1000] for i in range(100)] + [data[:700_000], _mixed(500_000, 3)]
    frames = [workloads.libzstd_compress(e, 19 if i % 2 else 3) for i, e in enumerate(entries)]
    A = _table(oracle, entries, frames)
    res = []
    for big in ("1000000000", "1"):
        os.environ["ZNIPPY_BX_BIG"] = big
        os.environ["ZNIPPY_FZ_ONLY"] = "1"
        try:
            ctx = hip.Context(0)
        finally:
            del os.environ["ZNIPPY_BX_BIG"], os.environ["ZNIPPY_FZ_ONLY"]
        try:
            res.append(_decode(ctx, A, reps=1))
        finally:
            ctx.close()
    (c1, s1, o1, _, _), (c2, s2, o2, _, _) = res
    assert c1 == c2 and (s1 == s2).all() and (o1 == o2).all()
    assert c1["corrupt_rows"] == 0 and o1.tobytes() == b"".join(entries)


def test_damaged_small_frames_agree_with_oracle(gpu_ctx, oracle):
    """Small text frames damaged anywhere (bit flips, random bytes, truncation, bursts, swaps), hundreds of rows in one
    table so that they share the batch kernels' waves with intact frames: the oracle's verdict is the GPU's, accepted
    mutants decode to the oracle's bytes, nothing is reported verified with different bytes, intact rows are untouched."""
    from test_gpu_fuzz import _run as fuzz_run
    data = _py_corpus(1 << 20)
    bases = [(data[i * 9000:i * 9000 + 10240], (19, 3, 1)[i % 3]) for i in range(12)]
    fuzz_run(gpu_ctx, oracle, bases, 40, 2024, 10, 100)


def test_more_blocks_than_item_slots(gpu_ctx, oracle):
    """A writer may cut its frames into far more blocks than ceil(size / 128 KiB) — libzstd with a 1 KiB window emits
    1 KiB blocks — and the batch path's item slots (sized from the index columns) then run out: the frames that no longer
    fit stay with the serial decoder, and the table kernel must not touch the slots the scan left unfilled.  A first
    table on the same context leaves stale items in the memory the second one's arrays are carved from (found by
    tools/soak_foreign.py: a memory access fault on the third table of a process)."""
    data = _py_corpus(8 << 20)
    first = [data[i * 2000:i * 2000 + 1500 + (i % 5) * 300] for i in range(3000)]
    A0 = _table(oracle, first, [workloads.libzstd_compress(e, 3) for e in first])
    c, status, out, st, kt = _decode(gpu_ctx, A0, reps=1)
    assert c["corrupt_rows"] == 0 and out.tobytes() == b"".join(first)
    entries = [data[i * 200_000:i * 200_000 + 262_144] for i in range(30)] + [data[i * 3000:i * 3000 + 2500] for i in range(200)]
    frames = [workloads.libzstd_compress_adv(e, level=(3, 19)[i % 2], window_log=10) for i, e in enumerate(entries[:30])]
    frames += [workloads.libzstd_compress(e, 3) for e in entries[30:]]
    nblk = 0
    for f in frames[:30]:   # count the blocks of the big frames: 256 of 1 KiB each
        assert not (f[4] >> 5) & 1   # a window descriptor, not a single segment
        pos, k = 6 + (1, 2, 4, 8)[f[4] >> 6], 0
        while True:
            bh = f[pos] | f[pos + 1] << 8 | f[pos + 2] << 16
            pos += 3 + (1 if (bh >> 1) & 3 == 1 else bh >> 3); k += 1
            if bh & 1: break
        nblk += k
    assert nblk > 30 * 200   # ~7,700 blocks against ~1,900 item slots
    A = _table(oracle, entries, frames)
    c, status, out, st, kt = _decode(gpu_ctx, A)
    assert c["corrupt_rows"] == 0 and c["decode_errors"] == 0, (c, np.nonzero(status)[0][:10])
    assert out.tobytes() == b"".join(entries)


def test_tables_the_fused_kernels_hand_over_entirely_skip_them(gpu_ctx, oracle):
    """When the fused kernels handed every row of a table over to the batch path (rows of real text: nothing a recogniser
    takes), later runs of the table give the rows to the batch path themselves and launch neither fused kernel.  Every run
    returns every row; rows replaced later by frames of other kinds (a periodic one the fused kernel would have taken, a raw
    one, a tiny one) still decode, through the batch path."""
    import torch
    from znippy_amd import hip
    data = _py_corpus(8 << 20)
    n = 2500
    entries = [data[i * 3000:i * 3000 + 2048 + (i % 9) * 700] for i in range(n)]
    frames = [workloads.libzstd_compress(e, 19 if i % 2 else 3) for i, e in enumerate(entries)]
    # replacements of the same content size, used after the third run
    swaps = {5: gen.text(len(entries[5])), 900: gen.incompressible(3, len(entries[900])), 2499: bytes(len(entries[2499]))}
    slot = [max(len(frames[i]), len(workloads.libzstd_compress(swaps[i], 3)) if i in swaps else 0) + 3 for i in range(n)]
    bo = (np.cumsum(slot) - np.array(slot)).astype(np.uint64)
    bs = np.array([len(f) for f in frames], np.uint64)
    us = np.array([len(e) for e in entries], np.uint64)
    oo = (np.cumsum(us) - us).astype(np.uint64)
    blob = np.zeros(int(sum(slot)) + 64, np.uint8)
    for i, f in enumerate(frames):
        blob[int(bo[i]):int(bo[i]) + len(f)] = np.frombuffer(f, np.uint8)
    ck = np.stack([np.frombuffer(oracle.blake3(e), dtype=np.uint8) for e in entries])
    d_blobs = torch.from_numpy(blob.copy()).cuda()
    total = int(us.sum())
    d_out = torch.zeros(total + 64, dtype=torch.uint8, device="cuda")
    rt = hip.RowTable(gpu_ctx, bo, bs, us, oo, None, ck)
    names = []
    for step in range(3):
        d_out.zero_()
        c, corrupt, status = rt.decode_verify(d_blobs, d_out)
        names.append(sorted(dict(gpu_ctx.kernel_times())))
        assert c["corrupt_rows"] == 0 and c["decode_errors"] == 0 and c["verified_bytes"] == total, (step, c)
        assert d_out[:total].cpu().numpy().tobytes() == b"".join(entries)
    assert "decode_verify_fused" in names[0] and "decode_verify_fused" not in names[2], names
    rt.close()
    # the same slots with other frames in three rows: a new table over the changed blobs, run until it skips the fused kernels too
    entries2, bs2 = list(entries), bs.copy()
    for i, e in swaps.items():
        f = workloads.libzstd_compress(e, 3)
        d_blobs[int(bo[i]):int(bo[i]) + len(f)] = torch.from_numpy(np.frombuffer(f, np.uint8).copy()).cuda()
        entries2[i] = e; bs2[i] = len(f)
    ck2 = np.stack([np.frombuffer(oracle.blake3(e), dtype=np.uint8) for e in entries2])
    rt = hip.RowTable(gpu_ctx, bo, bs2, us, oo, None, ck2)
    for step in range(3):
        d_out.zero_()
        c, corrupt, status = rt.decode_verify(d_blobs, d_out)
        assert c["corrupt_rows"] == 0 and c["decode_errors"] == 0, (step, c)
        assert d_out[:total].cpu().numpy().tobytes() == b"".join(entries2)
    rt.close()
