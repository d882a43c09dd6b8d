"""GPU parity: multi-block frames of ANOTHER writer (the container's libzstd: repeat offsets, Huffman trees and FSE
tables reused across blocks, matches reaching into earlier blocks) through the two-phase path — every block entropy-
decoded by its own workgroup, sequences executed frame by frame — against the source bytes, the oracle's digests and
the serial decoder's verdicts.  Replaces codec::decompress_into (codec.rs:L67-78) for the frames a reference archive
holds (level 19, common_config.rs:L37) when a chunk is larger than one 128 KiB block."""
import glob
import os

import numpy as np
import pytest

import gen
import workloads

pytestmark = pytest.mark.gpu


def _py_corpus(cap):
    """Real text: python sources of the image, in sorted order (deterministic on a given image)."""
    out, tot = [], 0
    for f in sorted(glob.glob("/usr/lib/python3.10/*.py")):
        try:
            b = open(f, "rb").read()
        except OSError:
            continue
        out.append(b)
        tot += len(b)
        if tot >= cap:
            break
    data = b"".join(out)
    if len(data) < cap:  # a bare image: fall back to the seeded word stream
        data += gen.pseudo_text(cap - len(data), seed=5)
    return data[:cap]


def _mixed(n, seed):
    """Text with incompressible and constant stretches: raw and RLE blocks between compressed ones."""
    rng = np.random.default_rng(seed)
    parts, tot = [], 0
    while tot < n:
        k = int(rng.integers(0, 4))
        m = int(rng.integers(20000, 400000))
        if k == 0:
            p = rng.integers(0, 256, size=m, dtype=np.uint8).tobytes()
        elif k == 1:
            p = bytes([int(rng.integers(0, 256))]) * m
        else:
            p = gen.pseudo_text(m, seed=int(rng.integers(0, 1 << 30)))
        parts.append(p)
        tot += m
    return b"".join(parts)[:n]


def _archive(oracle, entries, level):
    frames = [workloads.libzstd_compress(e, level) for e in entries]
    bs = np.array([len(f) for f in frames], np.uint64)
    bo = np.concatenate([[0], np.cumsum(bs)[:-1]]).astype(np.uint64)
    us = np.array([len(e) for e in entries], np.uint64)
    oo = np.concatenate([[0], np.cumsum(us)[:-1]]).astype(np.uint64)
    ck = np.stack([np.frombuffer(oracle.blake3(e), dtype=np.uint8) for e in entries])
    blobs = np.frombuffer(b"".join(frames) + bytes(64), dtype=np.uint8)
    return dict(blobs=blobs, bo=bo, bs=bs, us=us, oo=oo, ck=ck, frames=frames)


def _run(ctx, A):
    import torch
    from znippy_amd import hip
    d_blobs = torch.from_numpy(A["blobs"].copy()).cuda()
    total = int(A["us"].sum())
    d_out = torch.zeros(total + 64, dtype=torch.uint8, device="cuda")
    rt = hip.RowTable(ctx, A["bo"], A["bs"], A["us"], A["oo"], None, A["ck"])
    first = None
    for rep in range(3):  # the work lists of the two-phase path are filled in a different order every run: every run must agree
        d_out.zero_()
        c, corrupt, status = rt.decode_verify(d_blobs, d_out)
        got = (dict(c), sorted(int(x) for x in corrupt), status.copy(), rt.digests()[status >= 0].copy(), d_out.cpu().numpy()[:total].copy())
        if first is None:
            first = got
        else:
            assert got[0] == first[0] and got[1] == first[1] and (got[2] == first[2]).all(), rep
            assert (got[3] == first[3]).all() and (got[4] == first[4]).all(), rep
    return c, corrupt, status, first[4], dict(ctx.kernel_times())


@pytest.mark.parametrize("level", [1, 3, 19])
def test_real_text_frames_two_phase_only(gpu_ctx_fz_only, oracle, level):
    data = _py_corpus(6 << 20)
    cuts = [0, 300_000, 300_000 + 131_073, 1_500_000, 1_500_000 + 262_144, 4_000_000, len(data)]
    entries = [data[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
    A = _archive(oracle, entries, level)
    c, corrupt, status, out, kt = _run(gpu_ctx_fz_only, A)
    assert c["corrupt_rows"] == 0 and c["decode_errors"] == 0, (c, corrupt, status)
    assert c["verified_bytes"] == len(data)
    assert out.tobytes() == data
    assert "zstd_batch_sequences" in kt and "zstd_batch_execute" in kt


@pytest.mark.parametrize("level", [1, 19])
def test_mixed_block_types_two_phase_only(gpu_ctx_fz_only, oracle, level):
    entries = [_mixed(900_000, 1), _mixed(2_500_000, 2), gen.pseudo_text(700_001, 4), bytes(1_000_000),
               gen.incompressible(3, 400_000) if hasattr(gen, "incompressible") else os.urandom(400_000)]
    A = _archive(oracle, entries, level)
    c, corrupt, status, out, kt = _run(gpu_ctx_fz_only, A)
    assert c["corrupt_rows"] == 0 and c["decode_errors"] == 0, (c, corrupt, status)
    assert out.tobytes() == b"".join(entries)


def test_corrupt_foreign_frames_same_verdicts_as_serial(gpu_ctx, oracle):
    """Damaged frames: whatever the two-phase path makes of them, the row verdicts equal the serial decoder's (it is
    the one that reports) and the oracle's."""
    import torch
    from znippy_amd import hip
    data = _py_corpus(2 << 20)
    entries = [data[i * 400_000:(i + 1) * 400_000] for i in range(5)]
    A = _archive(oracle, entries, 19)
    blobs = A["blobs"].copy()
    rng = np.random.default_rng(3)
    hit = {}
    for i in (1, 3):  # flip a byte somewhere inside the frame body
        at = int(A["bo"][i]) + int(rng.integers(20, int(A["bs"][i]) - 4))
        blobs[at] ^= 0x5A
        hit[i] = at
    A2 = dict(A, blobs=blobs)
    c, corrupt, status, out, kt = _run(gpu_ctx, A2)
    for i in range(5):
        frame = blobs[int(A["bo"][i]):int(A["bo"][i] + A["bs"][i])].tobytes()
        try:
            want = oracle.zstd_decompress(frame)
        except Exception:
            want = None
        if want is None or len(want) != len(entries[i]):
            assert status[i] < 0, (i, status[i])
        else:
            assert status[i] >= 0
            got = out[int(A["oo"][i]):int(A["oo"][i] + A["us"][i])].tobytes()
            assert got == want
            assert (i in [int(x) for x in corrupt]) == (want != entries[i])
    assert status[0] >= 0 and status[2] >= 0 and status[4] >= 0


def test_two_phase_matches_serial_bytes(gpu_ctx, oracle):
    """Same archive through the batch path (lane = block), through the round-2 two-phase path (wave = block, ZNIPPY_NO_BX)
    and through the serial decoder alone (ZNIPPY_NO_BX + ZNIPPY_NO_FZ): identical bytes and counters."""
    from znippy_amd import hip
    data = _py_corpus(3 << 20)
    entries = [data[:1_000_000], data[1_000_000:1_200_000], data[1_200_000:]]
    A = _archive(oracle, entries, 19)
    c1, _, s1, o1, kt1 = _run(gpu_ctx, A)

    def other(env):
        for k in env:
            os.environ[k] = "1"
        try:
            ctx2 = hip.Context(0)
        finally:
            for k in env:
                del os.environ[k]
        try:
            return _run(ctx2, A)
        finally:
            ctx2.close()
    c2, _, s2, o2, kt2 = other(["ZNIPPY_NO_BX"])
    c3, _, s3, o3, kt3 = other(["ZNIPPY_NO_BX", "ZNIPPY_NO_FZ"])
    assert c1 == c2 == c3 and (s1 == s2).all() and (s1 == s3).all() and (o1 == o2).all() and (o1 == o3).all()
    assert o1.tobytes() == data
    assert "zstd_batch_execute" in kt1 and "zstd_foreign_entropy" not in kt1
    assert "zstd_foreign_entropy" in kt2 and "zstd_batch_execute" not in kt2
    assert "zstd_foreign_entropy" not in kt3 and "zstd_batch_execute" not in kt3


def test_mutated_real_text_frames_agree_with_oracle(gpu_ctx, oracle):
    """Real-text multi-block frames at level 19 (Treeless literals, Repeat_Mode tables, repeat offsets across blocks),
    damaged anywhere: the oracle's verdict is the GPU's, accepted mutants decode to the oracle's bytes, and nothing is
    ever reported verified with different bytes (the harness of test_gpu_fuzz.py)."""
    from test_gpu_fuzz import _run as fuzz_run
    data = _py_corpus(1 << 20)
    bases = [(data[:400_000], 19), (data[400_000:400_000 + 262_145], 19), (data[700_000:1_000_000], 3)]
    fuzz_run(gpu_ctx, oracle, bases, 60, 4242, 10, 60)


def test_checksummed_and_small_window_frames(gpu_ctx, oracle):
    """Multi-block frames with a content-checksum trailer stay with the serial decoder (it verifies XXH64); a damaged
    trailer is a per-row checksum error, as with libzstd.  A small window log (matches never reach further back than
    128 KiB) goes through the two-phase path like any other frame."""
    import torch
    from znippy_amd import hip
    data = _py_corpus(1 << 20)
    e = [data[:500_000], data[500_000:900_000], data[200_000:700_000]]
    frames = [workloads.libzstd_compress_adv(e[0], 19, checksum=True), workloads.libzstd_compress_adv(e[1], 3, checksum=True),
              workloads.libzstd_compress_adv(e[2], 19, window_log=17)]
    bad = bytearray(frames[1]); bad[-1] ^= 0x40            # the trailer itself
    frames.append(bytes(bad)); e.append(e[1])
    for f, d in zip(frames[:3], e[:3]):
        assert oracle.zstd_decompress(f) == d
    bs = np.array([len(f) for f in frames], np.uint64)
    bo = np.concatenate([[0], np.cumsum(bs)[:-1]]).astype(np.uint64)
    us = np.array([len(d) for d in e], np.uint64)
    oo = np.concatenate([[0], np.cumsum(us)[:-1]]).astype(np.uint64)
    ck = np.stack([np.frombuffer(oracle.blake3(d), dtype=np.uint8) for d in e])
    d_blobs = torch.from_numpy(np.frombuffer(b"".join(frames) + bytes(64), dtype=np.uint8).copy()).cuda()
    d_out = torch.zeros(int(us.sum()) + 64, dtype=torch.uint8, device="cuda")
    rt = hip.RowTable(gpu_ctx, bo, bs, us, oo, None, ck)
    c, corrupt, status = rt.decode_verify(d_blobs, d_out)
    st = rt.foreign_stats()
    assert list(status[:3]) == [0, 0, 0] and status[3] == -7, status      # ZNIPPY_E_CHECKSUM
    assert c["decode_errors"] == 1 and c["corrupt_rows"] == 0
    out = d_out.cpu().numpy()
    for i in range(3):
        assert out[int(oo[i]):int(oo[i] + us[i])].tobytes() == e[i]
    assert st["frames"] == 1, st                                          # only the checksum-less frame took the two-phase path
