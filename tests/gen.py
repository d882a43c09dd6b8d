"""Workload generators restated from the reference's test crate (inputs are regenerated,
never stored):
  generate_text_data    tests/tests/perf_bench.rs:L74-77   (45-byte phrase cycled)
  generate_binary_data  tests/tests/perf_bench.rs:L79-81   (i % 251)
  generate_random_data  tests/tests/perf_bench.rs:L83-92   (LCG seed 12345, byte = val >> 33)
  incompressible        tests/tests/repro_crate.rs:L8-16   (per-seed LCG)
plus `pseudo_text`, a seeded Zipf-ish word stream (not from the reference) that makes
libzstd emit Huffman literals + FSE-compressed sequence tables, so the decoder's entropy
paths are exercised by something less trivial than a 45-byte period.
"""
import numpy as np

PHRASE = b"The quick brown fox jumps over the lazy dog. "


def text(size: int) -> bytes:
    reps = size // len(PHRASE) + 1
    return (PHRASE * reps)[:size]


def binary(size: int) -> bytes:
    return (np.arange(size, dtype=np.uint64) % 251).astype(np.uint8).tobytes()


def _lcg_bytes(val: int, size: int, inc: int) -> bytes:
    # val_{k} = a^k * v0 + inc * (a^{k-1} + ... + 1)  (mod 2^64); vectorised by block doubling
    a = 6364136223846793005
    mask = (1 << 64) - 1
    out = np.empty(size, dtype=np.uint8)
    # advance sequentially in python for small sizes, vectorised jump-ahead for large
    if size <= 4096:
        v = val
        for i in range(size):
            v = (v * a + inc) & mask
            out[i] = (v >> 33) & 0xFF
        return out.tobytes()
    # vectorised: compute first B values sequentially, then jump-ahead by B using (A, C) = step^B
    B = 4096
    vals = np.empty(B, dtype=np.uint64)
    v = val
    for i in range(B):
        v = (v * a + inc) & mask
        vals[i] = v
    # jump-ahead coefficients for B steps
    A, Cc = 1, 0
    for _ in range(B):
        A = (A * a) & mask
        Cc = (Cc * a + inc) & mask
    A = np.uint64(A)
    Cc = np.uint64(Cc)
    pos = 0
    with np.errstate(over="ignore"):
        while pos < size:
            n = min(B, size - pos)
            out[pos:pos + n] = ((vals[:n] >> np.uint64(33)) & np.uint64(0xFF)).astype(np.uint8)
            vals = vals * A + Cc
            pos += n
    return out.tobytes()


def random_lcg(size: int) -> bytes:
    return _lcg_bytes(12345, size, 1)


def incompressible(seed: int, size: int) -> bytes:
    v0 = (seed * 0x9E3779B97F4A7C15 + 1) & ((1 << 64) - 1)
    return _lcg_bytes(v0, size, 1442695040888963407)


_WORDS = None


def pseudo_text(size: int, seed: int = 1) -> bytes:
    """Seeded word soup with a Zipf-like distribution, punctuation and some numbers."""
    global _WORDS
    rng = np.random.default_rng(seed)
    if _WORDS is None:
        wr = np.random.default_rng(7)
        letters = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", dtype=np.uint8)
        p = 1.0 / np.arange(1, 27)
        p /= p.sum()
        _WORDS = []
        for _ in range(3000):
            n = int(wr.integers(1, 11))
            _WORDS.append(bytes(wr.choice(letters, size=n, p=p)))
    nw = size // 4 + 16
    ranks = np.minimum(rng.zipf(1.3, size=nw) - 1, len(_WORDS) - 1)
    parts = []
    total = 0
    for i, r in enumerate(ranks):
        w = _WORDS[int(r)]
        if i % 11 == 10:
            w = w + b"."
        if i % 53 == 52:
            w = w + b"\n" + str(int(rng.integers(0, 100000))).encode()
        parts.append(w)
        total += len(w) + 1
        if total >= size:
            break
    return b" ".join(parts)[:size].ljust(size, b" ")
