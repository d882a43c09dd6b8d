"""Device-side twins of tests/gen.py for the BASELINE configs that are too big to generate with
numpy in a test run (2 GiB text, 500 MiB LCG).  Each is checked against gen.py on a prefix."""
import numpy as np
import torch

import gen


def text(size, device="cuda"):
    phrase = torch.from_numpy(np.frombuffer(gen.PHRASE, dtype=np.uint8).copy()).to(device)
    reps = size // len(gen.PHRASE) + 1
    return phrase.repeat(reps)[:size].contiguous()


def binary(size, device="cuda"):
    return (torch.arange(size, device=device, dtype=torch.int64) % 251).to(torch.uint8)


def _lcg(v0, size, inc, device="cuda"):
    """val_{k+1} = val_k * a + inc (mod 2^64), byte = val >> 33; int64 arithmetic wraps like u64."""
    a = 6364136223846793005
    mask = (1 << 64) - 1

    def s64(x):
        x &= mask
        return x - (1 << 64) if x >= (1 << 63) else x

    B = 1 << 16
    # first B values sequentially on the host (python ints), then jump-ahead by B on the device
    vals = np.empty(B, dtype=np.int64)
    v = v0
    for i in range(B):
        v = (v * a + inc) & mask
        vals[i] = s64(v)
    A, Cc = 1, 0
    for _ in range(B):
        A = (A * a) & mask
        Cc = (Cc * a + inc) & mask
    cur = torch.from_numpy(vals).to(device)
    A_t = torch.tensor(s64(A), dtype=torch.int64, device=device)
    C_t = torch.tensor(s64(Cc), dtype=torch.int64, device=device)
    out = torch.empty(size, dtype=torch.uint8, device=device)
    pos = 0
    while pos < size:
        n = min(B, size - pos)
        out[pos:pos + n] = ((cur[:n] >> 33) & 0xFF).to(torch.uint8)
        cur = cur * A_t + C_t
        pos += n
    return out


def random_lcg(size, device="cuda"):
    return _lcg(12345, size, 1, device)


def incompressible(seed, size, device="cuda"):
    v0 = (seed * 0x9E3779B97F4A7C15 + 1) & ((1 << 64) - 1)
    return _lcg(v0, size, 1442695040888963407, device)
