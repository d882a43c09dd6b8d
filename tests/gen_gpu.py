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
    """val_{k+1} = val_k * a + inc (mod 2^64), byte = val >> 33; int64 arithmetic wraps like u64.

    Built by doubling: the first k values and the coefficients of a k-step jump give the next k values in one
    multiply-add, so B values take log2(B) launches; the stream is then extended B at a time.  (The first version
    made 4 launches per 64 KiB — 32,000 for C4's 500 MiB — and crashed under rocprofv3 --pmc, inside the profiler's
    interception of a torch launch; see profiles/README.md.)"""
    a = 6364136223846793005
    mask = (1 << 64) - 1

    def s64(x):
        x &= mask
        return x - (1 << 64) if x >= (1 << 63) else x

    def t64(x):
        return torch.tensor(s64(x), dtype=torch.int64, device=device)

    if size == 0:
        return torch.empty(0, dtype=torch.uint8, device=device)
    B = 1 << 24
    cur = torch.tensor([s64((v0 * a + inc) & mask)], dtype=torch.int64, device=device)  # val_1
    A, Cc = a, inc                                   # coefficients of a jump by len(cur) = 1 step
    while cur.numel() < min(B, size):
        cur = torch.cat([cur, cur * t64(A) + t64(Cc)])
        A, Cc = (A * A) & mask, (Cc * A + Cc) & mask  # jump by twice as many steps
    out = torch.empty(size, dtype=torch.uint8, device=device)
    n0 = cur.numel()
    A_t, C_t = t64(A), t64(Cc)                       # jump by n0 steps
    pos = 0
    while pos < size:
        n = min(n0, size - pos)
        out[pos:pos + n] = ((cur[:n] >> 33) & 0xFF).to(torch.uint8)
        pos += n
        if pos < size:
            cur = cur * A_t + C_t
    return out


def random_lcg(size, device="cuda"):
    return _lcg(12345, size, 1, device)


def incompressible(seed, size, device="cuda"):
    v0 = (seed * 0x9E3779B97F4A7C15 + 1) & ((1 << 64) - 1)
    return _lcg(v0, size, 1442695040888963407, device)
