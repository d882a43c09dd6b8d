"""Device-side twins of tests/gen.py for the BASELINE configs that are too big to generate with
numpy in a test run (2 GiB text, 500 MiB LCG).  Each is checked against gen.py on a prefix."""
import numpy as np
import torch

import gen


def text(size, device="cuda", start=0):
    """bytes [start, start + size) of the cycled phrase"""
    phrase = torch.from_numpy(np.frombuffer(gen.PHRASE, dtype=np.uint8).copy()).to(device)
    ph = start % len(gen.PHRASE)
    reps = (size + ph) // len(gen.PHRASE) + 1
    return phrase.repeat(reps)[ph:ph + size].contiguous()


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


def _jump(v0, inc, steps):
    """the LCG's state after `steps` steps from v0 (square-and-multiply on the affine map, host integers)"""
    a, mask = 6364136223846793005, (1 << 64) - 1
    A, Cc = 1, 0            # identity
    pa, pc = a, inc         # one step
    while steps:
        if steps & 1:
            A, Cc = (pa * A) & mask, (pa * Cc + pc) & mask
        pa, pc = (pa * pa) & mask, (pa * pc + pc) & mask
        steps >>= 1
    return (A * v0 + Cc) & mask


def random_lcg(size, device="cuda", start=0):
    return _lcg(_jump(12345, 1, start), size, 1, device)


def incompressible(seed, size, device="cuda", start=0):
    """bytes [start, start + size) of the seed's stream"""
    v0 = (seed * 0x9E3779B97F4A7C15 + 1) & ((1 << 64) - 1)
    inc = 1442695040888963407
    return _lcg(_jump(v0, inc, start), size, inc, device)
