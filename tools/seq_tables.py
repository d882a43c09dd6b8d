"""Diagnostic: the sequences-section tables of a frame's first block (modes byte, normalised counts per table)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def read_ncount(buf, pos):
    bits = int.from_bytes(buf[pos:pos + 80], "little")
    bp = 0
    def rd(n):
        nonlocal bp
        v = (bits >> bp) & ((1 << n) - 1); bp += n
        return v
    tl = rd(4) + 5
    remaining, sym, out = (1 << tl) + 1, 0, {}
    while remaining > 1:
        thr = 1 << (remaining.bit_length() - 1); nbb = thr.bit_length()
        mx = 2 * thr - 1 - remaining
        low = (bits >> bp) & (thr - 1)
        if low < mx:
            v = low; bp += nbb - 1
        else:
            v = (bits >> bp) & (2 * thr - 1); bp += nbb
            if v >= thr: v -= mx
        cnt = v - 1
        remaining -= abs(cnt)
        if cnt: out[sym] = cnt
        sym += 1
        if cnt == 0:
            while True:
                f = rd(2); sym += f
                if f != 3: break
    return tl, out, pos + (bp + 7) // 8


def first_block_tables(frame):
    fhd = frame[4]
    p = 5 + {0: 1, 1: 2, 2: 4, 3: 8}[fhd >> 6]
    off = p + 3
    b0 = frame[off]; lt, sf = b0 & 3, (b0 >> 2) & 3
    h = int.from_bytes(frame[off:off + 5], "little")
    if lt <= 1:
        lh = 1 if (sf & 1) == 0 else (2 if sf == 1 else 3)
        regen = (h & 0xFF) >> 3 if (sf & 1) == 0 else ((h & 0xFFFF) >> 4 if sf == 1 else (h & 0xFFFFFF) >> 4)
        q = off + lh + (regen if lt == 0 else 1)
    else:
        lh, nb = (3, 10) if sf <= 1 else ((4, 14) if sf == 2 else (5, 18))
        q = off + lh + ((h >> (4 + nb)) & ((1 << nb) - 1))
    s0 = frame[q]
    hl = 1 if s0 < 128 else (2 if s0 < 255 else 3)
    nseq = s0 if s0 < 128 else (((s0 - 128) << 8) + frame[q + 1] if s0 < 255 else frame[q + 1] + (frame[q + 2] << 8) + 0x7F00)
    modes = frame[q + hl]
    pos = q + hl + 1
    res = {"nseq": nseq, "modes": modes}
    for name, sh in (("LL", 6), ("OF", 4), ("ML", 2)):
        m = (modes >> sh) & 3
        if m == 1:
            res[name] = ("rle", frame[pos]); pos += 1
        elif m == 2:
            tl, tab, pos = read_ncount(frame, pos)
            res[name] = (tl, tab)
        else:
            res[name] = ("predefined" if m == 0 else "repeat",)
    return res


if __name__ == "__main__":
    import importlib.util
    from znippy_amd import hip
    spec = importlib.util.spec_from_file_location("t", os.path.join(ROOT, "tests", "test_gpu_levels.py"))
    t = importlib.util.module_from_spec(spec); spec.loader.exec_module(t)
    ctx = hip.Context(0); ctx.set_level(19)
    for name, data in (("rle_records", t._rle_records(400, 6)), ("records", t._records(6000, 4))):
        print(name, first_block_tables(ctx.compress(data)))
