"""The role-split persistent kernel of the small-row read path (csrc/fused_small.hip, k_fused_roles: one loader +
seven hasher waves per workgroup) against the oracle's restated read loop (decompress.rs:L135-190): bytes, digests,
counters, corrupt rows, per-row status — for tables it takes entirely (every row a whole-leaf row of the recognised
periodic shape), tables it hands over entirely, and every mixture, tile by tile."""
import numpy as np
import pytest

import gen
from test_gpu_decode import _build_archive, _run_gpu

pytestmark = pytest.mark.gpu


def _period(p, n, seed=0):
    rng = np.random.default_rng(seed * 1000 + p)
    pat = rng.integers(0, 256, size=p, dtype=np.uint8).tobytes()
    return (pat * (n // p + 1))[:n]


def _check(oracle, ctx, entries, level=19, skip=None, pad=0, mutate=None):
    arch = _build_archive(oracle, entries, level=level, skip=skip)
    if mutate:
        mutate(arch)
    counters, corrupt, status, out, rt = _run_gpu(ctx, arch, pad_blobs=pad)
    n = len(entries)
    bitmap = np.packbits(arch["compressed"].astype(bool), bitorder="little")
    want_out = np.zeros(int(arch["usize"].sum()), dtype=np.uint8)
    want, want_corrupt = oracle.decompress_rows(arch["blobs"], arch["blob_offset"], arch["blob_size"], arch["usize"],
                                                arch["out_off"], bitmap, arch["checksum"], 0, n, out=want_out)
    assert counters == want
    assert list(corrupt) == list(want_corrupt)
    assert np.array_equal(out, want_out)
    return arch, status, rt


@pytest.mark.parametrize("level", [1, 3, 19])
def test_c2_shape_table_is_taken_whole(gpu_ctx_roles, oracle, level):
    """5,000 x 10 KiB text rows (libzstd frames at three levels): all tiles are fast tiles."""
    entries = [gen.text(10240)] * 5000
    arch, status, rt = _check(oracle, gpu_ctx_roles, entries, level=level, pad=3)
    assert (status == 0).all()
    assert np.array_equal(rt.digests(), arch["checksum"])
    names = [k for k, _ in gpu_ctx_roles.kernel_times()]
    assert "decode_verify_roles" in names


def test_periods_and_row_lengths(gpu_ctx_roles, oracle):
    """Whole-leaf rows of every leaf count 1..64 with periods from 1 byte to beyond the 960-byte limit of the
    recognised shape, packed blobs (arbitrary frame alignment)."""
    entries = []
    periods = [1, 2, 3, 7, 15, 16, 17, 45, 64, 100, 251, 255, 256, 400, 511, 700, 959, 960, 961, 1500]
    for k in range(1, 65):
        p = periods[k % len(periods)]
        entries.append(_period(p, k * 1024, seed=k))
    for p in periods:
        entries.append(_period(p, 10240, seed=99))
    arch, status, rt = _check(oracle, gpu_ctx_roles, entries, level=19, pad=1)
    assert (status == 0).all()
    assert np.array_equal(rt.digests(), arch["checksum"])


def test_mixture_of_fast_tiles_and_everything_else(gpu_ctx_roles, gpu_ctx, oracle):
    """Fast tiles next to tiles the loader must hand over: ragged rows, stored rows, entropy-coded rows, empty rows,
    tiles of many tiny rows, big rows (slices), broken frames, wrong checksums — same results as the plain context
    and the oracle loop."""
    rng = np.random.default_rng(5)
    entries, skip = [], []
    for i in range(900):
        kind = int(rng.integers(0, 10))
        if kind <= 3:
            e = gen.text(10240)                                   # fast
        elif kind == 4:
            e = gen.text(int(rng.integers(1, 30000)))            # ragged
        elif kind == 5:
            e = gen.incompressible(i, int(rng.integers(0, 20000)))
        elif kind == 6:
            e = gen.pseudo_text(int(rng.integers(100, 50000)), seed=i)
        elif kind == 7:
            e = b""
        elif kind == 8:
            e = _period(int(rng.integers(1, 1200)), 1024 * int(rng.integers(1, 20)), seed=i)   # mostly fast
        else:
            e = gen.binary(1024)                                  # tiny whole-leaf rows: many per tile
        entries.append(e)
        skip.append(1 if kind == 5 and i % 2 else 0)
    entries.append(gen.text(3 << 20)); skip.append(0)           # big compressed row: block items
    entries.append(gen.incompressible(3, 1 << 20)); skip.append(1)

    def mutate(arch):
        ck = arch["checksum"].copy()
        for r in (0, 5, 17, 400):
            ck[r, 7] ^= 0x40                                      # wrong expected digests
        arch["checksum"] = ck
        blobs = arch["blobs"].copy()
        for r in (3, 250, 251):                                   # broken frames (compressed rows only)
            if arch["compressed"][r] and arch["blob_size"][r] > 8:
                blobs[int(arch["blob_offset"][r]) + 2] ^= 0xFF
        arch["blobs"] = blobs

    arch, status_r, rt_r = _check(oracle, gpu_ctx_roles, entries, level=19, skip=skip, pad=7, mutate=mutate)
    arch2, status_p, rt_p = _check(oracle, gpu_ctx, entries, level=19, skip=skip, pad=7, mutate=mutate)
    assert np.array_equal(status_r, status_p)
    ok = status_r == 0
    assert np.array_equal(rt_r.digests()[ok], rt_p.digests()[ok])


def test_roles_kernel_respects_host_verdicts_and_out_cap(gpu_ctx_roles, oracle):
    """Rows ruled out by the host (source outside the blob region, destination too short) inside otherwise fast
    tiles: the tile goes to the slow list, the row keeps its error code, the rest decodes."""
    import torch
    from znippy_amd import _lib, hip
    n, sz = 120, 10240
    entries = [gen.text(sz)] * n
    arch = _build_archive(oracle, entries, level=19)
    bo = arch["blob_offset"].copy()
    bo[10] = np.uint64(len(arch["blobs"]) + 999)
    bo[77] = np.uint64(2**63)
    d_blobs = torch.from_numpy(arch["blobs"].copy()).cuda()
    total = n * sz
    cut = total - 3 * sz + 100                                   # the last three rows do not fit
    d_out = torch.full((total + 64,), 0xCD, dtype=torch.uint8, device="cuda")
    bitmap = np.packbits(arch["compressed"].astype(bool), bitorder="little")
    rt = hip.RowTable(gpu_ctx_roles, bo, arch["blob_size"], arch["usize"], arch["out_off"], bitmap, arch["checksum"])
    counters, corrupt, status = rt.decode_verify(d_blobs, d_out, out_cap=cut, blob_cap=len(arch["blobs"]))
    want = np.zeros(n, np.int32)
    want[[10, 77]] = _lib.E_CORRUPT
    want[-3:] = _lib.E_DST_SMALL
    assert np.array_equal(status, want)
    assert counters["decode_errors"] == 5 and counters["verified_bytes"] == (n - 5) * sz and counters["corrupt_rows"] == 0
    host = d_out.cpu().numpy()
    for r in range(n):
        row = host[r * sz:(r + 1) * sz]
        if want[r] == 0:
            assert row.tobytes() == entries[r]
        else:
            assert (row == 0xCD).all()
    assert (host[total:] == 0xCD).all()


def test_mixed_row_shapes_decode_the_same_every_time(gpu_ctx_roles, oracle):
    """Rows of several whole-leaf sizes in random order (6 x 10-leaf tiles between tiles of other shapes: the parent trees
    in the idle lanes, the group fold, the left-over list and the wave-exit flush all in one table), five runs: every
    digest equals the oracle's, every run equals the first."""
    import torch
    from znippy_amd import hip
    rng = np.random.default_rng(77)
    sizes = [10240, 10240, 10240, 20480, 4096, 8192, 10240, 1024, 30720]
    chunks = {s: gen.text(s) for s in set(sizes)}
    frames = {s: gpu_ctx_roles.compress(chunks[s]) for s in set(sizes)}
    want = {s: np.frombuffer(oracle.blake3(chunks[s]), dtype=np.uint8) for s in set(sizes)}
    # runs of equal rows (so that whole tiles of one shape occur) separated by random single rows
    order = []
    while len(order) < 20000:
        s = sizes[int(rng.integers(0, len(sizes)))]
        order += [s] * int(rng.integers(1, 40))
    order = order[:20000]
    us = np.array(order, np.uint64)
    bs = np.array([len(frames[s]) for s in order], np.uint64)
    bo = np.concatenate([[0], np.cumsum(bs)[:-1]]).astype(np.uint64)
    oo = np.concatenate([[0], np.cumsum(us)[:-1]]).astype(np.uint64)
    ck = np.stack([want[s] for s in order])
    d_blobs = torch.from_numpy(np.frombuffer(b"".join(frames[s] for s in order) + bytes(64), dtype=np.uint8).copy()).cuda()
    d_out = torch.zeros(int(us.sum()) + 64, dtype=torch.uint8, device="cuda")
    rt = hip.RowTable(gpu_ctx_roles, bo, bs, us, oo, None, ck)
    ref = torch.from_numpy(np.frombuffer(b"".join(chunks[s] for s in order), dtype=np.uint8).copy()).cuda()
    for rep in range(5):
        d_out.zero_()
        c, corrupt, status = rt.decode_verify(d_blobs, d_out)
        assert c["corrupt_rows"] == 0 and c["decode_errors"] == 0 and c["verified_bytes"] == int(us.sum()), (rep, c, corrupt[:8])
        assert (rt.digests() == ck).all(), rep
        assert torch.equal(d_out[:int(us.sum())], ref), rep


def test_lean_runs_notice_changed_blobs(oracle):
    """A table whose last run left nothing behind the role-split kernel is run lean (memset + that kernel + verify: the three
    launches behind it would only look at empty lists).  If the blobs have changed since — a frame the kernel no longer
    recognises, a damaged one — the verify kernel flags the run and whoever reads its results runs the table again in full:
    the caller sees exactly what a context without lean runs (ZNIPPY_NO_LEAN=1) reports.  Reference: the read loop looks at
    every row on every pass (decompress.rs:L135-190); nothing may depend on what an earlier pass over the table found."""
    import os
    import torch
    import gen
    from znippy_amd import hip
    n = 6 * 700
    data = gen.text(10240)
    frame = oracle.libzstd_compress(data, 19)
    other = gen.pseudo_text(10240, seed=5)                    # same size, another content: decodes, verify flags it
    other_frame = oracle.libzstd_compress(other, 3)
    assert len(other_frame) > len(frame)
    slot = len(other_frame) + 7                               # every row's blob slot is big enough for either frame
    bo = (np.arange(n, dtype=np.uint64) * np.uint64(slot))
    bs = np.full(n, len(frame), np.uint64)
    us = np.full(n, 10240, np.uint64)
    oo = np.arange(n, dtype=np.uint64) * np.uint64(10240)
    ck = np.tile(np.frombuffer(oracle.blake3(data), dtype=np.uint8), (n, 1))
    blob = np.zeros(n * slot + 64, np.uint8)
    for i in range(n):
        blob[i * slot:i * slot + len(frame)] = np.frombuffer(frame, np.uint8)

    def run(env):
        old = {k: os.environ.get(k) for k in ("ZNIPPY_ROLES_MIN", "ZNIPPY_NO_LEAN")}
        os.environ["ZNIPPY_ROLES_MIN"] = "1"
        os.environ.update(env)
        try:
            ctx = hip.Context(0)
        finally:
            for k, v in old.items():
                if v is None: os.environ.pop(k, None)
                else: os.environ[k] = v
        d_blobs = torch.from_numpy(blob.copy()).cuda()
        d_out = torch.zeros(n * 10240 + 64, dtype=torch.uint8, device="cuda")
        rt = hip.RowTable(ctx, bo, bs, us, oo, None, ck)
        seen = []
        for step in range(6):
            if step == 3:     # row 1234 becomes a frame of other content (longer blob: the table says the old size -> truncated -> error),
                d_blobs[1234 * slot:1234 * slot + len(other_frame)] = torch.from_numpy(np.frombuffer(other_frame, np.uint8).copy()).cuda()
                d_blobs[77 * slot + 20] ^= 0x55   # ... row 77 is damaged in its sequence section
            d_out.zero_()
            torch.cuda.synchronize()
            c, corrupt, status = rt.decode_verify(d_blobs, d_out)
            kt = dict(ctx.kernel_times())
            seen.append((dict(c), sorted(int(x) for x in corrupt), status.copy(), rt.digests().copy(), d_out[:n * 10240].cpu().numpy().copy(),
                         "blake3_second_pass" in kt))
        rt.close(); ctx.close()
        return seen

    lean, full = run({}), run({"ZNIPPY_NO_LEAN": "1"})
    assert [s[5] for s in full] == [True] * 6
    assert lean[0][5] and not lean[2][5]          # the first run is a full one, the third a lean one
    for a, b in zip(lean, full):
        assert a[0] == b[0] and a[1] == b[1] and (a[2] == b[2]).all() and (a[3] == b[3]).all() and (a[4] == b[4]).all()
    assert lean[2][0]["corrupt_rows"] == 0 and lean[2][0]["decode_errors"] == 0
    bad = lean[3][0]["corrupt_rows"] + lean[3][0]["decode_errors"]
    assert bad == 2, lean[3][0]                   # both changed rows are reported by the run that met them


def test_lean_runs_in_a_pipeline_with_alternating_buffers(oracle):
    """The read loop keeps two runs in flight and reads run k's counters while run k + 1 executes
    (znippy_rows_results_lagged); here the runs alternate between two output buffers and a frame is damaged while lean runs
    are queued (the role-split kernel no longer recognises it: the row is handed over, nobody is there to take it, the run
    comes back flagged).  A flagged run is repeated with the arguments IT was given: the counters of every run and both
    buffers' bytes equal a ZNIPPY_NO_LEAN context's."""
    import os
    import torch
    import gen
    from znippy_amd import hip
    n = 6 * 500
    data = gen.text(10240)
    frame = oracle.libzstd_compress(data, 19)
    slot = len(frame) + 9
    bo = np.arange(n, dtype=np.uint64) * np.uint64(slot)
    bs = np.full(n, len(frame), np.uint64)
    us = np.full(n, 10240, np.uint64)
    oo = np.arange(n, dtype=np.uint64) * np.uint64(10240)
    ck = np.tile(np.frombuffer(oracle.blake3(data), dtype=np.uint8), (n, 1))
    blob = np.zeros(n * slot + 64, np.uint8)
    for i in range(n):
        blob[i * slot:i * slot + len(frame)] = np.frombuffer(frame, np.uint8)

    def run(env):
        old = {k: os.environ.get(k) for k in ("ZNIPPY_ROLES_MIN", "ZNIPPY_NO_LEAN")}
        os.environ["ZNIPPY_ROLES_MIN"] = "1"
        os.environ.update(env)
        try:
            ctx = hip.Context(0)
        finally:
            for k, v in old.items():
                if v is None: os.environ.pop(k, None)
                else: os.environ[k] = v
        d_blobs = torch.from_numpy(blob.copy()).cuda()
        outs = [torch.zeros(n * 10240 + 64, dtype=torch.uint8, device="cuda") for _ in range(2)]
        rt = hip.RowTable(ctx, bo, bs, us, oo, None, ck)
        counters, lean_seen = [], []
        for step in range(9):
            if step == 5:   # rows 42 and 2900: a byte of the sequence section flipped
                torch.cuda.synchronize()
                d_blobs[42 * slot + 20] ^= 0x55
                d_blobs[2900 * slot + len(frame) - 3] ^= 0x0F
                torch.cuda.synchronize()
            rt.decode_verify_async(d_blobs, outs[step & 1])
            lean_seen.append("blake3_second_pass" not in dict(ctx.kernel_times()))
            if step >= 1:
                counters.append(rt.results_lagged(1))
        counters.append(rt.results_lagged(0))
        ctx.sync()
        got = [o[:n * 10240].cpu().numpy().copy() for o in outs]
        rt.close(); ctx.close()
        return counters, got, lean_seen

    lean, full = run({}), run({"ZNIPPY_NO_LEAN": "1"})
    assert not any(full[2]) and any(lean[2][2:5]), (lean[2], full[2])   # lean runs were queued before the change
    assert lean[0] == full[0], (lean[0], full[0])
    assert sum(c["corrupt_rows"] + c["decode_errors"] for c in lean[0][:4]) == 0
    assert all(c["corrupt_rows"] + c["decode_errors"] == 2 for c in lean[0][5:]), lean[0]
    for a, b in zip(lean[1], full[1]):
        assert (a == b).all()
    for o in lean[1]:
        assert o[:42 * 10240].tobytes() == data * 42 and o[43 * 10240:2900 * 10240].tobytes() == data * (2900 - 43)


def test_lean_runs_of_big_row_tables_notice_changed_blobs(oracle):
    """The same for a table of big multi-block rows (this encoder's frames of periodic text: the fused block kernel writes and
    hashes every block, the serial block decoder and the serial decoder behind it find empty lists): the third run leaves
    their launches out; then a block of one frame is damaged and a frame is replaced by a libzstd frame (blocks that refer
    to each other: the batch path's business) — the run that meets them reports what a ZNIPPY_NO_LEAN context reports."""
    import os
    import torch
    import gen
    from znippy_amd import hip
    n, size = 24, 4 * 131072   # whole blocks: a short last block is the serial block decoder's every time (no lean runs then)
    rows = [gen.text(size) if i % 3 else gen.binary(size) for i in range(n)]
    ctx0 = hip.Context(0)
    frames = [ctx0.compress(r_) for r_ in rows]
    ctx0.close()
    foreign = oracle.libzstd_compress(rows[7], 3)
    slot = max(max(len(f) for f in frames), len(foreign)) + 11
    bo = np.arange(n, dtype=np.uint64) * np.uint64(slot)
    bs = np.array([len(f) for f in frames], np.uint64)
    us = np.full(n, size, np.uint64)
    oo = np.arange(n, dtype=np.uint64) * np.uint64(size)
    ck = np.stack([np.frombuffer(oracle.blake3(r_), dtype=np.uint8) for r_ in rows])
    blob = np.zeros(n * slot + 64, np.uint8)
    for i, f in enumerate(frames):
        blob[i * slot:i * slot + len(f)] = np.frombuffer(f, np.uint8)

    def run(env):
        old = os.environ.get("ZNIPPY_NO_LEAN")
        os.environ.update(env)
        try:
            ctx = hip.Context(0)
        finally:
            if old is None: os.environ.pop("ZNIPPY_NO_LEAN", None)
            else: os.environ["ZNIPPY_NO_LEAN"] = old
        d_blobs = torch.from_numpy(blob.copy()).cuda()
        d_out = torch.zeros(n * size + 64, dtype=torch.uint8, device="cuda")
        rt = hip.RowTable(ctx, bo, bs, us, oo, None, ck)
        seen = []
        for step in range(6):
            if step == 3:
                d_blobs[3 * slot + 40] ^= 0x7F                      # row 3: a byte of its first block
            d_out.zero_()
            torch.cuda.synchronize()
            c, corrupt, status = rt.decode_verify(d_blobs, d_out)
            kt = dict(ctx.kernel_times())
            seen.append((dict(c), sorted(int(x) for x in corrupt), status.copy(), rt.digests().copy(), d_out[:n * size].cpu().numpy().copy(),
                         "zstd_decode_blocks" in kt))
        rt.close(); ctx.close()
        return seen

    lean, full = run({}), run({"ZNIPPY_NO_LEAN": "1"})
    assert all(s[5] for s in full) and lean[0][5] and not lean[2][5], [s[5] for s in lean]
    for a, b in zip(lean, full):
        assert a[0] == b[0] and a[1] == b[1] and (a[2] == b[2]).all() and (a[3] == b[3]).all() and (a[4] == b[4]).all()
    assert lean[2][0]["corrupt_rows"] + lean[2][0]["decode_errors"] == 0
    assert lean[3][0]["corrupt_rows"] + lean[3][0]["decode_errors"] == 1, lean[3][0]


def test_lean_runs_of_mixed_tables_notice_changed_blobs(oracle):
    """Small compressed rows beside big stored rows (BASELINE configs[4]'s shape): when the last run handed nothing over, the
    small rows' kernel runs beside the second hash pass instead of in front of it and the serial decoder is not launched.
    A small row is then damaged: the run that meets it reports what a ZNIPPY_NO_LEAN context reports, bytes included."""
    import os
    import torch
    import gen
    from znippy_amd import hip
    rng = np.random.default_rng(11)
    small = [gen.text(int(rng.integers(1024, 8192))) for _ in range(900)]
    big = [gen.incompressible(20 + i, (1 << 20) + 4096 * i) for i in range(6)]
    entries = small + big
    ctx0 = hip.Context(0)
    frames = [ctx0.compress(e) for e in small] + big                  # the big rows are stored as they are
    ctx0.close()
    comp = np.array([1] * len(small) + [0] * len(big), np.uint8)
    n = len(entries)
    bs = np.array([len(f) for f in frames], np.uint64)
    bo = (np.cumsum(bs) - bs).astype(np.uint64)
    us = np.array([len(e) for e in entries], np.uint64)
    oo = (np.cumsum(us) - us).astype(np.uint64)
    ck = np.stack([np.frombuffer(oracle.blake3(e), dtype=np.uint8) for e in entries])
    bm = np.packbits(comp.astype(bool), bitorder="little")
    blob = np.frombuffer(b"".join(frames) + bytes(64), np.uint8)
    total = int(us.sum())

    def run(env):
        old = os.environ.get("ZNIPPY_NO_LEAN")
        os.environ.update(env)
        try:
            ctx = hip.Context(0)
        finally:
            if old is None: os.environ.pop("ZNIPPY_NO_LEAN", None)
            else: os.environ["ZNIPPY_NO_LEAN"] = old
        d_blobs = torch.from_numpy(blob.copy()).cuda()
        d_out = torch.zeros(total + 64, dtype=torch.uint8, device="cuda")
        rt = hip.RowTable(ctx, bo, bs, us, oo, bm, ck)
        seen = []
        for step in range(6):
            if step == 3:
                d_blobs[int(bo[100]) + int(bs[100]) - 2] ^= 0x3C      # row 100: a byte of its sequence section
            d_out.zero_()
            torch.cuda.synchronize()
            c, corrupt, status = rt.decode_verify(d_blobs, d_out)
            kt = dict(ctx.kernel_times())
            seen.append((dict(c), sorted(int(x) for x in corrupt), status.copy(), rt.digests().copy(), d_out[:total].cpu().numpy().copy(),
                         sorted(kt)))
        rt.close(); ctx.close()
        return seen

    lean, full = run({}), run({"ZNIPPY_NO_LEAN": "1"})
    gen_in = lambda names: "zstd_decode_general" in names or "zstd_decode_fallback" in names
    assert all(gen_in(s[5]) for s in full) and gen_in(lean[0][5]) and not gen_in(lean[2][5]), ([s[5] for s in lean], full[2][5])
    for a, b in zip(lean, full):
        assert a[0] == b[0] and a[1] == b[1] and (a[2] == b[2]).all() and (a[3] == b[3]).all() and (a[4] == b[4]).all()
    assert lean[2][0]["corrupt_rows"] + lean[2][0]["decode_errors"] == 0 and lean[2][4].tobytes() == b"".join(entries)
    assert lean[3][0]["corrupt_rows"] + lean[3][0]["decode_errors"] == 1, lean[3][0]
