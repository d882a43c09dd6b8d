"""BASELINE.json's configurations as device-resident Round batches, shared by bench.py and the
full-size -m gpu tests.  Inputs are regenerated from the reference's generators (tests/gen.py, tests/gen_gpu.py),
never stored."""
import numpy as np

import gen
import gen_gpu

SLICE = 8 << 20          # stream_packer.rs:L31
SLOT = 200 << 20         # slot_packer.rs:L30 (BASELINE configs[2] calls these "Magazine slices")


def c5_layout(scale=1):
    """Synthetic stand-in for the 5 GB / 5k-file artifact repo: 3,500 .xml text files of 1-8 KiB
    (compress_dir_bench.rs:L45-68), 1,400 .jar of 100 KiB..2 MiB and 100 .jar of 20..69.5 MiB of incompressible
    bytes (repro_crate.rs:L8-16; store path), stream chunking (8 MiB slices).  -> (xml sizes, jar sizes).
    scale > 1: 1/scale of the files, the big jars 1/8 of their size (multi-rank rehearsals: "c5small")."""
    xml = [1024 + (i % 8) * 1024 for i in range(3500 // scale)]
    big = (20 << 20) if scale == 1 else (20 << 20) // 8
    jars = [100 * 1024 + (i % 20) * 100 * 1024 for i in range(1400 // scale)] + [big + i * (1 << 19) for i in range(max(100 // scale, 1))]
    return xml, jars


def layout(name):
    """-> dict(lens, skip, name, gen): the Rounds of one configuration; gen(torch, b0, b1) makes bytes [b0, b1) of the
    staging buffer on the device (a rank of a strong-scaling run builds only its own share)."""
    if name in ("c2", "c2small"):
        n = 100_000 if name == "c2" else 2_000
        label = ("100k x 10KiB text chunks (BASELINE configs[1])" if name == "c2"
                 else "2k x 10KiB text chunks (reduced; NOT the headline config)")

        def g(torch, b0, b1):  # every chunk is the same 10 KiB of the cycled phrase, restarted per chunk
            chunk = torch.from_numpy(np.frombuffer(gen.text(10 * 1024), dtype=np.uint8).copy()).cuda()
            c0, c1 = b0 // 10240, (b1 + 10239) // 10240
            return chunk.repeat(max(c1 - c0, 1))[b0 - c0 * 10240:b0 - c0 * 10240 + (b1 - b0)].contiguous()
        return dict(lens=np.full(n, 10240, np.uint64), skip=None, name=label, gen=g)
    if name in ("c3", "c3slot"):
        size = 2 << 30
        if name == "c3":
            lens = np.full(size // SLICE, SLICE, np.uint64)
            label = "single 2 GiB text file, 256 x 8 MiB slices (BASELINE configs[2] at the reference's slice size)"
        else:
            lens = np.array([SLOT] * (size // SLOT) + ([size % SLOT] if size % SLOT else []), dtype=np.uint64)
            label = "single 2 GiB text file, 11 slices of <= 200 MiB (BASELINE configs[2] as worded there)"
        return dict(lens=lens, skip=None, name=label, gen=lambda torch, b0, b1: gen_gpu.text(b1 - b0, start=b0))
    if name in ("c5", "c5small", "c5text"):
        xml, jars = c5_layout(16 if name == "c5small" else 1)
        lens, skip = list(xml), [0] * len(xml)
        for j in jars:
            for o in range(0, j, SLICE):
                lens.append(min(SLICE, j - o))
                skip.append(1)
        nx, nj = sum(xml), sum(jars)

        real = None
        if name == "c5text":  # the xml files hold REAL text (sources found in the image; word soup where it has none): nothing periodic
            raw = b"".join(image_corpus("text", nx + (1 << 20), whole_files=False))
            if len(raw) < nx:
                raw += gen.pseudo_text(nx - len(raw), seed=13)
            real = np.frombuffer(raw[:nx], dtype=np.uint8)

        def g(torch, b0, b1):  # [text of all xml files | ONE LCG stream cut into the jars]
            parts = []
            if b0 < nx and real is not None:
                parts.append(torch.from_numpy(real[b0:min(b1, nx)].copy()).cuda())
            elif b0 < nx:
                parts.append(gen_gpu.text(min(b1, nx) - b0, start=b0))
            if b1 > nx:
                j0 = max(b0, nx) - nx
                parts.append(gen_gpu.incompressible(7, b1 - nx - j0, start=j0))
            return torch.cat(parts) if len(parts) > 1 else parts[0]
        return dict(lens=np.array(lens, np.uint64), skip=np.array(skip, np.uint8), gen=g,
                    name=("mixed artifact repo stand-in: %d xml (1-8 KiB) + %d jars (100 KiB-%.1f MiB, store path), %.2f GB"
                          % (len(xml), len(jars), max(jars) / 2**20, (nx + nj) / 1e9)) + ({"c5": "", "c5text": "; the xml files hold real (non-periodic) text"}.get(name, " (reduced; NOT BASELINE's size)")))
    if name == "c2store":  # BASELINE configs[1]'s shape with content the reference would not compress (skip list: png, jpg, gz ...)
        n = 100_000
        return dict(lens=np.full(n, 10240, np.uint64), skip=np.ones(n, np.uint8), gen=lambda torch, b0, b1: gen_gpu.random_lcg(b1 - b0, start=b0),
                    name="100k x 10KiB incompressible chunks, store path (a repo of small pre-compressed files; NOT a BASELINE config)")
    if name in ("c4store", "c4codec"):
        size = 500 << 20
        lens = np.array([SLICE] * (size // SLICE) + ([size % SLICE] if size % SLICE else []), dtype=np.uint64)
        skip = np.ones(len(lens), np.uint8) if name == "c4store" else None
        return dict(lens=lens, skip=skip, gen=lambda torch, b0, b1: gen_gpu.random_lcg(b1 - b0, start=b0),
                    name="500 MiB LCG blob, 8 MiB slices, " + ("store path (random.jar)" if skip is not None else "codec path (random.bin)"))
    raise SystemExit(f"unknown workload {name}")


def build(name, torch, byte_range=None):
    """-> dict(d_src, lens, skip, name, base): the Rounds of one configuration over a resident staging buffer;
    byte_range = (b0, b1): only those bytes of it are built (d_src[0] is byte b0 = base)."""
    L = layout(name)
    total = int(L["lens"].sum())
    b0, b1 = byte_range if byte_range is not None else (0, total)
    return dict(d_src=L["gen"](torch, int(b0), int(b1)), lens=L["lens"], skip=L["skip"], name=L["name"], base=int(b0))


def libzstd_compress(data, level=19) -> bytes:
    """One zstd frame made by the system's libzstd (ctypes, no oracle involved): the stand-in for what the
    reference's level-19 codec writes into an archive (common_config.rs:L37).  Input generation only."""
    import ctypes as C
    z = C.CDLL("libzstd.so.1")
    z.ZSTD_compressBound.restype = C.c_size_t
    z.ZSTD_compressBound.argtypes = [C.c_size_t]
    z.ZSTD_compress.restype = C.c_size_t
    z.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    z.ZSTD_isError.restype = C.c_uint
    z.ZSTD_isError.argtypes = [C.c_size_t]
    raw = bytes(data)
    cap = z.ZSTD_compressBound(len(raw))
    out = C.create_string_buffer(cap)
    r = z.ZSTD_compress(out, cap, raw, len(raw), level)
    if z.ZSTD_isError(r):
        raise RuntimeError("ZSTD_compress failed")
    return out.raw[:r]


def libzstd_decompress(frame, cap) -> bytes:
    """The system's libzstd decoding one frame (ctypes, no oracle involved): an outside judge of the encoder's frames."""
    import ctypes as C
    z = C.CDLL("libzstd.so.1")
    z.ZSTD_decompress.restype = C.c_size_t
    z.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    z.ZSTD_isError.restype = C.c_uint
    z.ZSTD_isError.argtypes = [C.c_size_t]
    out = C.create_string_buffer(max(int(cap), 1))
    raw = bytes(frame)
    r = z.ZSTD_decompress(out, max(int(cap), 1), raw, len(raw))
    if z.ZSTD_isError(r):
        raise RuntimeError("ZSTD_decompress failed")
    return out.raw[:r]


def libzstd_compress_adv(data, level=19, checksum=False, window_log=0) -> bytes:
    """libzstd's advanced API (ZSTD_compress2): a frame with a content checksum trailer and/or a given window log."""
    import ctypes as C
    z = C.CDLL("libzstd.so.1")
    z.ZSTD_createCCtx.restype = C.c_void_p
    z.ZSTD_freeCCtx.argtypes = [C.c_void_p]
    z.ZSTD_CCtx_setParameter.restype = C.c_size_t
    z.ZSTD_CCtx_setParameter.argtypes = [C.c_void_p, C.c_int, C.c_int]
    z.ZSTD_compressBound.restype = C.c_size_t
    z.ZSTD_compressBound.argtypes = [C.c_size_t]
    z.ZSTD_compress2.restype = C.c_size_t
    z.ZSTD_compress2.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    z.ZSTD_isError.restype = C.c_uint
    z.ZSTD_isError.argtypes = [C.c_size_t]
    cctx = z.ZSTD_createCCtx()
    try:
        for prm, val in ((100, level), (201, 1 if checksum else 0)) + (((101, window_log),) if window_log else ()):
            if z.ZSTD_isError(z.ZSTD_CCtx_setParameter(cctx, prm, val)):
                raise RuntimeError(f"ZSTD_CCtx_setParameter({prm}, {val}) failed")
        raw = bytes(data)
        cap = z.ZSTD_compressBound(len(raw))
        out = C.create_string_buffer(cap)
        r = z.ZSTD_compress2(cctx, out, cap, raw, len(raw))
        if z.ZSTD_isError(r):
            raise RuntimeError("ZSTD_compress2 failed")
        return out.raw[:r]
    finally:
        z.ZSTD_freeCCtx(cctx)


def image_corpus(kind, cap, whole_files=True):
    """Real files found in the image (not a BASELINE config): kind "text" = Python / C++ sources, "binary" = shared
    objects; cut into rounds of at most 8 MiB, up to cap bytes (whole_files: the file that crosses the cap is taken to its end —
    the corpus of the real-data reports — otherwise the cut is at the round that crosses it)."""
    import glob, os
    pats = {"text": ["/usr/lib/python3.10/**/*.py", "/usr/lib/python3/dist-packages/**/*.py", "/opt/rocm/include/**/*.h*"],
            "binary": ["/opt/rocm/lib/*.so*", "/usr/lib/x86_64-linux-gnu/*.so*"]}[kind]
    out, tot = [], 0
    for pat in pats:
        for f in sorted(glob.glob(pat, recursive=True)):
            if os.path.islink(f) or not os.path.isfile(f):
                continue
            try:
                with open(f, "rb") as fh:  # (not more of a file than the cap can take: the first shared object is 1 GB)
                    b = fh.read() if whole_files else fh.read(int(cap - tot) + (8 << 20))
            except OSError:
                continue
            if not b:
                continue
            for o in range(0, len(b), 8 << 20):
                out.append(b[o:o + (8 << 20)])
                tot += len(out[-1])
                if tot >= cap and not whole_files:
                    return out
            if tot >= cap:
                return out
    return out
