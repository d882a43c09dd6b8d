// Fused small-row read path for CDNA4 (gfx950): one wavefront takes one Tile (several whole
// index rows, <= 64 BLAKE3 leaves in total), decodes every compressed row of the tile and then
// hashes all the tile's leaves at once (lane = leaf) while the decoded bytes are still in the
// XCD's L2 — the body of the reference's read worker loop (znippy-common/src/decompress.rs:L135-190)
// for up to 64 KiB of output per wave, with no intermediate pass over HBM.
//
// Rows of the common shape — one compressed block = raw literals + ONE sequence whose match repeats a period
// lying inside those literals — are recognised lane-parallel (lane u parses row u, parse_fast), never expanded
// anywhere: the BLAKE3 lanes read their 64-byte blocks straight from the staged frame window in LDS (the
// literals + 64 bytes of the period) and the same message registers are stored to the output, so such a row
// costs one read of its frame, one write of its bytes and the hash.
//
// Every other "simple" frame (raw/RLE blocks, compressed blocks with raw/RLE literals and predefined/RLE
// sequence tables, up to 16 sequences) is decoded by the wave row after row with SCALAR parsing
// (decode_simple: the frame lives in one VGPR, lane = dword of a 256-byte window, every header field /
// bitstream read is a v_readlane + SALU shift); overlapping matches are expanded once into an LDS pattern
// buffer and streamed out 1 KiB per wave-instruction.  Anything else (Huffman literals, FSE table
// descriptions) is handed to the general decoder through the pending list and hashed by the second pass.
//
// The parent trees of a workgroup's four tiles are folded together by one of its waves (BlockFold).
#include "common.h"
#include "hash_dev.h"

#include <algorithm>
#include <cstring>

namespace zn {

constexpr uint32_t EOFF_MAX = 960;              // longest period expanded in LDS
constexpr uint32_t EBUF = EOFF_MAX + 1024 + 64; // pattern buffer per wave (= 2 KiB: it shares the wave's slice of the fold nodes)
constexpr int F_E_CORRUPT = -5, F_E_UNSUP = -6, F_E_DST = -4;
constexpr int F_NOT_SIMPLE = 1;

struct DTab {
    uint16_t next;
    uint8_t nbits, addbits;
    uint32_t base;
};
__constant__ DTab c_dll[64], c_dml[64], c_dof[32];
__constant__ uint32_t c_llb[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18,
                                   20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512, 1024, 2048,
                                   4096, 8192, 16384, 32768, 65536};
__constant__ uint8_t c_lla[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1,
                                  1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
__constant__ uint32_t c_mlb[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20,
                                   21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 37,
                                   39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051,
                                   4099, 8195, 16387, 32771, 65539};
__constant__ uint8_t c_mla[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                  0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1,
                                  1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11,
                                  12, 13, 14, 15, 16};

// ---- host: decoding tables of the predefined distributions -> __constant__ -----------------------
static void host_fse_build(const int8_t *norm, int nsym, int log, int kind, DTab *t) {
    static const uint32_t llb[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18,
                                     20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512, 1024, 2048,
                                     4096, 8192, 16384, 32768, 65536};
    static const uint8_t lla[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1,
                                    1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
    static const uint32_t mlb[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20,
                                     21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 37,
                                     39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051,
                                     4099, 8195, 16387, 32771, 65539};
    static const uint8_t mla[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1,
                                    1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11,
                                    12, 13, 14, 15, 16};
    const int size = 1 << log;
    uint8_t sym[512];
    uint16_t next[64];
    int high = size;
    for (int s = 0; s < nsym; s++)
        if (norm[s] == -1) { sym[--high] = (uint8_t)s; next[s] = 1; }
    const int step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
    int pos = 0;
    for (int s = 0; s < nsym; s++) {
        if (norm[s] <= 0) continue;
        next[s] = (uint16_t)norm[s];
        for (int i = 0; i < norm[s]; i++) {
            sym[pos] = (uint8_t)s;
            do { pos = (pos + step) & mask; } while (pos >= high);
        }
    }
    for (int i = 0; i < size; i++) {
        uint32_t s = sym[i], ns = next[s]++;
        int nb = log - (31 - __builtin_clz(ns));
        t[i].next = (uint16_t)((ns << nb) - size);
        t[i].nbits = (uint8_t)nb;
        if (kind == 0) { t[i].base = llb[s]; t[i].addbits = lla[s]; }
        else if (kind == 2) { t[i].base = mlb[s]; t[i].addbits = mla[s]; }
        else { t[i].base = 1u << s; t[i].addbits = (uint8_t)s; }
    }
}

void init_fused_tables() {
    static const int8_t ll[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2,
                                  2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
    static const int8_t ml[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                  1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                  1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
    static const int8_t of[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1,
                                  1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
    DTab tl[64], tm[64], to[32];
    host_fse_build(ll, 36, 6, 0, tl);
    host_fse_build(ml, 53, 6, 2, tm);
    host_fse_build(of, 29, 5, 1, to);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(c_dll), tl, sizeof tl);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(c_dml), tm, sizeof tm);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(c_dof), to, sizeof to);
}

__device__ int g_abl = 0;  // diagnostic ablations (ZNIPPY_DBG bits 16/32/64), never set in normal runs
// diagnostic stamp accumulators (ZNIPPY_DBG & 8 only): per-wave LDS slots, flushed once at the end
#define STAMP_BEGIN() unsigned long long st__ = acc ? __builtin_amdgcn_s_memtime() : 0
#define STAMP_BEGIN2() st__ = acc ? __builtin_amdgcn_s_memtime() : 0
#define STAMP_END(slot)                                                        \
    do {                                                                       \
        if (acc) {                                                             \
            unsigned long long e__ = __builtin_amdgcn_s_memtime();             \
            if ((threadIdx.x & 63) == 0) acc[slot] += e__ - st__;              \
            st__ = e__;                                                        \
        }                                                                      \
    } while (0)

// ---- device helpers ---------------------------------------------------------------------------------
constexpr uint32_t WIN = 512;       // bytes of every row's frame staged in LDS by the tile prologue
constexpr uint32_t WROWS = 6;       // rows per tile that get a staged window (others load on demand)
constexpr uint32_t WSTRIDE = WIN + 96;  // + slack: 16-byte over-reads and the 64 period bytes appended for hashing

__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int fhib(uint32_t v) { return 31 - __clz(v); }
__device__ __forceinline__ void fwave_mem_sync() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// scalar a % b without the VALU reciprocal sequence (operands are wave-uniform -> SALU loop)
__device__ __forceinline__ uint32_t smod(uint32_t a, uint32_t b) {
    if (a < b) return a;
    uint32_t sh = (uint32_t)(__clz(b) - __clz(a));
    uint32_t d = b << sh;
    for (;;) {
        if (a >= d) a -= d;
        if (sh == 0) break;
        d >>= 1;
        sh--;
    }
    return a;
}

// x mod d for per-lane x < 2^22 and wave-uniform d >= 1 (inv = 1/d): float quotient, off by at most one
__device__ __forceinline__ uint32_t lmod(uint32_t x, uint32_t d, float inv) {
    const uint32_t q = (uint32_t)((float)x * inv);
    int32_t r = (int32_t)(x - q * d);
    if (r < 0) r += (int32_t)d;
    else if ((uint32_t)r >= d) r -= (int32_t)d;
    return (uint32_t)r;
}

// The frame as the wave sees it: bytes [0, min(n,WIN)) staged in LDS (`wl`, zero padded) and, for
// scalar parsing, a 256-byte register window (lane l = dword l of [wbase, wbase+256)).
struct FrameWin {
    const uint8_t *src;  // global
    uint8_t *wl;         // LDS copy of the first WIN bytes, or nullptr
    uint32_t n, wbase, w;
    __device__ __forceinline__ void load(uint32_t base) {
        const uint32_t lane = threadIdx.x & 63;
        wbase = base;
        const uint32_t o = base + 4 * lane;
        uint32_t v = 0;
        if (o + 4 <= n) __builtin_memcpy(&v, src + o, 4);
        else
            for (uint32_t k = 0; k < 4; k++)
                if (o + k < n) v |= (uint32_t)src[o + k] << (8 * k);
        w = v;
    }
    // 8 bytes at byte position pos (wave-uniform)
    __device__ __forceinline__ uint64_t u64(uint32_t pos) {
        const uint32_t rel = pos - wbase, idx = rel >> 2, sh = (rel & 3) * 8;
        const uint32_t d0 = __builtin_amdgcn_readlane(w, idx), d1 = __builtin_amdgcn_readlane(w, idx + 1),
                       d2 = __builtin_amdgcn_readlane(w, idx + 2);
        uint64_t lo = (uint64_t)d0 | ((uint64_t)d1 << 32);
        return sh ? (lo >> sh) | ((uint64_t)d2 << (64 - sh)) : lo;
    }
    __device__ __forceinline__ uint64_t fwd(uint32_t pos) {
        if (pos < wbase || pos + 12 > wbase + 256) load(pos);
        return u64(pos);
    }
    __device__ __forceinline__ uint64_t bwd(uint32_t pos) {
        if (pos < wbase || pos + 12 > wbase + 256) load(pos >= 240 ? pos - 240 : 0);
        return u64(pos);
    }
    // LDS pointer to frame bytes [at, at+len) if they are inside the staged window
    __device__ __forceinline__ const uint8_t *lds(uint32_t at, uint32_t len) const {
        return (wl && at + len <= WIN) ? wl + at : nullptr;
    }
};

// backward bit reader over frame bytes [s_begin, ...): bp = unread bits.  A 64-bit scalar cache
// holds stream bits [cb, cb+64); it is refilled (3 readlanes) only when a read leaves it.
struct SBits {
    uint32_t s_begin;
    int32_t bp;
    uint64_t cw;
    int32_t cb;  // bit index of cw's bit 0; INT32_MIN = empty
    __device__ __forceinline__ uint32_t read(FrameWin &W, uint32_t nb) {  // nb <= 32, scalar
        if (nb == 0) return 0;
        uint32_t v = 0;
        if (bp > 0) {
            const uint32_t take = (uint32_t)bp < nb ? (uint32_t)bp : nb;  // real bits available
            const int32_t lo_bit = bp - (int32_t)take;
            if (lo_bit < cb || bp > cb + 64) {
                // refill so that the window ends at the byte holding bit bp-1
                int32_t top_byte = (bp + 7) >> 3;
                int32_t b0 = top_byte > 8 ? top_byte - 8 : 0;
                cw = W.bwd(s_begin + (uint32_t)b0);
                cb = b0 * 8;
            }
            const uint64_t x = cw >> (uint32_t)(lo_bit - cb);
            v = (uint32_t)(x & ((1ull << take) - 1)) << (nb - take);
        }
        bp -= (int32_t)nb;
        return v;
    }
};

// ---- wave-cooperative byte movers (64 lanes) ----------------------------------------------------
// global -> global
__device__ __forceinline__ void fwave_copy(uint8_t *dst, const uint8_t *src, uint32_t n, uint32_t lane) {
    if (n < 128) {
        for (uint32_t i = lane; i < n; i += 64) dst[i] = src[i];
        return;
    }
    const uint32_t head = (uint32_t)((16 - ((uintptr_t)dst & 15)) & 15);
    if (lane < head) dst[lane] = src[lane];
    const uint32_t body = (n - head) >> 4;
    for (uint32_t i = lane; i < body; i += 64) {
        uint4 v = ld16(src + head + (size_t)i * 16);
        *reinterpret_cast<uint4 *>(dst + head + (size_t)i * 16) = v;
    }
    const uint32_t done = head + body * 16;
    if (done + lane < n) dst[done + lane] = src[done + lane];
}

// LDS -> global, n <= WIN
__device__ __forceinline__ void fwave_copy_l2g(uint8_t *dst, const uint8_t *l, uint32_t n, uint32_t lane) {
    for (uint32_t i = lane; i < n; i += 64) dst[i] = l[i];
}

__device__ __forceinline__ void fwave_fill(uint8_t *dst, uint8_t byte, uint32_t n, uint32_t lane) {
    if (n < 128) {
        for (uint32_t i = lane; i < n; i += 64) dst[i] = byte;
        return;
    }
    const uint32_t head = (uint32_t)((16 - ((uintptr_t)dst & 15)) & 15);
    if (lane < head) dst[lane] = byte;
    const uint32_t body = (n - head) >> 4;
    const uint32_t w = byte * 0x01010101u;
    const uint4 v = make_uint4(w, w, w, w);
    for (uint32_t i = lane; i < body; i += 64) *reinterpret_cast<uint4 *>(dst + head + (size_t)i * 16) = v;
    const uint32_t done = head + body * 16;
    if (done + lane < n) dst[done + lane] = byte;
}

// dst[i] = pat[i % off], i < ml, for an OVERLAPPING match (off < ml) with off <= EOFF_MAX.
// The period comes from LDS (`pl`, staged frame literals) or from final global bytes (`pg`).
// E = this wave's LDS pattern buffer: E[i] = pat[i % off] for i < off + min(ml, 1024) is built by
// doubling inside LDS (ds ops of one wave are ordered), then streamed out 1 KiB per step with
// 16-byte aligned stores; head and tail bytes go out lane-parallel.
__device__ __forceinline__ void fwave_expand(uint8_t *dst, const uint8_t *pg, const uint8_t *pl, uint32_t off,
                                             uint32_t ml, uint8_t *E, uint32_t lane, unsigned long long *acc) {
    STAMP_BEGIN();
    // 1) the period itself into E (16-byte pieces may spill up to 15 bytes; rewritten below)
    if (pl) {
        for (uint32_t i = lane * 16; i < off; i += 1024) {
            uint4 v;
            __builtin_memcpy(&v, pl + i, 16);  // staged window has 32 bytes of readable padding
            __builtin_memcpy(E + i, &v, 16);
        }
    } else {
        const uint32_t full = off & ~15u;
        for (uint32_t i = lane * 16; i < full; i += 1024) {
            uint4 v = ld16(pg + i);
            __builtin_memcpy(E + i, &v, 16);
        }
        if (lane < (off & 15)) E[full + lane] = pg[full + lane];
    }
    // 2) the rest of E: first 64 more bytes of the period one byte per lane, then every further 16-byte piece is read at
    // its phase inside [0, off + 16) and stored aligned; the pieces do not depend on one another, so they go 1 KiB per
    // round (the first version doubled the filled part step by step: a chain of dependent LDS round trips).
    const uint32_t need = off + (ml < 1024 ? ml : 1024);
    const float inv = 1.0f / (float)off;
    E[off + lane] = E[lmod(lane, off, inv)];
    for (uint32_t c = ((off + 64 + 15) >> 4) + lane; 16 * c < need; c += 64) {
        uint4 v;
        __builtin_memcpy(&v, E + lmod(16 * c, off, inv), 16);
        *reinterpret_cast<uint4 *>(E + 16 * c) = v;
    }
    // (bytes [off + 64, 16 * ceil((off + 64) / 16)) belong to no piece: fill them bytewise)
    {
        const uint32_t lo = off + 64, hi = ((off + 64 + 15) >> 4) << 4;
        if (lo + lane < hi && lo + lane < need) E[lo + lane] = E[lmod(lo + lane, off, inv)];
    }
    STAMP_END(4);
    if (g_abl & 16) return;  // ablation: no stream-out
    // 3) stream out
    const uint32_t head = (uint32_t)((16 - ((uintptr_t)dst & 15)) & 15);
    if (ml < head + 16) {  // too short for an aligned piece
        for (uint32_t i = lane; i < ml; i += 64) dst[i] = E[i];
        return;
    }
    if (lane < head) dst[lane] = E[lane];
    uint32_t x = head, s = smod(head, off);
    const uint32_t step = smod(1024, off);
    uint8_t *d = dst + head + 16 * lane;
    while (x + 4096 <= ml) {  // 4 KiB per trip: four independent LDS reads in flight before the stores
        uint32_t s1 = s + step; if (s1 >= off) s1 -= off;
        uint32_t s2 = s1 + step; if (s2 >= off) s2 -= off;
        uint32_t s3 = s2 + step; if (s3 >= off) s3 -= off;
        uint4 v0, v1, v2, v3;
        __builtin_memcpy(&v0, E + s + 16 * lane, 16);
        __builtin_memcpy(&v1, E + s1 + 16 * lane, 16);
        __builtin_memcpy(&v2, E + s2 + 16 * lane, 16);
        __builtin_memcpy(&v3, E + s3 + 16 * lane, 16);
        *reinterpret_cast<uint4 *>(d) = v0;
        *reinterpret_cast<uint4 *>(d + 1024) = v1;
        *reinterpret_cast<uint4 *>(d + 2048) = v2;
        *reinterpret_cast<uint4 *>(d + 3072) = v3;
        d += 4096;
        x += 4096;
        s = s3 + step;
        if (s >= off) s -= off;
    }
    while (x + 1024 <= ml) {
        uint4 v;
        __builtin_memcpy(&v, E + s + 16 * lane, 16);
        *reinterpret_cast<uint4 *>(d) = v;
        d += 1024;
        x += 1024;
        s += step;
        if (s >= off) s -= off;
    }
    const uint32_t rem = ml - x;             // < 1024
    const uint32_t full16 = rem >> 4;        // whole 16-byte pieces left
    if (lane < full16) {
        uint4 v;
        __builtin_memcpy(&v, E + s + 16 * lane, 16);
        *reinterpret_cast<uint4 *>(d) = v;
    }
    const uint32_t tail = rem & 15, tbase = full16 * 16;
    if (lane < tail) dst[x + tbase + lane] = E[s + tbase + lane];
    STAMP_END(5);
}

// LZ match of the fused path: dst[i] = dst[i - off].  `pl`/`pg_fwd`: the period forwarded from
// the frame's literals when it lies inside the literal run just copied (no read-after-write
// through memory); otherwise the period is re-read from the output after a store drain.
__device__ __forceinline__ void fwave_match(uint8_t *dst, const uint8_t *pg_fwd, const uint8_t *pl, uint32_t off,
                                            uint32_t ml, uint8_t *E, uint32_t lane, unsigned long long *acc) {
    const uint8_t *pg = pg_fwd;
    if (!pg && !pl) {
        fwave_mem_sync();  // earlier output of this wave must have landed before it is re-read
        pg = dst - off;
    }
    if (off >= ml) {  // no overlap: plain copy
        if (pl) fwave_copy_l2g(dst, pl, ml, lane);
        else fwave_copy(dst, pg, ml, lane);
        return;
    }
    if (off <= EOFF_MAX) {
        if (!(g_abl & 32)) fwave_expand(dst, pg, pl, off, ml, E, lane, acc);
        return;
    }
    // long period + overlap (rare in small rows): byte-parallel modulo copy from the final period
    if (pl) { for (uint32_t i = lane; i < ml; i += 64) dst[i] = pl[i % off]; }
    else { for (uint32_t i = lane; i < ml; i += 64) dst[i] = pg[i % off]; }
}

// Decode one simple frame with the calling wave.  Returns 0, F_NOT_SIMPLE or a negative error.
struct Periodic {  // row = literal prefix + ONE overlapping match to the end (hashable from LDS)
    uint32_t ok, lit_at, L0, off;
};

__device__ int decode_simple(const uint8_t *src, uint32_t n, uint8_t *out, uint64_t usize, uint8_t *E,
                             uint8_t *wl, Periodic *per, unsigned long long *acc) {
    per->ok = 0;
    STAMP_BEGIN();
    const uint32_t lane = threadIdx.x & 63;
    FrameWin W;
    W.src = src; W.n = n; W.wl = wl;
    if (wl) {  // staged by the tile prologue: the register window comes from LDS, not from HBM
        W.wbase = 0;
        W.w = *reinterpret_cast<const uint32_t *>(wl + 4 * lane);
    } else {
        W.load(0);
    }
    if (n < 5) return F_E_CORRUPT;
    uint64_t h = W.fwd(0);
    if ((uint32_t)h != 0xFD2FB528u) return F_E_CORRUPT;
    const uint32_t fhd = (uint32_t)(h >> 32) & 0xFF;
    const uint32_t fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, did_flag = fhd & 3;
    if (fhd & 8) return F_E_CORRUPT;
    const uint32_t has_ck = (fhd >> 2) & 1;
    if (has_ck) return F_NOT_SIMPLE;  // content checksum (XXH64): verified by the general decoder
    const uint32_t fcs_bytes = fcs_flag == 0 ? single : (1u << fcs_flag);
    const uint32_t did_bytes = did_flag == 3 ? 4 : did_flag;
    uint32_t pos = 5 + (single ? 0 : 1);
    if (n < pos + did_bytes + fcs_bytes) return F_E_CORRUPT;
    if (did_bytes) {
        uint32_t did = (uint32_t)(W.fwd(pos) & (did_bytes == 4 ? 0xFFFFFFFFull : ((1ull << (8 * did_bytes)) - 1)));
        if (did) return F_E_UNSUP;
        pos += did_bytes;
    }
    if (!fcs_bytes) return F_E_UNSUP;
    uint64_t fcs = W.fwd(pos);
    if (fcs_bytes < 8) fcs &= (1ull << (8 * fcs_bytes)) - 1;
    if (fcs_bytes == 2) fcs += 256;
    pos += fcs_bytes;
    if (fcs != usize) return F_E_CORRUPT;

    uint32_t opos = 0;
    uint32_t r0 = 1, r1 = 4, r2 = 8;
    const uint32_t osize = (uint32_t)usize;  // small rows only (<= 64 KiB)
    for (;;) {
        if (pos + 3 > n) return F_E_CORRUPT;
        const uint32_t bh = (uint32_t)W.fwd(pos) & 0xFFFFFF;
        pos += 3;
        const uint32_t last = bh & 1, btype = (bh >> 1) & 3, bsize = bh >> 3;
        if (btype == 3) return F_E_CORRUPT;
        if (btype == 0) {
            if (pos + bsize > n || opos + bsize > osize) return F_E_CORRUPT;
            if (const uint8_t *l = W.lds(pos, bsize)) fwave_copy_l2g(out + opos, l, bsize, lane);
            else fwave_copy(out + opos, src + pos, bsize, lane);
            opos += bsize; pos += bsize;
        } else if (btype == 1) {
            if (pos + 1 > n || opos + bsize > osize) return F_E_CORRUPT;
            fwave_fill(out + opos, (uint8_t)(W.fwd(pos) & 0xFF), bsize, lane);
            opos += bsize; pos += 1;
        } else {
            if (pos + bsize > n || bsize > 128 * 1024 || bsize < 2) return F_E_CORRUPT;
            const uint32_t bend = pos + bsize;
            // literals section: raw or RLE only
            const uint32_t lh = (uint32_t)W.fwd(pos);
            const uint32_t ltype = lh & 3, sf = (lh >> 2) & 3;
            if (ltype >= 2) return F_NOT_SIMPLE;
            uint32_t regen, lhdr;
            if ((sf & 1) == 0) { regen = (lh & 0xFF) >> 3; lhdr = 1; }
            else if (sf == 1) { regen = (lh & 0xFFFF) >> 4; lhdr = 2; }
            else { regen = (lh & 0xFFFFFF) >> 4; lhdr = 3; }
            uint32_t lit_at;   // frame position of raw literals
            uint32_t rle_byte = 0;
            pos += lhdr;
            if (ltype == 0) {
                if (pos + regen > bend) return F_E_CORRUPT;
                lit_at = pos;
                pos += regen;
            } else {
                if (pos + 1 > bend) return F_E_CORRUPT;
                rle_byte = (uint32_t)W.fwd(pos) & 0xFF;
                lit_at = 0;
                pos += 1;
            }
            // sequences header
            if (pos >= bend) return F_E_CORRUPT;
            const uint32_t sh = (uint32_t)W.fwd(pos);
            uint32_t nseq;
            const uint32_t b0 = sh & 0xFF;
            if (b0 == 0) { nseq = 0; pos += 1; }
            else if (b0 < 128) { nseq = b0; pos += 1; }
            else if (b0 < 255) { nseq = ((b0 - 128) << 8) + ((sh >> 8) & 0xFF); pos += 2; }
            else { nseq = ((sh >> 8) & 0xFFFF) + 0x7F00; pos += 3; }
            uint32_t lit_pos = 0;
            // This path executes sequences one at a time, each match straight into HBM: right for the one or two
            // long matches of periodic rows, wrong for real text.  Rows with many sequences go to the general
            // decoder, which executes 64 of them per step inside an LDS window.
            if (nseq > 16 && opos == 0) return F_NOT_SIMPLE;
            if (nseq) {
                if (pos >= bend) return F_E_CORRUPT;
                const uint32_t modes = (uint32_t)W.fwd(pos) & 0xFF;
                pos += 1;
                if (modes & 3) return F_E_CORRUPT;
                const uint32_t m_ll = (modes >> 6) & 3, m_of = (modes >> 4) & 3, m_ml = (modes >> 2) & 3;
                if (m_ll > 1 || m_of > 1 || m_ml > 1) return F_NOT_SIMPLE;
                uint32_t rl_base = 0, rl_add = 0, ro_base = 0, ro_add = 0, rm_base = 0, rm_add = 0;
                if (m_ll) {
                    if (pos >= bend) return F_E_CORRUPT;
                    uint32_t s = (uint32_t)W.fwd(pos) & 0xFF; pos++;
                    if (s > 35) return F_E_CORRUPT;
                    rl_base = c_llb[s]; rl_add = c_lla[s];
                }
                if (m_of) {
                    if (pos >= bend) return F_E_CORRUPT;
                    uint32_t s = (uint32_t)W.fwd(pos) & 0xFF; pos++;
                    if (s > 31) return F_E_CORRUPT;
                    ro_base = 1u << s; ro_add = s;
                }
                if (m_ml) {
                    if (pos >= bend) return F_E_CORRUPT;
                    uint32_t s = (uint32_t)W.fwd(pos) & 0xFF; pos++;
                    if (s > 52) return F_E_CORRUPT;
                    rm_base = c_mlb[s]; rm_add = c_mla[s];
                }
                if (pos >= bend) return F_E_CORRUPT;
                const uint32_t lastb = (uint32_t)W.bwd(bend - 1) & 0xFF;
                if (!lastb) return F_E_CORRUPT;
                SBits B;
                B.s_begin = pos;
                B.bp = (int32_t)((bend - pos) * 8 - (8 - fhib(lastb)));
                B.cw = 0;
                B.cb = INT32_MIN / 2;
                uint32_t sl = m_ll ? 0 : B.read(W, 6);
                uint32_t so = m_of ? 0 : B.read(W, 5);
                uint32_t sm = m_ml ? 0 : B.read(W, 6);
                for (uint32_t i = 0; i < nseq; i++) {
                    uint32_t ob, oa, onx = 0, onb = 0, mb, ma, mnx = 0, mnb = 0, lb, la, lnx = 0, lnb = 0;
                    if (m_of) { ob = ro_base; oa = ro_add; } else { const DTab e = c_dof[so]; ob = e.base; oa = e.addbits; onx = e.next; onb = e.nbits; }
                    if (m_ml) { mb = rm_base; ma = rm_add; } else { const DTab e = c_dml[sm]; mb = e.base; ma = e.addbits; mnx = e.next; mnb = e.nbits; }
                    if (m_ll) { lb = rl_base; la = rl_add; } else { const DTab e = c_dll[sl]; lb = e.base; la = e.addbits; lnx = e.next; lnb = e.nbits; }
                    const uint32_t ov = ob + B.read(W, oa);
                    const uint32_t ml = mb + B.read(W, ma);
                    const uint32_t ll = lb + B.read(W, la);
                    if (i + 1 < nseq) {
                        sl = lnx + B.read(W, lnb);
                        sm = mnx + B.read(W, mnb);
                        so = onx + B.read(W, onb);
                    }
                    if (B.bp < 0) return F_E_CORRUPT;
                    uint32_t offset;
                    if (ov > 3) { offset = ov - 3; r2 = r1; r1 = r0; r0 = offset; }
                    else {
                        const uint32_t idx = ov - 1 + (ll == 0 ? 1 : 0);
                        if (idx == 0) offset = r0;
                        else {
                            offset = idx == 1 ? r1 : (idx == 2 ? r2 : r0 - 1);
                            if (offset == 0) return F_E_CORRUPT;
                            if (idx > 1) r2 = r1;
                            r1 = r0; r0 = offset;
                        }
                    }
                    if (lit_pos + ll > regen) return F_E_CORRUPT;
                    if ((uint64_t)opos + ll + ml > osize) return F_E_CORRUPT;
                    // literals
                    if (ll) {
                        if (ltype == 0) {
                            if (const uint8_t *l = W.lds(lit_at + lit_pos, ll)) fwave_copy_l2g(out + opos, l, ll, lane);
                            else fwave_copy(out + opos, src + lit_at + lit_pos, ll, lane);
                        } else fwave_fill(out + opos, (uint8_t)rle_byte, ll, lane);
                    }
                    opos += ll; lit_pos += ll;
                    if (offset > opos) return F_E_CORRUPT;
                    // match: the period is forwarded from the frame when it lies inside this literal run
                    if (offset <= ll) {
                        if (ltype == 0) {
                            const uint32_t pat_at = lit_at + lit_pos - offset;
                            const uint8_t *pl = W.lds(pat_at, offset);
                            STAMP_END(6);
                            fwave_match(out + opos, src + pat_at, pl, offset, ml, E, lane, acc);
                            STAMP_BEGIN2();
                            // whole row = these literals + this one periodic match: keep out[0..L0+64) in LDS
                            // (the literals are already staged; 64 more bytes of the period come from E)
                            if (pl && nseq == 1 && last && opos == ll && ll == regen && (uint64_t)ll + ml == osize &&
                                offset < ml && offset <= EOFF_MAX && ml >= 64 && lit_at + ll + 64 <= WSTRIDE) {
                                W.wl[lit_at + ll + lane] = E[lane];
                                per->ok = 1; per->lit_at = lit_at; per->L0 = ll; per->off = offset;
                            }
                        } else fwave_fill(out + opos, (uint8_t)rle_byte, ml, lane);
                    } else {
                        fwave_match(out + opos, nullptr, nullptr, offset, ml, E, lane, acc);
                    }
                    opos += ml;
                }
                if (B.bp != 0) return F_E_CORRUPT;
            } else if (pos != bend) {
                return F_E_CORRUPT;
            }
            const uint32_t rest = regen - lit_pos;
            if ((uint64_t)opos + rest > osize) return F_E_CORRUPT;
            if (rest) {
                if (ltype == 0) {
                    if (const uint8_t *l = W.lds(lit_at + lit_pos, rest)) fwave_copy_l2g(out + opos, l, rest, lane);
                    else fwave_copy(out + opos, src + lit_at + lit_pos, rest, lane);
                } else fwave_fill(out + opos, (uint8_t)rle_byte, rest, lane);
            }
            opos += rest;
            pos = bend;
        }
        if (last) break;
    }
    STAMP_END(7);
    if (opos != osize) return F_E_CORRUPT;
    if (has_ck && pos + 4 > n) return F_E_CORRUPT;
    return 0;
}

// ---- lane-parallel recognition of the common small-row shape ------------------------------------------
// Most small rows of compressible data are ONE compressed block = raw literals + ONE sequence whose match
// repeats a period lying inside those literals (a 10 KiB text chunk: 45 literal bytes + a 10,195-byte
// match at offset 45).  Parsing such a frame is ~40 dependent steps; done as scalar code row after row
// it was the longest serial chain of the kernel.  Here lane u parses row u on its own (every staged row
// of the tile at once, straight-line code, no divergence), and a row that is not of this shape —
// or is malformed in any way — is left to decode_simple, which reports the precise status.
struct FastRow {
    uint32_t ok, lit_at, L0, off;
};
struct FastTabs {  // lane i holds entry i: state -> base | addbits << 24 (LL, ML), state -> code (OF), code -> base | addbits << 24
    uint32_t ll, ml, of, lls, mls;
};
__device__ __forceinline__ uint64_t lds8(const uint8_t *p) {
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
__device__ __forceinline__ uint4 lds16(const uint8_t *p) {
    uint4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}
// The block part: `pos` = position of the block header in the window, `n` = bytes of the frame that the window
// holds.  FRAME_END: the block must be the frame's last one and end exactly at n (a whole small frame); otherwise it
// may be any compressed block that ends inside the window (one block item of a big frame, zstd_decode.hip).
template <bool FRAME_END>
__device__ __forceinline__ FastRow parse_fast_block(const uint8_t *w, uint32_t pos, uint32_t n, uint64_t usize, bool ok, const FastTabs &T) {
    const uint64_t b = lds8(w + pos);  // block header (3 bytes) + literals header (<= 3)
    const uint32_t bh = (uint32_t)b & 0xFFFFFF;
    ok &= FRAME_END ? (bh & 7) == 5 : ((bh >> 1) & 3) == 2;  // compressed (and the last block)
    pos += 3;
    const uint32_t bend = pos + (bh >> 3);
    ok &= FRAME_END ? bend == n : bend <= n;
    const uint32_t lh = (uint32_t)(b >> 24);
    ok &= (lh & 3) == 0;  // raw literals
    const uint32_t sf = (lh >> 2) & 3;
    const uint32_t regen = (sf & 1) == 0 ? (lh & 0xFF) >> 3 : (sf == 1 ? (lh & 0xFFFF) >> 4 : (lh & 0xFFFFFF) >> 4);
    const uint32_t lit_at = pos + ((sf & 1) == 0 ? 1u : (sf == 1 ? 2u : 3u));  // Size_Format 00/10: 1 byte, 01: 2, 11: 3 (RFC 8878 3.1.1.3.1.1)
    ok &= regen <= WIN;
    uint32_t sp = lit_at + regen;  // sequences section
    ok &= sp + 3 <= bend;
    sp = ok ? sp : 0;
    const uint64_t sh = lds8(w + sp);
    ok &= (sh & 0xFF) == 1;  // one sequence
    const uint32_t modes = (uint32_t)(sh >> 8) & 0xFF;
    const uint32_t m_ll = modes >> 6, m_of = (modes >> 4) & 3, m_ml = (modes >> 2) & 3;
    ok &= (modes & 3) == 0 && m_ll <= 1 && m_of <= 1 && m_ml <= 1;  // predefined or RLE tables
    uint32_t q = 2;
    const uint32_t s_ll = (uint32_t)(sh >> 16) & 0xFF;
    q += m_ll & 1;
    const uint32_t s_of = (uint32_t)(sh >> (8 * q)) & 0xFF;
    q += m_of & 1;
    const uint32_t s_ml = (uint32_t)(sh >> (8 * q)) & 0xFF;
    q += m_ml & 1;
    ok &= (!m_ll || s_ll <= 35) && (!m_of || s_of <= 31) && (!m_ml || s_ml <= 52);
    const uint32_t bs = sp + q;  // the bitstream: at most 8 bytes, held in one register
    ok &= bs < bend && bend - bs <= 8 && bend >= 8;
    const uint32_t slen = ok ? bend - bs : 1;
    const uint64_t cw = lds8(w + (ok ? bend - 8 : 0));
    const uint32_t lastb = (uint32_t)(cw >> 56);
    ok &= lastb != 0;
    int32_t bp = (int32_t)(slen * 8) - (int32_t)(8 - fhib(lastb | 1));
    const int32_t base = 64 - 8 * (int32_t)slen;
    auto rd = [&](uint32_t nb) -> uint32_t {  // nb <= 31; running out of bits leaves bp < 0 (checked once, below)
        bp -= (int32_t)nb;
        const int32_t s = bp + base;
        return (uint32_t)((cw >> (s < 0 ? 0 : s)) & ((1ull << nb) - 1));
    };
    const uint32_t st_ll = m_ll ? 0 : rd(6);
    const uint32_t st_of = m_of ? 0 : rd(5);
    const uint32_t st_ml = m_ml ? 0 : rd(6);
    const uint32_t e_ll_s = __shfl(T.ll, st_ll & 63), e_ll_r = __shfl(T.lls, s_ll & 63);
    const uint32_t e_ml_s = __shfl(T.ml, st_ml & 63), e_ml_r = __shfl(T.mls, s_ml & 63);
    const uint32_t c_of_s = __shfl(T.of, st_of & 31);
    const uint32_t e_ll = m_ll ? e_ll_r : e_ll_s, e_ml = m_ml ? e_ml_r : e_ml_s, c_of = (m_of ? s_of : c_of_s) & 31;
    const uint32_t ov = (1u << c_of) + rd(c_of);
    const uint32_t ml = (e_ml & 0xFFFFFF) + rd(e_ml >> 24);
    const uint32_t ll = (e_ll & 0xFFFFFF) + rd(e_ll >> 24);
    ok &= bp == 0;
    // first sequence of the frame / of a self-contained block, ll > 0: repeat codes 1..3 mean the initial offsets
    // 1, 4, 8 (RFC 8878 3.1.1.5).  A block item that uses one is not self-contained: left to the block decoder,
    // which flags the frame.
    const uint32_t off = ov > 3 ? ov - 3 : (ov == 1 ? 1u : (ov == 2 ? 4u : 8u));
    ok &= FRAME_END || ov > 3;
    ok &= ll == regen && ll >= 1 && (uint64_t)ll + ml == usize && off <= ll && off < ml && off <= EOFF_MAX && ml >= 64 &&
          lit_at + ll + 64 <= WSTRIDE;
    FastRow r;
    r.ok = ok ? 1u : 0u; r.lit_at = lit_at; r.L0 = ll; r.off = off;
    return r;
}

// Lane-parallel: is this staged frame certainly NOT one of the simple shapes?  A well-formed frame header whose first block
// is a compressed block with Huffman-coded literals (type 2 or 3): the scalar parser would say F_NOT_SIMPLE after reading the
// same bytes, one row after the other.  (A table of real text is all such rows: they go to the batch path at once.)
__device__ __forceinline__ bool not_simple_fast(const uint8_t *w, uint32_t n, uint64_t usize, bool want) {
    bool ok = want && n >= 12 && usize <= 0xFFFFFFull;
    const uint64_t h0 = lds8(w);
    ok &= (uint32_t)h0 == 0xFD2FB528u;
    const uint32_t fhd = (uint32_t)(h0 >> 32) & 0xFF;
    ok &= (fhd & 0x0F) == 0;  // no reserved bit, no content checksum, no dictionary id
    const uint32_t single = (fhd >> 5) & 1, fcs_flag = fhd >> 6;
    const uint32_t fcs_bytes = fcs_flag == 0 ? single : (1u << fcs_flag);
    ok &= fcs_bytes != 0;
    uint32_t pos = 6 - single;
    uint64_t f = lds8(w + pos);
    if (fcs_bytes < 8) f &= (1ull << (8 * fcs_bytes)) - 1;
    if (fcs_bytes == 2) f += 256;
    ok &= f == usize;
    pos += fcs_bytes;  // <= 14
    const uint64_t b = lds8(w + pos);  // block header (3 bytes) + the literals header's first byte
    const uint32_t bh = (uint32_t)b & 0xFFFFFF;
    ok &= ((bh >> 1) & 3) == 2 && (bh >> 3) >= 2 && pos + 3 + (bh >> 3) <= n && (bh >> 3) <= 128 * 1024;
    ok &= (((uint32_t)(b >> 24)) & 3) >= 2;
    return ok;
}

__device__ __forceinline__ FastRow parse_fast(const uint8_t *w, uint32_t n, uint64_t usize, bool want, const FastTabs &T) {
    bool ok = want && n >= 12 && n <= WIN && usize >= 65 && usize <= 0xFFFFFFull;
    const uint64_t h0 = lds8(w);
    ok &= (uint32_t)h0 == 0xFD2FB528u;
    const uint32_t fhd = (uint32_t)(h0 >> 32) & 0xFF;
    ok &= (fhd & 0x0F) == 0;  // no reserved bit, no content checksum, no dictionary id
    const uint32_t single = (fhd >> 5) & 1, fcs_flag = fhd >> 6;
    const uint32_t fcs_bytes = fcs_flag == 0 ? single : (1u << fcs_flag);
    ok &= fcs_bytes != 0;
    uint32_t pos = 6 - single;
    uint64_t f = lds8(w + pos);
    if (fcs_bytes < 8) f &= (1ull << (8 * fcs_bytes)) - 1;
    if (fcs_bytes == 2) f += 256;
    ok &= f == usize;
    pos += fcs_bytes;  // <= 14
    return parse_fast_block<true>(w, pos, n, usize, ok, T);
}

// Writing a recognised row from its window: out[i] = Y[i] for i < L0, Y[B + (i - B) mod off] after that
// (B = L0 - off), with Y = the row's literals in the staged window, extended by 64 bytes of the period, so that
// any 16 output bytes are 16 CONTIGUOUS bytes of Y — at i itself (i + 16 <= L0 + 64) or inside the period copy at
// B + r.  The row goes out as aligned 16-byte stores read straight from the window, 1 KiB per wave-instruction,
// with no expansion buffer and no dependence on bytes already written.  Used for tiles with a ragged leaf (their
// rows are hashed from the output); in a tile of whole leaves the hash lanes write the rows themselves
// (LdsSrc::store_mask) and only prepare() runs.
struct Emitter {
    uint8_t *WL;       // this wave's staged windows
    uint8_t *outbase;  // output region
    uint32_t lane;
    // lane u = row u of the tile
    uint32_t f_lit, f_L0, f_off, c_len_lo;
    uint64_t c_oo;
    // schedule (wave-uniform)
    uint32_t todo;   // bit u: recognised row u not opened yet
    // the open row (wave-uniform) and this lane's position in it
    uint32_t open_, i0, body16, off, step, B, lim;
    uint8_t *Y, *out;
    uint32_t x, r;  // per lane: byte position of piece i0 + lane, (x - B) mod off

    // extend every recognised row's window by 64 period bytes (the hash and the trips both rely on it)
    __device__ __forceinline__ void prepare(uint32_t mask) {
        todo = mask; open_ = 0;
        for (uint32_t m = mask; m; m &= m - 1) {
            const uint32_t u = (uint32_t)__builtin_ctz(m);
            const uint32_t lit_at = __builtin_amdgcn_readlane(f_lit, u), L0 = __builtin_amdgcn_readlane(f_L0, u),
                           o = __builtin_amdgcn_readlane(f_off, u);
            uint8_t *y = WL + u * WSTRIDE + lit_at;
            y[L0 + lane] = y[L0 - o + lmod(lane, o, 1.0f / (float)o)];
        }
    }
    // one step = (open the next row: head and tail bytes) + one trip of up to four 1 KiB stores
    __device__ __forceinline__ bool step_one() {
        if (!open_) {
            if (!todo) return false;
            const uint32_t u = (uint32_t)__builtin_ctz(todo);
            todo &= todo - 1;
            const uint32_t lit_at = __builtin_amdgcn_readlane(f_lit, u), L0 = __builtin_amdgcn_readlane(f_L0, u);
            off = __builtin_amdgcn_readlane(f_off, u);
            const uint32_t osize = __builtin_amdgcn_readlane(c_len_lo, u);
            const uint64_t oo = ((uint64_t)__builtin_amdgcn_readlane((uint32_t)(c_oo >> 32), u) << 32) |
                                __builtin_amdgcn_readlane((uint32_t)c_oo, u);
            out = outbase + oo;
            Y = WL + u * WSTRIDE + lit_at;
            B = L0 - off;
            lim = L0 + 64;
            const float inv = 1.0f / (float)off;
            const uint32_t head = (uint32_t)((16 - ((uintptr_t)out & 15)) & 15);
            body16 = (osize - head) >> 4;  // whole 16-byte pieces (>= 3: a recognised row has >= 65 bytes)
            step = smod(1024, off);
            const uint32_t rb = smod(B, off);
            if (lane < head) out[lane] = Y[lane];
            const uint32_t tail = (osize - head) & 15, p = head + 16 * body16 + lane;
            if (lane < tail) {
                uint32_t rt = lmod(p, off, inv) + off - rb;
                if (rt >= off) rt -= off;
                out[p] = p < lim ? Y[p] : Y[B + rt];
            }
            x = head + 16 * lane;
            r = lmod(x, off, inv) + off - rb;  // (x - B) mod off
            if (r >= off) r -= off;
            i0 = 0;
            open_ = 1;
        }
        {
            const uint32_t i = i0 + lane;
            uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0, v2 = v0, v3 = v0;
            const uint32_t r0 = r;
            uint32_t r1 = r0 + step; if (r1 >= off) r1 -= off;
            uint32_t r2 = r1 + step; if (r2 >= off) r2 -= off;
            uint32_t r3 = r2 + step; if (r3 >= off) r3 -= off;
            r = r3 + step; if (r >= off) r -= off;
            const bool p0 = i < body16, p1 = i + 64 < body16, p2 = i + 128 < body16, p3 = i + 192 < body16;
            if (p0) v0 = lds16(x + 16 <= lim ? Y + x : Y + B + r0);
            if (p1) v1 = lds16(x + 1040 <= lim ? Y + x + 1024 : Y + B + r1);
            if (p2) v2 = lds16(Y + B + r2);  // x + 2048 is past any window
            if (p3) v3 = lds16(Y + B + r3);
            uint8_t *d = out + x;
            if (p0) *reinterpret_cast<uint4 *>(d) = v0;
            if (p1) *reinterpret_cast<uint4 *>(d + 1024) = v1;
            if (p2) *reinterpret_cast<uint4 *>(d + 2048) = v2;
            if (p3) *reinterpret_cast<uint4 *>(d + 3072) = v3;
            x += 4096;
            i0 += 256;
            if (i0 >= body16) open_ = 0;
        }
        return true;
    }
    __device__ __forceinline__ void drain() {
        if (g_abl & 16) return;  // ablation: no stream-out
        while (step_one()) {}
    }
};
// The parent trees of a block's four tiles are folded TOGETHER by one of its waves: a parent level costs a whole
// compress pass of a wave however few lanes take part (a tile of six 10-leaf rows: 4 passes with 30, 12, 6, 6 busy
// lanes), so every wave leaves its leaf chaining values and unit shapes in LDS and one wave folds all of them with
// the nodes of a level packed over its 64 lanes — 6 passes for the block instead of 4 x 4.
struct BlockFold {
    uint32_t *nodes;                // 4 x 64 chaining values
    uint32_t *tab_n, *tab_off, *tab_out;  // 4 x FOLD_UNITS unit descriptors (n = 0: nothing to fold)
};
constexpr uint32_t FOLD_UNITS = 16;  // tiles with more units fold on the spot

__device__ __forceinline__ void fused_tile(const FusedArgs &a, const uint32_t wave, const BlockFold &bf) {
    __shared__ __attribute__((aligned(16))) uint8_t s_W[4][WROWS * WSTRIDE];
    const uint32_t lane = threadIdx.x & 63;
    const Tile t = a.h.tiles[wave];
    if (t.n_units == 0) return;  // slices of big rows: general decoder + second hash pass
    // the pattern buffer of the scalar decoder lives in this wave's slice of the node array it fills at the end
    static_assert(EBUF <= 64 * 8 * 4, "pattern buffer must fit the wave's node slice");
    uint8_t *const E = reinterpret_cast<uint8_t *>(bf.nodes + (size_t)(threadIdx.x >> 6) * 64 * 8);
    uint8_t *const WL = s_W[threadIdx.x >> 6];
    const bool stamp = (a.dbg & 8) && a.dbg_buf;  // diagnostic only: phase durations -> a.dbg_buf (never an output)
    __shared__ unsigned long long s_acc[4][8];
    if (lane < 8) s_acc[threadIdx.x >> 6][lane] = 0;
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    if (stamp) t0 = __builtin_amdgcn_s_memtime();

    // ---- tile prologue: the rows' index columns, one row per lane (coalesced), then the first
    // WIN bytes of up to WROWS frames staged into LDS with all loads in flight at once ----
    uint32_t c_sel = 0, c_bs = 0;
    uint64_t c_len = 0, c_src = 0, c_oo = 0;
    int32_t c_st = 0;  // host verdict on the row (bad source / output range): < 0 = leave it alone
    if (lane < t.n_units) {
        const uint32_t row = t.first_unit + lane;
        if (a.preset) c_st = a.status[row];
        c_sel = a.h.sel[row];
        c_len = a.h.len[row];
        c_src = a.h.offA[row] - a.h.baseA;
        c_oo = a.h.offB[row];
        c_bs = (uint32_t)a.blob_size[row];
    }
    __shared__ uint16_t s_desc[4][3][WROWS];  // per wave: ybase, B, off of each staged row
    uint16_t *const d_y = s_desc[threadIdx.x >> 6][0], *const d_B = s_desc[threadIdx.x >> 6][1],
                   *const d_off = s_desc[threadIdx.x >> 6][2];
    __shared__ int32_t s_st[4][64];  // per wave: status of each row of the tile (0 = hash it)
    int32_t *const l_st = s_st[threadIdx.x >> 6];
    if (lane < WROWS) d_y[lane] = 0xFFFF;
    l_st[lane] = c_st;
    uint32_t need_reread = 0;  // some compressed row of the tile must be hashed from its global output

    // A tile of many small rows has windows for WROWS of them at a time: rows WROWS.. are taken group by group
    // first — staged, recognised lane-parallel, written straight from their windows, hashed later from the output —
    // and the windows are then reused for the tile's first rows, which are hashed from them.
    uint64_t done64 = 0;  // rows already written by this pre-pass
    if (t.n_units > WROWS && !(a.dbg & (2 | 128))) {
        FastTabs T;
        {
            const DTab eL = c_dll[lane], eM = c_dml[lane], eO = c_dof[lane & 31];
            T.ll = eL.base | (uint32_t)eL.addbits << 24;
            T.ml = eM.base | (uint32_t)eM.addbits << 24;
            T.of = eO.addbits;
            T.lls = lane < 36 ? c_llb[lane] | (uint32_t)c_lla[lane] << 24 : 0u;
            T.mls = lane < 53 ? c_mlb[lane] | (uint32_t)c_mla[lane] << 24 : 0u;
        }
#pragma unroll 1
        for (uint32_t g0 = WROWS; g0 < t.n_units; g0 += WROWS) {
#pragma unroll 1
            for (uint32_t u = 0; u < WROWS && g0 + u < t.n_units; u++) {
                const uint64_t so = ((uint64_t)__shfl((uint32_t)(c_src >> 32), g0 + u) << 32) | __shfl((uint32_t)c_src, g0 + u);
                const uint32_t n = __shfl(c_bs, g0 + u), o = 8 * lane;
                const uint8_t *p = a.h.srcA + so + o;
                uint64_t v = 0;
                if (__shfl(c_sel, g0 + u) && __shfl(c_st, g0 + u) == 0) {
                    if (o + 8 <= n) __builtin_memcpy(&v, p, 8);
                    else
                        for (uint32_t k = 0; k < 8; k++)
                            if (o + k < n) v |= (uint64_t)p[k] << (8 * k);
                }
                *reinterpret_cast<uint64_t *>(WL + u * WSTRIDE + 8 * lane) = v;
                if (lane < (WSTRIDE - WIN) / 4) *reinterpret_cast<uint32_t *>(WL + u * WSTRIDE + WIN + 4 * lane) = 0;
            }
            const uint32_t gl = g0 + lane < 64 ? g0 + lane : 63;
            const uint32_t g_sel = __shfl(c_sel, gl), g_bs = __shfl(c_bs, gl);
            const uint64_t g_len = ((uint64_t)__shfl((uint32_t)(c_len >> 32), gl) << 32) | __shfl((uint32_t)c_len, gl);
            const uint64_t g_oo = ((uint64_t)__shfl((uint32_t)(c_oo >> 32), gl) << 32) | __shfl((uint32_t)c_oo, gl);
            const bool want = lane < WROWS && g0 + lane < t.n_units && g_sel && g_oo + g_len <= a.out_cap && __shfl(c_st, gl) == 0;
            const FastRow f = parse_fast(WL + (lane < WROWS ? lane : 0) * WSTRIDE, g_bs, g_len, want, T);
            const uint32_t m = (uint32_t)__ballot(f.ok != 0);
            if (m) {
                Emitter e2;
                e2.WL = WL; e2.outbase = a.h.srcB; e2.lane = lane;
                e2.f_lit = f.lit_at; e2.f_L0 = f.L0; e2.f_off = f.off; e2.c_len_lo = (uint32_t)g_len; e2.c_oo = g_oo;
                e2.prepare(m);
                e2.drain();
                done64 |= (uint64_t)m << g0;
                need_reread = 1;
            }
        }
    }

    if (!(a.dbg & 2)) {
        const uint32_t nw = t.n_units < WROWS ? t.n_units : WROWS;
        uint2 wv[WROWS];
#pragma unroll
        for (uint32_t u = 0; u < WROWS; u++) {
            wv[u] = make_uint2(0, 0);
            if (u < nw) {
                const uint64_t so = __shfl(c_src, u);
                const uint32_t n = __shfl(c_bs, u), o = 8 * lane;
                const uint8_t *p = a.h.srcA + so + o;
                if (__shfl(c_sel, u) && __shfl(c_st, u) == 0) {
                    if (o + 8 <= n) __builtin_memcpy(&wv[u], p, 8);
                    else {
                        uint64_t v = 0;
                        for (uint32_t k = 0; k < 8; k++)
                            if (o + k < n) v |= (uint64_t)p[k] << (8 * k);
                        wv[u] = make_uint2((uint32_t)v, (uint32_t)(v >> 32));
                    }
                }
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < WROWS; u++) {
            if (u < nw) {
                *reinterpret_cast<uint2 *>(WL + u * WSTRIDE + 8 * lane) = wv[u];
                if (lane < (WSTRIDE - WIN) / 4) *reinterpret_cast<uint32_t *>(WL + u * WSTRIDE + WIN + 4 * lane) = 0;
            }
        }
    }

    // every staged compressed row is parsed by its own lane; rows of the common shape never see the scalar parser
    FastRow fr{0, 0, 0, 0};
    if (!(a.dbg & (2 | 128))) {
        FastTabs T;
        {
            const DTab eL = c_dll[lane], eM = c_dml[lane], eO = c_dof[lane & 31];
            T.ll = eL.base | (uint32_t)eL.addbits << 24;
            T.ml = eM.base | (uint32_t)eM.addbits << 24;
            T.of = eO.addbits;
            T.lls = lane < 36 ? c_llb[lane] | (uint32_t)c_lla[lane] << 24 : 0u;
            T.mls = lane < 53 ? c_mlb[lane] | (uint32_t)c_mla[lane] << 24 : 0u;
        }
        const bool want = lane < t.n_units && lane < WROWS && c_sel && c_oo + c_len <= a.out_cap && c_st == 0;
        fr = parse_fast(WL + (lane < WROWS ? lane : 0) * WSTRIDE, c_bs, c_len, want, T);
        // rows that are certainly for the general decoders: handed over by their own lanes, one atomic per wave
        const bool ho = !fr.ok && not_simple_fast(WL + (lane < WROWS ? lane : 0) * WSTRIDE, c_bs, c_len, want);
        const uint64_t hom = __ballot(ho);
        if (hom) {
            uint32_t at = 0;
            if (lane == 0) at = atomicAdd(a.pending_count, (uint32_t)__popcll(hom));
            at = (uint32_t)__builtin_amdgcn_readlane((int)at, 0);
            if (ho) {
                const uint32_t row = t.first_unit + lane;
                a.status[row] = F_NOT_SIMPLE;
                l_st[lane] = F_NOT_SIMPLE;
                a.pending[at + (uint32_t)__popcll(hom & ((1ull << lane) - 1ull))] = row;
            }
            done64 |= hom;  // the row loop below skips them
        }
    }
    const uint32_t fmask = (uint32_t)__ballot(fr.ok != 0);
    Emitter em;
    em.WL = WL; em.outbase = a.h.srcB; em.lane = lane;
    em.f_lit = fr.lit_at; em.f_L0 = fr.L0; em.f_off = fr.off; em.c_len_lo = (uint32_t)c_len; em.c_oo = c_oo;
    em.prepare(fmask);
    if (fr.ok && !(a.dbg & 4)) {
        d_y[lane] = (uint16_t)(lane * WSTRIDE + fr.lit_at);
        d_B[lane] = (uint16_t)(fr.L0 - fr.off);
        d_off[lane] = (uint16_t)fr.off;
    }
    // a tile with any ragged leaf is hashed by the generic leaf path, which reads every row from the output and has
    // no room for side work: write the recognised rows before the hash
    const bool early = __ballot(lane < t.n_units && (c_len == 0 || ((uint32_t)c_len & 1023) != 0)) != 0ull || (a.dbg & (1 | 4));
    if (early && fmask) { em.drain(); need_reread = 1; }
    // otherwise every leaf of the tile is full and the recognised rows are written by the lanes that hash them
    if (stamp) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); t1 = __builtin_amdgcn_s_memtime(); }
    // decode = short bursts of scalar parsing + store issue: let it issue ahead of SIMD-mates that are in
    // their VALU-bound hash phase, so the stores get out early and drain while this wave hashes
    if (!(a.dbg & 64)) __builtin_amdgcn_s_setprio(3);
    for (uint32_t u = 0; u < t.n_units && !(a.dbg & 2); u++) {
        const uint32_t row = t.first_unit + u;
        if (!uni(__shfl(c_sel, u))) continue;  // stored row: copied while it is hashed below
        if ((int32_t)uni((uint32_t)__shfl(c_st, u)) < 0) continue;  // host verdict stands
        const uint64_t usize = ((uint64_t)uni((uint32_t)(__shfl(c_len, u) >> 32)) << 32) | uni((uint32_t)__shfl(c_len, u));
        const uint64_t ooff = ((uint64_t)uni((uint32_t)(__shfl(c_oo, u) >> 32)) << 32) | uni((uint32_t)__shfl(c_oo, u));
        const uint64_t soff = ((uint64_t)uni((uint32_t)(__shfl(c_src, u) >> 32)) << 32) | uni((uint32_t)__shfl(c_src, u));
        const uint32_t bsz = uni(__shfl(c_bs, u));
        if ((u < 32 && (fmask >> u & 1)) || (done64 >> u & 1)) continue;  // recognised row: written from its window
        int rc;
        Periodic per;
        per.ok = 0;
        if (ooff + usize > a.out_cap) rc = F_E_DST;
        else rc = decode_simple(a.h.srcA + soff, bsz, a.h.srcB + ooff, usize, E, u < WROWS ? WL + u * WSTRIDE : nullptr, &per,
                                stamp ? s_acc[threadIdx.x >> 6] : nullptr);
        rc = (int)uni((uint32_t)rc);
        const bool periodic = rc == 0 && uni(per.ok) && u < WROWS && !(a.dbg & 4);
        if (periodic) {
            if (lane == 0) {
                d_y[u] = (uint16_t)(u * WSTRIDE + per.lit_at);
                d_B[u] = (uint16_t)(per.L0 - per.off);
                d_off[u] = (uint16_t)per.off;
            }
        } else if (rc == 0) {
            need_reread = 1;
        }
        if (rc != 0 && lane == 0) {
            a.status[row] = rc;
            l_st[u] = rc;
            if (rc == F_NOT_SIMPLE) a.pending[atomicAdd(a.pending_count, 1u)] = row;
        }
    }
    // Rows hashed from LDS do not depend on their stores having landed, so the store queue drains
    // in the background while this wave hashes.  Only a global re-read needs the drain.
    __builtin_amdgcn_s_setprio(0);
    if (need_reread) fwave_mem_sync();
    if (stamp) t2 = __builtin_amdgcn_s_memtime();
    if (!(a.dbg & 1)) {
        LdsSrc ls{WL, d_y, d_B, d_off, WROWS, l_st, c_len, c_src, c_oo, c_sel, (early || (g_abl & 16)) ? 0u : fmask};
        LeafOut lo;
        hash_tile_leaves<true, true>(a.h, t, &ls, lo);
        if (t.n_units <= FOLD_UNITS && !(a.dbg & 256)) {
            const uint32_t w = threadIdx.x >> 6;
            if (lo.active) {
                uint4 *d = reinterpret_cast<uint4 *>(bf.nodes + (size_t)(w * 64 + lane) * 8);
                d[0] = make_uint4(lo.cv[0], lo.cv[1], lo.cv[2], lo.cv[3]);
                d[1] = make_uint4(lo.cv[4], lo.cv[5], lo.cv[6], lo.cv[7]);
            }
            const uint32_t act = __shfl(lo.active ? 1u : 0u, lo.u_head & 63);  // a unit is hashed as a whole or not at all
            if (lane < t.n_units) {
                bf.tab_n[w * FOLD_UNITS + lane] = act ? lo.u_cnt : 0u;
                bf.tab_off[w * FOLD_UNITS + lane] = w * 64 + lo.u_head;
                bf.tab_out[w * FOLD_UNITS + lane] = t.first_unit + lane;
            }
        } else {
            fold_tile_now(a.h, t, lo);
        }
    }
    if (stamp) {
        t3 = __builtin_amdgcn_s_memtime();
        if (lane == 0) {
            atomicAdd(&a.dbg_buf[0], t1 - t0);
            atomicAdd(&a.dbg_buf[1], t2 - t1);
            atomicAdd(&a.dbg_buf[2], t3 - t2);
            atomicAdd(&a.dbg_buf[3], 1ull);
            for (int q = 4; q < 8; q++) atomicAdd(&a.dbg_buf[q], s_acc[threadIdx.x >> 6][q]);
        }
    }
}

__device__ __forceinline__ void lds_barrier() {  // workgroup barrier that orders LDS only (pending global stores stay in flight)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__global__ __launch_bounds__(256, 5) void k_fused_small(FusedArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t s_nodes[4 * 64 * 8];
    __shared__ uint32_t s_tab[3][4 * FOLD_UNITS];
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool clk = (a.dbg & 32768) && a.dbg_buf && blockIdx.x == 2000 && threadIdx.x == 0;
    const unsigned long long c0 = clk ? __builtin_amdgcn_s_memtime() : 0, r0 = clk ? __builtin_amdgcn_s_memrealtime() : 0;
    const BlockFold bf{s_nodes, s_tab[0], s_tab[1], s_tab[2]};
    // the work: every tile of the plan, or the tiles k_fused_roles left on its list (count known on the device only:
    // the grid is capped and strides over it, so an empty list costs a launch of blocks that exit at once)
    const uint32_t n_work = a.tile_list ? *a.tile_count : a.h.n_tiles;
    for (uint32_t base = blockIdx.x * 4, it = 0; base < n_work; base += gridDim.x * 4, it++) {
        if (lane < FOLD_UNITS) bf.tab_n[w * FOLD_UNITS + lane] = 0;
        const uint32_t idx = base + w;
        if (idx < n_work) fused_tile(a, a.tile_list ? a.tile_list[idx] : idx, bf);
        lds_barrier();
        // the folding wave changes from block to block, so that over the blocks resident on a CU the extra passes
        // spread over its four SIMDs
        if (w == ((blockIdx.x + it) * 2654435761u) >> 30) {
            FoldQueue<4> fq;
            fq.n_tab = 4 * FOLD_UNITS;
            fq.tb_n = bf.tab_n[lane];
            fq.tb_off = bf.tab_off[lane];
            fq.tb_out = bf.tab_out[lane];
            fq.tb_root = 1;
            if (__ballot(fq.tb_n != 0) != 0ull) {
                FoldQueue<4> fd = fq.dense16();
                uint32_t U = 0;
                const uint32_t n = (a.dbg & 262144) ? 0u : fd.uniform(&U);
                if (n) fd.fold_uniform_and_write(s_nodes, a.h, n, U);  // every row of the four tiles has the same length
                else fq.fold_and_write(s_nodes, a.h);
            }
        }
        if (base + gridDim.x * 4 < n_work) lds_barrier();  // the node array and the tables are reused by the next round
    }
    if (clk) { a.dbg_buf[0] = __builtin_amdgcn_s_memtime() - c0; a.dbg_buf[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

// ---- role-split persistent kernel ---------------------------------------------------------------------------
// Tiles whose rows are ALL whole-leaf rows of the recognised periodic shape (every BASELINE text row) are taken by
// persistent workgroups — one per CU — of a LOADER wave and fifteen HASHER waves around a ring of slots in LDS:
//   loader : pulls four tiles at a time from a global cursor, loads their index columns and frames (lane = row),
//            recognises the rows lane-parallel (parse_fast) and publishes the four windows as a group of slots.  It
//            owns every global load of the workgroup and runs up to seven groups ahead of the hashers.
//   hasher : takes the next ready slot, gives every row's window 64 more bytes of its period, hashes the tile's
//            <= 64 leaves straight from the windows (an all-LDS loop) while its lanes store the message registers
//            as the row's bytes, leaves the chaining values in the slot; the wave that completes a group folds the
//            four tiles' parent trees together and writes the digests.  A hasher never waits for a global load.
// Any other tile goes to the slow list and k_fused_small afterwards.
//
// What was measured on the way (MI355X, 100k x 10 KiB rows; profiles/README.md):
//  * `if (lane == 0) x = atomicAdd(..); x = readfirstlane(x);` is a trap: x is 0 in the other lanes, and the compiler's
//    structured control flow ran the loop body a second time for lanes 1..63 with that 0 — hashers went through
//    "take 0" again in the middle of a run, wrote chaining values over slot 0's fresh windows and arrived at group 0 a
//    fifth time.  Every atomic here is executed by ALL lanes (lane 0 adds, the others add 0).  (For a while this was
//    taken for a hardware problem with 16-byte LDS stores at odd addresses and then with half-wave tearing of flag
//    reads; neither was the cause, but the code keeps byte / aligned LDS stores and scalar flag reads.)
//  * the loader writing the rows itself (whole lines, 1 KiB per store instruction, emit_periodic_row) made it the
//    bottleneck: such a store costs the issuing wave ~200 cycles whatever feeds it (the CU's store path moves
//    ~5-10 bytes per cycle), 95-110 k cycles per group of four tiles; two loaders per workgroup halved each one's
//    rate; so the hash lanes store their registers as in k_fused_small (R_LOADER_EMITS = 0);
//  * a volatile access through a generic pointer is a FLAT instruction; the ring's control words are explicit LDS
//    accesses.
#ifndef ZN_R_SLOTS
#define ZN_R_SLOTS 32
#endif
#ifndef ZN_R_WAVES
#define ZN_R_WAVES 16
#endif
#ifndef ZN_R_LOADERS
#define ZN_R_LOADERS 1
#endif
#ifndef ZN_R_LOADER_EMITS
#define ZN_R_LOADER_EMITS 0
#endif
constexpr uint32_t R_SLOTS = ZN_R_SLOTS, R_GROUP = 4, R_WAVES = ZN_R_WAVES, R_LOADERS = ZN_R_LOADERS;
constexpr bool R_LOADER_EMITS = ZN_R_LOADER_EMITS != 0;
static_assert((R_SLOTS & (R_SLOTS - 1)) == 0 && R_SLOTS >= 2 * R_GROUP, "ring size: a power of two, at least two groups");
constexpr uint32_t R_SLOT_BYTES = WROWS * WSTRIDE;        // the windows; afterwards the tile's leaf CVs (first 2 KiB)
constexpr uint32_t R_SLOT_NODES = R_SLOT_BYTES / 32;
static_assert(R_SLOT_BYTES % 32 == 0 && R_SLOT_BYTES >= 64 * 32, "a slot must hold 64 chaining values, node-aligned");

// Control words of the ring.  A wave reads a word with ONE ds_read_b32, but that instruction is served in two halves
// of 32 lanes: a store from another wave may land between them and leave the halves with different values.  Every
// read is made scalar (lane 0's copy) the moment it arrives, so that a branch on it is uniform by construction.
typedef __attribute__((address_space(3))) uint32_t lds_u32;
__device__ __forceinline__ uint32_t lds_ld(const uint32_t *p) { return __builtin_amdgcn_readfirstlane(*(const volatile lds_u32 *)p); }
__device__ __forceinline__ void lds_st(uint32_t *p, uint32_t v) { *(volatile lds_u32 *)p = v; }
__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

struct alignas(16) RolesShared {
    uint8_t slots[R_SLOTS][R_SLOT_BYTES];
    // per hasher wave: the 60 leaf chaining values of the tile it hashed LAST (6 rows x 10 leaves), folded in place by
    // lanes 60..63 while the wave's next tile is hashed (roles_hash_6x10).  (Kept in front of the descriptor arrays: the
    // idle operations of a level's last pass read a few hundred bytes past the 2 KiB — never write.)
    __attribute__((aligned(16))) uint32_t tree[R_WAVES - R_LOADERS][64 * 8];
    uint64_t oo[R_SLOTS][WROWS];
    uint32_t len[R_SLOTS][WROWS];
    uint32_t tn[R_SLOTS][WROWS], toff[R_SLOTS][WROWS];  // fold table of the slot's units (written by its hasher)
    uint16_t dy[R_SLOTS][WROWS], dB[R_SLOTS][WROWS], doff[R_SLOTS][WROWS];
    uint32_t first[R_SLOTS], nunits[R_SLOTS], nleaves[R_SLOTS];
    uint32_t ready, take, finished;      // slots published / claimed so far; loaders that are done
    uint32_t next_iter;                  // the loaders' iteration counter (an iteration = one group of four slots)
    uint32_t cnt[R_SLOTS / R_GROUP];     // hashed slots of the group's current use
    uint32_t gen[R_SLOTS / R_GROUP];     // completed uses of the group (a loader refills it when gen == its next use)
};

// One recognised row written by the whole wave from its window (R_LOADER_EMITS; lane = byte / 16-byte piece).
// Y = the row's window at its literals:
//   out[i] = Y[i]                         for i <  L0          (the literals; the period is their last `off` bytes)
//   out[i] = Y[B + (i - B) mod off]       for i >= B = L0 - off
// First the window gets 64 more bytes of the period behind the literals, so that any 16 output bytes are 16
// CONTIGUOUS window bytes (at i itself, or at B + phase); then the row goes out as aligned 16-byte global stores,
// 1 KiB per instruction, four in flight; head and tail bytes lane-parallel.
__device__ __forceinline__ void emit_periodic_row(uint8_t *Y, uint32_t L0, uint32_t off, float inv, uint8_t *out, uint32_t osize,
                                                  uint32_t lane) {
    const uint32_t B = L0 - off, lim = L0 + 64;
    Y[L0 + lane] = Y[B + lmod(lane, off, inv)];
    const uint32_t head = (uint32_t)((16 - ((uintptr_t)out & 15)) & 15);
    const uint32_t body16 = (osize - head) >> 4;
    const uint32_t step = smod(1024, off), rb = smod(B, off);
    if (lane < head) out[lane] = Y[lane];                                 // head < 16 <= lim
    const uint32_t tail = (osize - head) & 15, p = head + 16 * body16 + lane;
    if (lane < tail) {
        uint32_t rt = lmod(p, off, inv) + off - rb;                       // (p - B) mod off
        if (rt >= off) rt -= off;
        out[p] = p < lim ? Y[p] : Y[B + rt];
    }
    uint32_t x = head + 16 * lane;
    uint32_t r = lmod(x, off, inv) + off - rb;                            // (x - B) mod off
    if (r >= off) r -= off;
    for (uint32_t i0 = 0; i0 < body16; i0 += 256) {
        const uint32_t i = i0 + lane;
        uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0, v2 = v0, v3 = v0;
        const uint32_t r0 = r;
        uint32_t r1 = r0 + step; if (r1 >= off) r1 -= off;
        uint32_t r2 = r1 + step; if (r2 >= off) r2 -= off;
        uint32_t r3 = r2 + step; if (r3 >= off) r3 -= off;
        r = r3 + step; if (r >= off) r -= off;
        const bool p0 = i < body16, p1 = i + 64 < body16, p2 = i + 128 < body16, p3 = i + 192 < body16;
        if (p0) v0 = lds16(x + 16 <= lim ? Y + x : Y + B + r0);           // only the row's first piece can lie in the literals
        if (p1) v1 = lds16(Y + B + r1);
        if (p2) v2 = lds16(Y + B + r2);
        if (p3) v3 = lds16(Y + B + r3);
        uint8_t *d = out + x;
        if (p0) *reinterpret_cast<uint4 *>(d) = v0;
        if (p1) *reinterpret_cast<uint4 *>(d + 1024) = v1;
        if (p2) *reinterpret_cast<uint4 *>(d + 2048) = v2;
        if (p3) *reinterpret_cast<uint4 *>(d + 3072) = v3;
        x += 4096;
    }
}

__device__ __forceinline__ void roles_loader(const FusedArgs &a, RolesShared &S) {
    const uint32_t lane = threadIdx.x & 63, j = lane >> 4, u = lane & 15;  // 16 lanes per tile, lane u = row u of tile j
    FastTabs T;
    {
        const DTab eL = c_dll[lane], eM = c_dml[lane], eO = c_dof[lane & 31];
        T.ll = eL.base | (uint32_t)eL.addbits << 24;
        T.ml = eM.base | (uint32_t)eM.addbits << 24;
        T.of = eO.addbits;
        T.lls = lane < 36 ? c_llb[lane] | (uint32_t)c_lla[lane] << 24 : 0u;
        T.mls = lane < 53 ? c_mlb[lane] | (uint32_t)c_mla[lane] << 24 : 0u;
    }
    for (;;) {
        // iterations are numbered across the workgroup's loaders (their slots are published in that order); the tiles
        // come from the global cursor.  A loader whose batch lies behind the last tile still fills and publishes its
        // iteration — four empty slots — because another loader may hold a later iteration with real tiles; then
        // it stops.
        // (lane 0's value is given to EVERY lane before it is made scalar: readfirstlane reads the first ACTIVE lane, and
        // the compiler may re-evaluate it at a later point where lane 0 is masked off — it did: hashers came out with
        // take = 0 in the middle of a run and worked on slot 0 a second time)
        // (EVERY lane executes the atomic — lane 0 adds, the others add 0 — and lane 0's result is made scalar.  The
        // usual `if (lane == 0) x = atomicAdd(..)` leaves x = 0 in the other lanes, and the compiler's structured
        // control flow ran the rest of the loop body a SECOND time for lanes 1..63 with that 0: measured — hashers went
        // through "take 0" again in the middle of a run and overwrote slot 0.)
        const uint32_t iter = uni(atomicAdd(&S.next_iter, lane == 0 ? 1u : 0u));
        const uint32_t base = uni(atomicAdd(a.cursor, lane == 0 ? R_GROUP : 0u));
        const bool last = base >= a.h.n_tiles;
        const uint32_t grp = iter & (R_SLOTS / R_GROUP - 1), use = iter / (R_SLOTS / R_GROUP);
        // ---- the four tiles' index columns: issued before the wait for the group, they travel meanwhile ----
        const uint32_t ti = base + j;
        Tile t{0, 0, 0, 0, 0, 0};
        if (ti < a.h.n_tiles) t = a.h.tiles[ti];
        const bool small = t.n_units >= 1;                // (a slice of a big row otherwise: not this kernel's)
        const bool cand = small && t.n_units <= WROWS;    // every row gets a window
        const bool rowv = cand && u < t.n_units;
        uint32_t c_sel = 0, c_bs = 0;
        uint64_t c_len = 0, c_src = 0, c_oo = 0;
        int32_t c_st = 0;
        if (rowv) {
            const uint32_t row = t.first_unit + u;
            if (a.preset) c_st = a.status[row];
            c_sel = a.h.sel[row];
            c_len = a.h.len[row];
            c_src = a.h.offA[row] - a.h.baseA;
            c_oo = a.h.offB[row];
            const uint64_t bs = a.blob_size[row];
            c_bs = bs > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)bs;
        }
        const bool want = rowv && c_sel && c_st == 0 && c_len != 0 && (c_len & 1023) == 0 && c_len <= 0x10000 &&
                          c_oo + c_len <= a.out_cap && c_bs >= 12 && c_bs <= WIN;
        while (lds_ld(&S.gen[grp]) != use) __builtin_amdgcn_s_sleep(4);  // the group's previous use has been folded
        asm volatile("" ::: "memory");  // (volatile accesses order only against one another: pin the window stores below)
        // ---- the frames, lane = row: 16 bytes per step into the row's window (the last partial piece bytewise: the
        // blob region promises nothing behind its last byte) ----
        const uint32_t slot = grp * R_GROUP + j;
        uint8_t *const win = S.slots[slot] + (u < WROWS ? u : 0) * WSTRIDE;
        const uint8_t *const src = a.h.srcA + c_src;
        const uint32_t nb = want ? c_bs : 0, full = nb >> 4, tail = nb & 15;
        for (uint32_t i = 0; __ballot(i < full) != 0ull; i++)
            if (i < full) {
                const uint4 v = ld16(src + 16 * i);
                *reinterpret_cast<uint4 *>(win + 16 * i) = v;
            }
        if (tail) {
            uint32_t w4[4] = {0, 0, 0, 0};
            for (uint32_t k = 0; k < tail; k++) w4[k >> 2] |= (uint32_t)src[16 * full + k] << (8 * (k & 3));
            *reinterpret_cast<uint4 *>(win + 16 * full) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
        }
        const FastRow fr = parse_fast(win, c_bs, c_len, want, T);
        const uint64_t badm = __ballot(rowv && !fr.ok);
        const bool fast = cand && ((badm >> (16 * j)) & 0xFFFFull) == 0;
        if (small && !fast && u == 0) a.tile_list[atomicAdd(a.tile_count, 1u)] = ti;  // k_fused_small takes it
        if (u == 0) {
            S.first[slot] = t.first_unit;
            S.nunits[slot] = fast ? t.n_units : 0u;
            S.nleaves[slot] = t.n_leaves;
        }
        const bool mine = fast && rowv;
        if (mine) {
            S.len[slot][u] = (uint32_t)c_len;
            S.oo[slot][u] = c_oo;
            S.dy[slot][u] = (uint16_t)(u * WSTRIDE + fr.lit_at);
            S.dB[slot][u] = (uint16_t)(fr.L0 - fr.off);
            S.doff[slot][u] = (uint16_t)fr.off;
        }
        lds_fence();
        if (R_LOADER_EMITS) {  // (off by default: see the measurements at the top)
            const float inv_l = 1.0f / (float)(mine ? fr.off : 1u);
            for (uint64_t m = __ballot(mine); m && !(a.dbg & 16); m &= m - 1) {
                const uint32_t l = (uint32_t)__builtin_ctzll(m);
                const uint32_t sl = grp * R_GROUP + (l >> 4);
                const uint32_t lit_at = __builtin_amdgcn_readlane(fr.lit_at, l), L0 = __builtin_amdgcn_readlane(fr.L0, l),
                               off = __builtin_amdgcn_readlane(fr.off, l), osize = __builtin_amdgcn_readlane((uint32_t)c_len, l);
                const float inv = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(inv_l), l));
                const uint64_t oo = ((uint64_t)__builtin_amdgcn_readlane((uint32_t)(c_oo >> 32), l) << 32) |
                                    __builtin_amdgcn_readlane((uint32_t)c_oo, l);
                emit_periodic_row(S.slots[sl] + (l & 15) * WSTRIDE + lit_at, L0, off, inv, a.h.srcB + oo, osize, lane);
            }
            lds_fence();
        }
        while (lds_ld(&S.ready) != iter * R_GROUP) __builtin_amdgcn_s_sleep(2);  // publish in iteration order
        if (lane == 0) lds_st(&S.ready, (iter + 1) * R_GROUP);
        if (last) break;
    }
    lds_fence();
    atomicAdd(&S.finished, lane == 0 ? 1u : 0u);
}

// ---- parent trees without passes of their own ------------------------------------------------------------------
// A tile of BASELINE's shape is 6 rows x 10 leaves = 60 busy lanes per compress pass, and its 54 parent nodes used to
// cost the workgroup 1.5 more passes per tile (folded four tiles at a time by whoever finished the group).  Here the
// four idle lanes of every leaf pass compute parent nodes — of the tile this wave hashed BEFORE the current one, whose
// leaf CVs wait in the wave's private node array: 16 passes x 4 lanes = 64 slots for 54 nodes, level by level
// (10 leaves: 5 + 2 + 1 + 1 per row), every node written over its left child.  Node operation n of the schedule:
//   n in [ 0,30)  level 1   row n/5, pair j = n%5:   (10 row + 2j, + 1)
//   n in [32,44)  level 2   row (n-32)/2, j:         (10 row + 4j, + 2)      [the fifth level-1 node is carried]
//   n in [44,50)  level 3   row n-44:                (10 row, + 4)
//   n in [52,58)  root      row n-52:                (10 row, + 8)  -> the row's digest
// Pass b runs operations 4b .. 4b+3, so a level starts a full pass after the previous one ended.
__device__ __forceinline__ bool tree_op_6x10(uint32_t n, uint32_t &l, uint32_t &r, bool &root) {
    root = false;
    if (n < 30) { const uint32_t row = n / 5, j = n - 5 * row; l = row * 10 + 2 * j; r = l + 1; return true; }
    if (n >= 32 && n < 44) { const uint32_t m = n - 32, row = m >> 1, j = m & 1; l = row * 10 + 4 * j; r = l + 2; return true; }
    if (n >= 44 && n < 50) { l = (n - 44) * 10; r = l + 4; return true; }
    if (n >= 52 && n < 58) { l = (n - 52) * 10; r = l + 8; root = true; return true; }
    l = r = 0;
    return false;
}

// The leaf pass of a 6 x 10 tile staged in `slot` (windows extended by the caller), rows stored from the message registers;
// lanes 60..63 fold the previous tile's tree (have_prev) in `tree` and write its digests — as VIRTUAL LEAVES: a parent's
// message is its two children's chaining values, which sit
// in the node array at a fixed distance from each other on every level — so lanes 60..63 fetch their 64 message bytes with
// the very loads the leaf lanes use (base pointer = the node array, no period to wrap around), with a per-lane stride
// from pass to pass and a per-lane gap between the two halves of the message:
//   level 1, passes  0..7    operation n = 4b + s:  nodes 2n | 2n+1            offset 64 s (+256 per pass), gap 0
//   level 2, passes  8..10   m = 4(b-8) + s:        10 row + 4j | + 2          320 (s>>1) + 128 (s&1) (+640), gap 32
//   level 3, passes 11..12   row = 4(b-11) + s:     10 row | + 4               320 s (+1280), gap 96
//   root,    passes 13..14   row = 4(b-13) + s:     10 row | + 8  -> digest    320 s (+1280), gap 224
// A result goes over its left child (the address just read).  What is left per pass for the parent lanes is the reset of
// their chaining value, one select for the flags and the result's two stores.
__device__ __forceinline__ void roles_hash_6x10(const FusedArgs &a, RolesShared &S, uint32_t slot, uint32_t *tree, bool have_prev,
                                                   uint32_t prev_first, uint32_t cv[8]) {
    typedef __attribute__((address_space(3))) uint8_t lds8b;
    const uint32_t lane = threadIdx.x & 63;
    const bool leaf = lane < 60;
    const uint32_t row = leaf ? lane / 10 : 0, kk = leaf ? lane - 10 * row : 0, ps = lane & 3;
    const lds8b *const Y = leaf ? (const lds8b *)(S.slots[slot] + S.dy[slot][row]) : (const lds8b *)tree;
    const uint32_t yB = S.dB[slot][row], yoff = S.doff[slot][row];
    const uint32_t yL0 = leaf ? yB + yoff : 0xFFFFFFFFu;
    uint32_t p = leaf ? kk << 10 : 64 * ps, step = leaf ? 64u : 256u, gap = 0;
    const uint32_t p1 = (kk << 10) > yB + yoff ? (kk << 10) : (((yB + yoff) >> 6) + 1) << 6;  // first block read through the period
    uint32_t r = (p1 - yB) % yoff;
    const uint32_t step64 = 64 % yoff;
    uint8_t *const dst = a.h.srcB + S.oo[slot][row] + (kk << 10);
    uint32_t *const dig = a.h.digests + (size_t)(prev_first + ps) * 8;
    // passes in which this parent lane has an operation: all of 0..14, less the last pass of levels 1, 3 and 4 for lanes 2, 3
    const uint32_t vmask = (leaf || !have_prev) ? 0u : (ps < 2 ? 0x7FFFu : 0x7FFFu & ~((1u << 7) | (1u << 12) | (1u << 14)));
    const bool store = leaf && !(a.dbg & 16);
    b3::set_iv(cv);
    uint4 n0, n1, n2, n3;
    uint32_t wq = 0, wq_next = 0;  // where the pass's result goes: the offset its message was read from
    auto fetch = [&]() {
        uint32_t q = p;
        if (p > yL0) {
            q = yB + r;
            r += step64;
            if (r >= yoff) r -= yoff;
        }
        const lds_u4 *q3 = (const lds_u4 *)(Y + q), *q4 = (const lds_u4 *)(Y + q + 32 + gap);
        const u4v a0 = q3[0], a1 = q3[1], a2 = q4[0], a3 = q4[1];
        n0 = make_uint4(a0.x, a0.y, a0.z, a0.w); n1 = make_uint4(a1.x, a1.y, a1.z, a1.w);
        n2 = make_uint4(a2.x, a2.y, a2.z, a2.w); n3 = make_uint4(a3.x, a3.y, a3.z, a3.w);
        wq_next = q;
        p += step;
    };
    fetch();
#pragma unroll 1
    for (uint32_t b = 0; b < 16; b++) {
        uint32_t m[16] = {n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, n3.x, n3.y, n3.z, n3.w};
        wq = wq_next;
        if (b < 15) {
            if (b == 7 || b == 10 || b == 12) {  // the parent lanes' next level (uniform branch; three passes of sixteen)
                if (!leaf) {
                    if (b == 7) { p = 320 * (ps >> 1) + 128 * (ps & 1); step = 640; gap = 32; }
                    else if (b == 10) { p = 320 * ps; step = 1280; gap = 96; }
                    else { p = 320 * ps; gap = 224; }
                }
            }
            fetch();
        }
        if (store) {
            uint8_t *d = dst + b * 64;
            st16(d, make_uint4(m[0], m[1], m[2], m[3]));
            st16(d + 16, make_uint4(m[4], m[5], m[6], m[7]));
            st16(d + 32, make_uint4(m[8], m[9], m[10], m[11]));
            st16(d + 48, make_uint4(m[12], m[13], m[14], m[15]));
        }
        const uint32_t fl = (b == 0 ? b3::CHUNK_START : 0u) | (b == 15 ? b3::CHUNK_END : 0u);
        const uint32_t fp = b3::PARENT | (b >= 13 ? b3::ROOT : 0u);
        if (!leaf) b3::set_iv(cv);
        b3::compress(cv, m, kk, 0, 64, leaf ? fl : fp);
        if ((vmask >> b) & 1u) {
            if (b >= 13) {
                uint4 *o = reinterpret_cast<uint4 *>(dig + (b - 13) * 32);
                o[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
                o[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
            } else {
                lds_u4a *o = (lds_u4a *)(Y + wq);
                o[0] = u4v{cv[0], cv[1], cv[2], cv[3]};
                o[1] = u4v{cv[4], cv[5], cv[6], cv[7]};
            }
        }
    }
}

// The tree of the last tile a wave hashed, with nobody's leaf pass to ride on: four level steps, all lanes.
__device__ __forceinline__ void roles_flush_tree_6x10(const FusedArgs &a, uint32_t *tree, uint32_t prev_first) {
    const uint32_t lane = threadIdx.x & 63;
#pragma unroll 1
    for (uint32_t lv = 0; lv < 4; lv++) {
        const uint32_t base = lv == 0 ? 0u : (lv == 1 ? 32u : (lv == 2 ? 44u : 52u));
        const uint32_t cntl = lv == 0 ? 30u : (lv == 1 ? 12u : 6u);
        uint32_t l = 0, rr = 0;
        bool root = false;
        const bool on = lane < cntl && tree_op_6x10(base + lane, l, rr, root);
        uint32_t cv[8], L[8], R[8];
        const lds_u4a *pl = (const lds_u4a *)(tree + l * 8), *pr = (const lds_u4a *)(tree + rr * 8);
        const u4v l0 = pl[0], l1 = pl[1], r0 = pr[0], r1 = pr[1];
        L[0] = l0.x; L[1] = l0.y; L[2] = l0.z; L[3] = l0.w; L[4] = l1.x; L[5] = l1.y; L[6] = l1.z; L[7] = l1.w;
        R[0] = r0.x; R[1] = r0.y; R[2] = r0.z; R[3] = r0.w; R[4] = r1.x; R[5] = r1.y; R[6] = r1.z; R[7] = r1.w;
        b3::parent(cv, L, R, root);
        __builtin_amdgcn_wave_barrier();  // every lane has read its children (a node only ever replaces its own left child)
        if (on) {
            if (root) {
                uint4 *o = reinterpret_cast<uint4 *>(a.h.digests + (size_t)(prev_first + l / 10) * 8);
                o[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
                o[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
            } else {
                lds_u4a *o = (lds_u4a *)(tree + l * 8);
                o[0] = u4v{cv[0], cv[1], cv[2], cv[3]};
                o[1] = u4v{cv[4], cv[5], cv[6], cv[7]};
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

__device__ __forceinline__ void roles_hasher(const FusedArgs &a, RolesShared &S) {
    const uint32_t lane = threadIdx.x & 63;
    HashArgs h = a.h;
    h.pass = PASS_ALL;
    uint32_t *const tree = S.tree[(threadIdx.x >> 6) - R_LOADERS];
    bool have_prev = false;       // `tree` holds the leaf CVs of a 6 x 10 tile whose parents are still to be computed
    uint32_t prev_first = 0;      // ... its first row
    const bool inline_tree = !(a.dbg & 524288);
    for (;;) {
        const uint32_t take = uni(atomicAdd(&S.take, lane == 0 ? 1u : 0u));  // every lane executes it: see roles_loader
        bool go = false;
        for (;;) {
            const uint32_t f = lds_ld(&S.finished);  // (read BEFORE ready: a loader's last slots are published before it says so)
            if (lds_ld(&S.ready) > take) { go = true; break; }
            if (f == R_LOADERS) break;               // nothing more will be published
            __builtin_amdgcn_s_sleep(2);
        }
        if (!go) {
            if (have_prev) roles_flush_tree_6x10(a, tree, prev_first);
            return;
        }
        asm volatile("" ::: "memory");
        const uint32_t slot = take & (R_SLOTS - 1), grp = slot / R_GROUP;
        const uint32_t nu = lds_ld(&S.nunits[slot]);
        if (nu) {
            const Tile t{lds_ld(&S.first[slot]), nu, 0, lds_ld(&S.nleaves[slot]), 0, 0};
            const uint32_t lu = lane < nu ? lane : 0;
            if (!R_LOADER_EMITS) {
                // 64 more bytes of each row's period behind its literals (lane = byte):
                // every row's byte is read first, then all are written — two LDS round trips for the tile
                uint8_t bytes[WROWS];
#pragma unroll
                for (uint32_t q = 0; q < WROWS; q++) {
                    bytes[q] = 0;
                    if (q < nu) {
                        const uint32_t off = S.doff[slot][q];
                        bytes[q] = S.slots[slot][S.dy[slot][q] + S.dB[slot][q] + lmod(lane, off, 1.0f / (float)off)];
                    }
                }
#pragma unroll
                for (uint32_t q = 0; q < WROWS; q++)
                    if (q < nu) S.slots[slot][S.dy[slot][q] + S.dB[slot][q] + S.doff[slot][q] + lane] = bytes[q];
            }
            // BASELINE's tile shape: the previous tile's parent tree rides in the four idle lanes of this tile's passes
            const bool shape610 = inline_tree && !R_LOADER_EMITS && nu == 6 && t.n_leaves == 60 && !(a.dbg & 8192) &&
                                  __ballot(lane < 6 && S.len[slot][lu] != 10240u) == 0ull;
            if (shape610) {
                uint32_t cv[8];
                roles_hash_6x10(a, S, slot, tree, have_prev, prev_first, cv);
                if (lane < 60) {  // the previous tree is done (its last node ran in pass 14): this tile's leaves take its place
                    lds_u4a *o = (lds_u4a *)(tree + lane * 8);
                    o[0] = u4v{cv[0], cv[1], cv[2], cv[3]};
                    o[1] = u4v{cv[4], cv[5], cv[6], cv[7]};
                }
                have_prev = true;
                prev_first = t.first_unit;
                if (lane < WROWS) S.tn[slot][lane] = 0u;  // nothing left for the group's fold
            } else {
            // the rows are stored by the lanes that hash them (their message registers) unless the loader wrote them
            LdsSrc ls{S.slots[slot], S.dy[slot], S.dB[slot], S.doff[slot], WROWS, nullptr,
                      lane < nu ? (uint64_t)S.len[slot][lu] : 0ull, 0ull, S.oo[slot][lu], 1u,
                      (R_LOADER_EMITS || (a.dbg & 16)) ? 0u : 0x3Fu, 0ull};
            LeafOut lo;
            if (a.dbg & 8192) {  // ablation: no hashing (the loaders' pace alone)
                lo.active = false; lo.u_cnt = 0; lo.u_head = 0;
#pragma unroll
                for (int q = 0; q < 8; q++) lo.cv[q] = 0;
            } else
                hash_tile_leaves<!R_LOADER_EMITS, true>(h, t, &ls, lo);
            // the windows are dead (every lane has read its last block): the leaf CVs take their place
            if (lo.active) {
                uint4 *d = reinterpret_cast<uint4 *>(S.slots[slot] + lane * 32);
                d[0] = make_uint4(lo.cv[0], lo.cv[1], lo.cv[2], lo.cv[3]);
                d[1] = make_uint4(lo.cv[4], lo.cv[5], lo.cv[6], lo.cv[7]);
            }
            const uint32_t act = __shfl(lo.active ? 1u : 0u, lo.u_head & 63);
            if (lane < WROWS) {
                S.tn[slot][lane] = (lane < nu && act) ? lo.u_cnt : 0u;
                S.toff[slot][lane] = (slot % R_GROUP) * R_SLOT_NODES + lo.u_head;
            }
            }
        } else if (lane < WROWS) {
            S.tn[slot][lane] = 0u;
        }
        lds_fence();
        const uint32_t old = uni(atomicAdd(&S.cnt[grp], lane == 0 ? 1u : 0u));
        if (old == R_GROUP - 1) {
            // this wave hashed the group's last slot: fold the four tiles' parent trees together, write the digests
            asm volatile("" ::: "memory");
            const uint32_t jj = lane >> 4, uu = lane & 15, sl = grp * R_GROUP + jj;
            FoldQueue<4> fq;
            fq.n_tab = 64;
            fq.tb_n = uu < WROWS ? S.tn[sl][uu] : 0u;
            fq.tb_off = uu < WROWS ? S.toff[sl][uu] : 0u;
            fq.tb_out = S.first[sl] + uu;
            fq.tb_root = 1;
            if (__ballot(fq.tb_n != 0) != 0ull && !(a.dbg & 131072)) {
                uint32_t *const nodes = reinterpret_cast<uint32_t *>(S.slots[grp * R_GROUP]);
                FoldQueue<4> fd = fq.dense16();
                uint32_t U = 0;
                const uint32_t n = (a.dbg & 262144) ? 0u : fd.uniform(&U);
                if (n) fd.fold_uniform_and_write(nodes, h, n, U);  // every row of the four tiles has the same length
                else fq.fold_and_write(nodes, h);
            }
            lds_fence();
            if (lane == 0) {
                lds_st(&S.cnt[grp], 0u);
                lds_st(&S.gen[grp], lds_ld(&S.gen[grp]) + 1u);
            }
        }
    }
}

__global__ __launch_bounds__(R_WAVES * 64) void k_fused_roles(FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_roles_raw[];  // dynamic: more than the 64 KiB a static array may take
    RolesShared &S = *reinterpret_cast<RolesShared *>(s_roles_raw);
    if (threadIdx.x == 0) { S.ready = 0; S.take = 0; S.finished = 0; S.next_iter = 0; }
    if (threadIdx.x < R_SLOTS / R_GROUP) { S.cnt[threadIdx.x] = 0; S.gen[threadIdx.x] = 0; }
    __syncthreads();
    const bool clk = (a.dbg & 32768) && a.dbg_buf && blockIdx.x == 7 && threadIdx.x == 64 * R_LOADERS;
    const unsigned long long c0 = clk ? __builtin_amdgcn_s_memtime() : 0, r0 = clk ? __builtin_amdgcn_s_memrealtime() : 0;
    if (threadIdx.x < 64 * R_LOADERS) {
        // the loader's instruction stream is one long dependent chain with little VALU in it: at top priority it
        // issues whenever it is ready and costs the hashers a few per cent of the issue slots
        if (!(a.dbg & 64)) __builtin_amdgcn_s_setprio(3);
        roles_loader(a, S);
    } else roles_hasher(a, S);
    if (clk) { a.dbg_buf[0] = __builtin_amdgcn_s_memtime() - c0; a.dbg_buf[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

void launch_fused_roles(const FusedArgs &a, int cus, hipStream_t s) {
    if (!a.h.n_tiles) return;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_fused_roles), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(RolesShared));
        attr_set = true;
    }
    const uint32_t want = (a.h.n_tiles + R_GROUP - 1) / R_GROUP;  // one group per workgroup at least
    const uint32_t per_cu = sizeof(RolesShared) > 80 * 1024 ? 1u : 2u;
    uint32_t grid = std::min<uint32_t>((uint32_t)cus * per_cu, want);
    if (a.lds_pad) grid = std::min<uint32_t>(grid, a.lds_pad);  // diagnostic: ZNIPPY_LDS_PAD caps the grid (one workgroup = a deterministic ring)
    hipLaunchKernelGGL(k_fused_roles, dim3(grid), dim3(R_WAVES * 64), sizeof(RolesShared), s, a);
}

// ---- big rows: block items of the common shape --------------------------------------------------------
// A frame written block by block from periodic data (a 2 GiB text file in 8 MiB or 200 MiB rounds) is thousands of
// 128 KiB blocks, each "literals + one sequence repeating a period inside them" — the shape parse_fast_block
// recognises.  One wave takes one 64 KiB slice tile of such a block: it stages the first bytes of the BLOCK, parses
// them, and hashes its 64 leaves straight from that window while storing the message registers, exactly as the
// small-row kernel does for whole rows; the slice's chaining value goes to tile_cv for k_merge_big.  A block done
// this way is skipped by the block decoder and its two tiles by the second hash pass, so the row's bytes are
// written once and never read back.  Anything else about the block (another shape, a short last block, a frame the
// scan gave up on) leaves it to those kernels, untouched.
__global__ __launch_bounds__(256, 4) void k_fused_blocks(FusedBlocksArgs a) {
    // one 9 KiB area per wave: the copy stage of a raw block's leaf loop (whole lines), and afterwards — the first
    // 2 KiB of it — the wave's 64 leaf CVs for the workgroup's fold (node index = byte offset / 32)
    __shared__ __attribute__((aligned(16))) uint8_t s_area[4][STAGE_FULL_BYTES];
    uint32_t *const s_nodes = reinterpret_cast<uint32_t *>(&s_area[0][0]);
    constexpr uint32_t AREA_NODES = STAGE_FULL_BYTES / 32;  // nodes between two waves' areas
    static_assert(STAGE_FULL_BYTES % 32 == 0, "area must be a whole number of nodes");
    __shared__ uint32_t s_tab[3][4 * FOLD_UNITS];
    __shared__ __attribute__((aligned(16))) uint8_t s_Wb[4][WSTRIDE];
    __shared__ uint16_t s_d[4][3];
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane < FOLD_UNITS) s_tab[0][w * FOLD_UNITS + lane] = 0;
    const uint32_t idx = blockIdx.x * 4 + w;
    if (idx < a.n_bt) {
        const uint32_t ti = a.bt_tile[idx], item = a.bt_item[idx];
        const Tile t = a.h.tiles[ti];
        const uint32_t row = t.first_unit;
        const uint32_t src_pos = a.item_src[item];
        const uint64_t usize = a.h.len[row];
        const uint64_t origin = (uint64_t)(t.first_leaf >> 7) << 17;
        if (src_pos != 0xFFFFFFFFu && a.row_flag[row] == 0 && t.n_leaves == 64 && usize - origin >= 128 * 1024) {
            uint8_t *const WL = s_Wb[w];
            const uint64_t n = a.blob_size[row];
            const uint8_t *const src = a.h.srcA + (a.h.offA[row] - a.h.baseA) + src_pos;
            const uint32_t avail = n - src_pos < WIN ? (uint32_t)(n - src_pos) : WIN;
            {
                const uint32_t o = 8 * lane;
                uint64_t v = 0;
                if (o + 8 <= avail) __builtin_memcpy(&v, src + o, 8);
                else
                    for (uint32_t k = 0; k < 8; k++)
                        if (o + k < avail) v |= (uint64_t)src[o + k] << (8 * k);
                *reinterpret_cast<uint64_t *>(WL + o) = v;
                if (lane < (WSTRIDE - WIN) / 4) *reinterpret_cast<uint32_t *>(WL + WIN + 4 * lane) = 0;
            }
            FastTabs T;
            {
                const DTab eL = c_dll[lane], eM = c_dml[lane], eO = c_dof[lane & 31];
                T.ll = eL.base | (uint32_t)eL.addbits << 24;
                T.ml = eM.base | (uint32_t)eM.addbits << 24;
                T.of = eO.addbits;
                T.lls = lane < 36 ? c_llb[lane] | (uint32_t)c_lla[lane] << 24 : 0u;
                T.mls = lane < 53 ? c_mlb[lane] | (uint32_t)c_mla[lane] << 24 : 0u;
            }
            const uint32_t bh = uni(*reinterpret_cast<const uint32_t *>(WL)) & 0xFFFFFF;
            if (avail >= 3 && ((bh >> 1) & 3) == 0 && (bh >> 3) == 128 * 1024 && src_pos + 3 + 128 * 1024ull <= n) {
                // raw block: its bytes sit in the frame — hash them while they are copied out (store-path loop)
                LeafOut lo;
                hash_tile_leaves<true, false, true>(a.h, t, nullptr, lo, s_area[w], src + 3 - origin);
                uint4 *d = reinterpret_cast<uint4 *>(s_nodes + (size_t)(w * AREA_NODES + lane) * 8);
                d[0] = make_uint4(lo.cv[0], lo.cv[1], lo.cv[2], lo.cv[3]);
                d[1] = make_uint4(lo.cv[4], lo.cv[5], lo.cv[6], lo.cv[7]);
                if (lane == 0) {
                    s_tab[0][w * FOLD_UNITS] = 64;
                    s_tab[1][w * FOLD_UNITS] = w * AREA_NODES;
                    s_tab[2][w * FOLD_UNITS] = t.cv_index;
                    a.tile_done[ti] = 1;
                    a.item_done[item] = 1;
                }
            }
            const FastRow fr = parse_fast_block<false>(WL, 0, avail, 128 * 1024, avail >= 12, T);  // every lane: the same block
            if (uni(fr.ok)) {
                const uint32_t lit_at = uni(fr.lit_at), L0 = uni(fr.L0), off = uni(fr.off);
                uint8_t *y = WL + lit_at;
                y[L0 + lane] = y[L0 - off + lmod(lane, off, 1.0f / (float)off)];
                if (lane == 0) { s_d[w][0] = (uint16_t)lit_at; s_d[w][1] = (uint16_t)(L0 - off); s_d[w][2] = (uint16_t)off; }
                LdsSrc ls{WL, &s_d[w][0], &s_d[w][1], &s_d[w][2], 1, nullptr, 0, 0, 0, 0, (a.dbg & 16) ? 0u : 1u, origin};
                LeafOut lo;
                hash_tile_leaves<true, true, true>(a.h, t, &ls, lo, s_area[w]);  // row stores leave as whole lines through the stage
                uint4 *d = reinterpret_cast<uint4 *>(s_nodes + (size_t)(w * AREA_NODES + lane) * 8);
                d[0] = make_uint4(lo.cv[0], lo.cv[1], lo.cv[2], lo.cv[3]);
                d[1] = make_uint4(lo.cv[4], lo.cv[5], lo.cv[6], lo.cv[7]);
                if (lane == 0) {
                    s_tab[0][w * FOLD_UNITS] = 64;
                    s_tab[1][w * FOLD_UNITS] = w * AREA_NODES;
                    s_tab[2][w * FOLD_UNITS] = t.cv_index;
                    a.tile_done[ti] = 1;
                    a.item_done[item] = 1;  // the block's other tile comes to the same verdict (same block, same test)
                }
            }
        }
    }
    lds_barrier();
    if (w != (blockIdx.x * 2654435761u) >> 30) return;
    FoldQueue<4> fq;
    fq.n_tab = 4 * FOLD_UNITS;
    fq.tb_n = s_tab[0][lane];
    fq.tb_off = s_tab[1][lane];
    fq.tb_out = s_tab[2][lane];
    fq.tb_root = 0;  // slice CVs: the merge kernel finishes the row
    if (__ballot(fq.tb_n != 0) == 0ull) return;
    fq.fold_and_write(s_nodes, a.h);
}

void launch_fused_blocks(const FusedBlocksArgs &a, hipStream_t s) {
    if (!a.n_bt) return;
    hipLaunchKernelGGL(k_fused_blocks, dim3((a.n_bt + 3) / 4), dim3(256), 0, s, a);
}

void set_fused_dbg(unsigned long long *) {}
void set_fused_abl(int v) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_abl), &v, sizeof v); }

void launch_fused_small(const FusedArgs &a, hipStream_t s, int grid_cap) {
    if (!a.h.n_tiles) return;
    uint32_t grid = (a.h.n_tiles + 3) / 4;
    if (grid_cap > 0 && grid > (uint32_t)grid_cap) grid = (uint32_t)grid_cap;
    hipLaunchKernelGGL(k_fused_small, dim3(grid), dim3(256), a.lds_pad, s, a);
}

}  // namespace zn
