// Device helpers shared by the zstd decode kernels (zstd_decode.hip, zstd_batch.hip): constants and code tables of
// RFC 8878, bit readers, wave-cooperative copies, the LDS output window.  Internal.
#pragma once
#include "common.h"

namespace zn {

constexpr int E_TRUNC = -5, E_CORRUPT = -5, E_UNSUP = -6, E_DST = -4;
constexpr uint32_t BLOCK_MAX = 128 * 1024;
constexpr int LIT_SCRATCH_BYTES = 128 * 1024 + 64;
constexpr int SEQ_BATCH = 256;
constexpr uint32_t BIG_COPY = 8192;  // copies at least this long are shared by all waves of the workgroup
constexpr uint32_t EXP_OFF_MAX = 4096;  // longest period expanded through LDS
// Narrow variant (4 waves per frame: many small/medium frames): output window in LDS (aliases the pattern
// buffer): [history | chunk].  Short sequences are executed inside it — literals and matches of up to 64
// sequences at a time, one per lane — and the chunk is streamed to HBM when full, so a match never waits for
// the store queue.  Sequences longer than WIN_SEQ_MAX go straight to HBM, as everything does in the wide variant.
constexpr uint32_t WIN_HIST = 4096, WIN_CAP = 8192, WIN_SEQ_MAX = 2048;

struct FseEntry {
    uint16_t next;    // new-state base
    uint8_t nbits;    // bits to read for the state update
    uint8_t addbits;  // extra bits of the value
    uint32_t base;    // value baseline
};

static __constant__ uint32_t c_ll_base[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18,
                                       20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512, 1024, 2048,
                                       4096, 8192, 16384, 32768, 65536};
static __constant__ uint8_t c_ll_bits[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1,
                                      1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
static __constant__ uint32_t c_ml_base[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20,
                                       21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 37,
                                       39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051,
                                       4099, 8195, 16387, 32771, 65539};
static __constant__ uint8_t c_ml_bits[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                      0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1,
                                      1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11,
                                      12, 13, 14, 15, 16};
static __constant__ int8_t c_ll_default[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2,
                                        2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
static __constant__ int8_t c_ml_default[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                        1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                        1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
static __constant__ int8_t c_of_default[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1,
                                        1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};

enum { K_LL = 0, K_OF = 1, K_ML = 2 };

// NW = waves of the workgroup: sizes the pattern buffer (16 bytes per thread per trip)
template <int NW>
struct SharedT {
    FseEntry ll[512], ml[512], of[256];  // tables described in the current/previous block
    FseEntry dll[64], dml[64], dof[32];  // predefined tables (built once per workgroup)
    FseEntry rle[3];
    uint16_t huf[2048];                  // symbol | nbits << 8
    uint8_t weights[256];
    uint16_t sym_start[256];
    uint16_t sym_len[256];
    int16_t norm[256];
    uint16_t fse_next[256];
    uint8_t fse_sym[512];
    uint32_t seq_ll[2][SEQ_BATCH], seq_ml[2][SEQ_BATCH], seq_off[2][SEQ_BATCH];  // two batches: one being decoded, one being executed
    // pattern buffer for long overlapping matches: E[i] = period[i % off], i < off + 16 * threads
    __attribute__((aligned(16))) uint8_t ebuf[(NW == 4 ? WIN_HIST + WIN_CAP : EXP_OFF_MAX + 16 * 64 * NW) + 64];
    // per-row / per-block state broadcast from lane 0
    int32_t err;
    uint32_t row, skip, claim;
    uint32_t blk_type, blk_size, blk_last;
    uint32_t lit_kind;  // 0 raw (pointer into src), 1 rle, 2 scratch
    uint32_t lit_len, lit_rle;
    uint64_t lit_src;  // offset in blob (raw) -- relative to frame src
    uint32_t huf_log, huf_valid;
    uint32_t n_streams, stream_off[4], stream_len[4], stream_out[4], stream_n[4];
    uint32_t sel[3], log_[3], valid[3];
    uint32_t tree_off, tree_n;  // a Huffman tree description waits at bsrc + tree_off (tree_n bytes available; 0 = none): read by wave 0 after lane 0's header pass
    uint32_t bld[3];  // tables described in this block: alphabet size, 0 = nothing to build (built by waves 0..2 after lane 0 has read the counts)
    uint32_t nseq, batch_n;
    uint64_t src_pos, src_end;  // byte offsets inside the frame's blob
    uint64_t out_pos, out_end;  // byte offsets inside the row's output (out_pos: flushed to HBM)
    uint64_t blk_base;          // block items: first output byte of the item (matches may not reach in front of it)
    uint32_t win_n, hist_n;     // narrow variant: bytes waiting in the window's chunk part / valid history bytes in front of it
    uint32_t lit_pos;
    uint32_t rep[3];
    uint32_t has_cksum;
    uint64_t content_size;
    // sequence bitstream state kept by lane 0 across batches
    int64_t bs_pos;
    uint32_t bs_off, st_ll, st_of, st_ml;
};

__device__ __forceinline__ int hibit(uint32_t v) { return 31 - __clz(v); }
// wave-uniform value -> SGPR: arithmetic on it runs on the scalar unit, branches on it are scalar branches
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t uni64(uint64_t v) { return ((uint64_t)uni((uint32_t)(v >> 32)) << 32) | uni((uint32_t)v); }

// v_readlane with a wave-uniform (not necessarily constant) lane index: one instruction, where __shfl(v, j) is an LDS
// crossbar round trip (ds_bpermute, ~100+ cycles)
__device__ __forceinline__ uint32_t rdlane_u(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
// Inclusive prefix sum over the 64 lanes on the DPP path (row shifts inside the 16-lane rows, then the two row
// broadcasts gfx9 has): 6 VALU moves + 6 adds.  The shuffle version (6 x __shfl_up) is 6 LDS crossbar round trips —
// it was 1,500 cycles of every 64-sequence group.
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);  // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    return v;
}

// lane l (uniform) of four registers replaced by four uniform values: four v_writelane.  (Round 2 wrote M0 by hand in
// inline asm to share one lane select between the four; M0 is a reserved register the compiler does not expect an asm
// statement to clobber, so the compiler now places the lane select itself: this clang has no __builtin_amdgcn_writelane,
// the LLVM intrinsic is reached by its name.)
extern "C" __device__ int zn_writelane(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");
__device__ __forceinline__ void wrlane4_u(uint32_t &v0, uint32_t &v1, uint32_t &v2, uint32_t &v3, uint32_t x0, uint32_t x1, uint32_t x2,
                                          uint32_t x3, uint32_t l) {
    v0 = (uint32_t)zn_writelane((int)x0, (int)l, (int)v0);
    v1 = (uint32_t)zn_writelane((int)x1, (int)l, (int)v1);
    v2 = (uint32_t)zn_writelane((int)x2, (int)l, (int)v2);
    v3 = (uint32_t)zn_writelane((int)x3, (int)l, (int)v3);
}

__device__ __forceinline__ uint64_t rdlane64_u(uint64_t v, uint32_t l) {
    return ((uint64_t)rdlane_u((uint32_t)(v >> 32), l) << 32) | rdlane_u((uint32_t)v, l);
}

__device__ __forceinline__ uint64_t load8_guard(const uint8_t *p, const uint8_t *end) {
    if (p + 8 <= end) {
        uint64_t v;
        __builtin_memcpy(&v, p, 8);
        return v;
    }
    uint64_t v = 0;
    for (int i = 0; i < 8 && p + i < end; i++) v |= (uint64_t)p[i] << (8 * i);
    return v;
}

// Backward bit reader over [base, base+n): `pos` = unread bits, reads return the bits just
// below pos, zero-filled below bit 0 (RFC 8878 §4.1).
struct BitR {
    const uint8_t *base, *end;
    int64_t pos;
    uint64_t win;
    int64_t wbase;
    __device__ __forceinline__ void refill() {
        int64_t b0 = ((pos + 7) >> 3) - 8;
        if (b0 < 0) b0 = 0;
        win = load8_guard(base + b0, end);
        wbase = b0 * 8;
    }
    __device__ __forceinline__ bool init(const uint8_t *p, uint32_t n, const uint8_t *blob_end) {
        if (n == 0) return false;
        uint8_t last = p[n - 1];
        if (last == 0) return false;
        base = p; end = blob_end;
        pos = (int64_t)n * 8 - (8 - hibit(last));
        refill();
        return true;
    }
    __device__ __forceinline__ uint32_t peek(uint32_t nb) {  // nb <= 32
        if (nb == 0) return 0;
        int64_t s = pos - wbase - (int64_t)nb;
        if (s < 0 && wbase > 0) { refill(); s = pos - wbase - (int64_t)nb; }
        uint64_t v = s >= 0 ? (win >> s) : (s > -64 ? (win << (-s)) : 0ull);
        if (pos < (int64_t)nb) {
            // bits below 0 read as zero: keep only the top `pos` real bits
            if (pos <= 0) return 0;
            uint64_t real = win & ((pos >= 64) ? ~0ull : ((1ull << pos) - 1));  // wbase == 0 here
            v = real << ((int64_t)nb - pos);
        }
        return (uint32_t)(v & ((nb >= 32) ? 0xFFFFFFFFull : ((1ull << nb) - 1)));
    }
    __device__ __forceinline__ uint32_t read(uint32_t nb) {
        uint32_t v = peek(nb);
        pos -= nb;
        return v;
    }
};

// Forward bit reader (FSE table descriptions).
struct FwdR {
    const uint8_t *p;
    uint32_t n;
    uint32_t bitpos;
    __device__ __forceinline__ uint32_t read(uint32_t nb) {
        uint32_t v = 0;
        for (uint32_t i = 0; i < nb; i++) {
            uint32_t bp = bitpos + i;
            uint32_t bit = (bp >> 3) < n ? (p[bp >> 3] >> (bp & 7)) & 1u : 0u;
            v |= bit << i;
        }
        bitpos += nb;
        return v;
    }
};

// ---------------------------------------------------------------------------------------------
// cooperative copies.  `nthreads` lanes (tid in [0,nthreads)) move n bytes src -> dst; ranges
// do not overlap.  Long copies go 16 B per lane with the DESTINATION aligned (coalesced 1 KiB
// per wave-instruction); unaligned sources use the hardware's unaligned dwordx4 loads.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void coop_copy(uint8_t *dst, const uint8_t *src, uint64_t n, uint32_t tid,
                                          uint32_t nthreads) {
    if (n < 256) {
        for (uint64_t i = tid; i < n; i += nthreads) dst[i] = src[i];
        return;
    }
    uint32_t head = (uint32_t)((16 - ((uintptr_t)dst & 15)) & 15);
    if (tid < head) dst[tid] = src[tid];
    uint64_t body = (n - head) >> 4;
    uint8_t *d = dst + head;
    const uint8_t *s = src + head;
    for (uint64_t i = tid; i < body; i += nthreads) {
        uint4 v;
        __builtin_memcpy(&v, s + i * 16, 16);
        *reinterpret_cast<uint4 *>(d + i * 16) = v;
    }
    uint64_t done = head + body * 16;
    if (done + tid < n) dst[done + tid] = src[done + tid];  // tail < 16 bytes
}

__device__ __forceinline__ void coop_fill(uint8_t *dst, uint8_t byte, uint64_t n, uint32_t tid, uint32_t nthreads) {
    if (n < 256) {
        for (uint64_t i = tid; i < n; i += nthreads) dst[i] = byte;
        return;
    }
    uint32_t head = (uint32_t)((16 - ((uintptr_t)dst & 15)) & 15);
    if (tid < head) dst[tid] = byte;
    uint64_t body = (n - head) >> 4;
    uint32_t w = byte * 0x01010101u;
    uint4 v = make_uint4(w, w, w, w);
    uint8_t *d = dst + head;
    for (uint64_t i = tid; i < body; i += nthreads) *reinterpret_cast<uint4 *>(d + i * 16) = v;
    uint64_t done = head + body * 16;
    if (done + tid < n) dst[done + tid] = byte;
}

// stores of this wave complete (and visible to its later loads) before continuing
__device__ __forceinline__ void wave_mem_sync() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// LZ match: dst[i] = dst[i - off] for i in [0, n), executed by `nthreads` lanes.  An overlapping
// match (off < n) is a periodic extension; it is laid down by period doubling so that every step
// is a non-overlapping cooperative copy whose source is already final.
// Long overlapping match by the whole workgroup: expand the period once into LDS
// (E[i] = period[i % off] for i < off + 16*NT, built by tail-free doubling), then every trip all NT
// threads store 16 aligned bytes each (NT*16 contiguous bytes per trip) — no global read-after-write,
// one barrier per doubling step instead of one store drain + barrier per step through memory.
template <int NWAVES>
__device__ __forceinline__ void wg_expand_match(uint8_t *dst, uint32_t off, uint64_t n, uint32_t tid, uint8_t *E) {
    constexpr uint32_t NT = NWAVES * 64, CHUNK = NT * 16;
    const uint8_t *pat = dst - off;  // final: the caller's barrier drained every earlier store
    {
        const uint32_t full = off & ~15u;
        for (uint32_t i = tid * 16; i < full; i += CHUNK) {
            uint4 v;
            __builtin_memcpy(&v, pat + i, 16);
            __builtin_memcpy(E + i, &v, 16);
        }
        if (tid < (off & 15)) E[full + tid] = pat[full + tid];
    }
    __syncthreads();
    const uint32_t need = off + CHUNK;
    uint32_t w = off;
    while (w < need) {
        const uint32_t c = w < need - w ? w : need - w;
        for (uint32_t i = tid * 16; i < c; i += CHUNK) {
            uint4 v;
            __builtin_memcpy(&v, E + i, 16);
            __builtin_memcpy(E + w + i, &v, 16);  // may spill < 16 bytes past c: rewritten by the next step / slack
        }
        w += c;
        __syncthreads();
    }
    const uint32_t head = (uint32_t)((16 - ((uintptr_t)dst & 15)) & 15);
    if (tid < head) dst[tid] = E[tid];
    uint64_t x = head;
    uint32_t s = head % off;
    const uint32_t step = CHUNK % off;
    uint8_t *d = dst + head + 16 * tid;
    while (x + CHUNK <= n) {
        uint4 v;
        __builtin_memcpy(&v, E + s + 16 * tid, 16);
        *reinterpret_cast<uint4 *>(d) = v;
        d += CHUNK;
        x += CHUNK;
        s += step;
        if (s >= off) s -= off;
    }
    const uint32_t rem = (uint32_t)(n - x), full16 = rem >> 4;
    if (tid < full16) {
        uint4 v;
        __builtin_memcpy(&v, E + s + 16 * tid, 16);
        *reinterpret_cast<uint4 *>(d) = v;
    }
    const uint32_t tail = rem & 15, tbase = full16 * 16;
    if (tid < tail) dst[x + tbase + tid] = E[s + tbase + tid];
}

template <int NWAVES>
__device__ __forceinline__ void coop_match(uint8_t *dst, uint32_t off, uint64_t n, uint32_t tid, bool all_waves,
                                           uint8_t *E) {
    const uint32_t nthreads = all_waves ? NWAVES * 64 : 64;
    if (off >= n) {
        coop_copy(dst, dst - off, n, tid, nthreads);
        return;
    }
    if (all_waves && NWAVES > 1 && off <= EXP_OFF_MAX && n >= 4ull * NWAVES * 64 * 16) {
        wg_expand_match<NWAVES>(dst, off, n, tid, E);
        return;
    }
    uint64_t w = 0;
    if (off < 64) {
        // seed: the first bytes of the run straight from the period
        uint64_t c = n < 64 ? n : 64;
        if (tid < c) dst[tid] = dst[(int64_t)(tid % off) - (int64_t)off];
        w = c;
        if (all_waves && NWAVES > 1) __syncthreads(); else wave_mem_sync();
    }
    while (w < n) {
        uint64_t avail = w + off;
        uint64_t pm = (avail / off) * off;  // largest multiple of the period already laid down
        uint64_t c = n - w < pm ? n - w : pm;
        coop_copy(dst + w, dst + w - pm, c, tid, nthreads);
        w += c;
        if (w < n) {
            if (all_waves && NWAVES > 1) __syncthreads(); else wave_mem_sync();
        }
    }
}

// ---------------------------------------------------------------------------------------------
// LDS output window (wave 0).  W = S.ebuf; the chunk starts at W + WIN_HIST and stands for output
// bytes [chunk_abs, chunk_abs + win_n); `hist_n` bytes in front of it are the output just before.
// ---------------------------------------------------------------------------------------------
// Per-lane copies.  A wave-wide call costs what its slowest lane costs, and a lane's loop of "load, wait, store" pays
// one memory round trip per trip (1-2 us from HBM, ~100 cycles from LDS): a 15-byte tail copied byte by byte from
// global memory held 64 sequences up for 15 round trips.  So: every load of a stretch is issued before its first
// store (64 bytes per trip), and a tail is 8 + 4 + 2 + 1 bytes (or one 16-byte piece that re-copies bytes already
// done, where source and destination are disjoint) instead of a byte loop.  None of them writes past n.
// The window is addressed as LDS (address space 3), not through generic pointers: a generic access is a FLAT
// instruction, which resolves its aperture in the texture path (~500 cycles per dependent read -> write step, and it
// ties LDS traffic to vmcnt); ds_read / ds_write take ~100.  gfx950 executes ds_*_b64 / b128 at any alignment.
typedef __attribute__((address_space(3))) uint8_t lds8;
#define LDS_CP(dst, src, n) __builtin_memcpy((__attribute__((address_space(3))) void *)(dst), (src), (n))
#define LDS_LD(dst, src, n) __builtin_memcpy((dst), (const __attribute__((address_space(3))) void *)(src), (n))

__device__ __forceinline__ void lane_tail_copy(lds8 *d, const uint8_t *g, uint32_t n) {  // n < 16, pieces in ascending order
    uint64_t a = 0; uint32_t b = 0; uint16_t c = 0; uint8_t e = 0;
    const uint32_t k4 = (n & 8), k2 = (n & 8) + (n & 4), k1 = (n & 8) + (n & 4) + (n & 2);
    if (n & 8) __builtin_memcpy(&a, g, 8);
    if (n & 4) __builtin_memcpy(&b, g + k4, 4);
    if (n & 2) __builtin_memcpy(&c, g + k2, 2);
    if (n & 1) e = g[k1];
    if (n & 8) LDS_CP(d, &a, 8);
    if (n & 4) LDS_CP(d + k4, &b, 4);
    if (n & 2) LDS_CP(d + k2, &c, 2);
    if (n & 1) d[k1] = e;
}
// global -> LDS (disjoint by construction)
__device__ __forceinline__ void lane_copy_g2l(lds8 *d, const uint8_t *g, uint32_t n) {
    uint32_t k = 0;
    for (; k + 64 <= n; k += 64) {
        uint4 v0, v1, v2, v3;
        __builtin_memcpy(&v0, g + k, 16); __builtin_memcpy(&v1, g + k + 16, 16);
        __builtin_memcpy(&v2, g + k + 32, 16); __builtin_memcpy(&v3, g + k + 48, 16);
        LDS_CP(d + k, &v0, 16); LDS_CP(d + k + 16, &v1, 16);
        LDS_CP(d + k + 32, &v2, 16); LDS_CP(d + k + 48, &v3, 16);
    }
    const uint32_t rem = n - k;  // < 64
    if (n >= 16) {
        // up to three whole pieces + one piece that ends exactly at n (it overlaps the one before: same bytes again)
        uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0, v2 = v0, vl;
        const uint32_t c16 = rem >> 4;
        if (c16 > 0) __builtin_memcpy(&v0, g + k, 16);
        if (c16 > 1) __builtin_memcpy(&v1, g + k + 16, 16);
        if (c16 > 2) __builtin_memcpy(&v2, g + k + 32, 16);
        __builtin_memcpy(&vl, g + n - 16, 16);
        if (c16 > 0) LDS_CP(d + k, &v0, 16);
        if (c16 > 1) LDS_CP(d + k + 16, &v1, 16);
        if (c16 > 2) LDS_CP(d + k + 32, &v2, 16);
        if (rem & 15) LDS_CP(d + n - 16, &vl, 16);
    } else lane_tail_copy(d, g, n);
}
// the first n (< 16) bytes of a 16-byte register value, as 8 + 4 + 2 + 1 byte stores
__device__ __forceinline__ void store_prefix16(lds8 *d, const uint4 v, uint32_t n) {
    uint64_t cur = (uint64_t)v.x | ((uint64_t)v.y << 32);
    if (n & 8) { LDS_CP(d, &cur, 8); d += 8; cur = (uint64_t)v.z | ((uint64_t)v.w << 32); }
    if (n & 4) { const uint32_t x = (uint32_t)cur; LDS_CP(d, &x, 4); d += 4; cur >>= 32; }
    if (n & 2) { const uint16_t x = (uint16_t)cur; LDS_CP(d, &x, 2); d += 2; cur >>= 16; }
    if (n & 1) *d = (uint8_t)cur;
}
// forward copy inside LDS, d = s + dist with dist >= 16 or dist >= n (ranges may overlap: whole 16-byte pieces go in
// ascending order — LDS operations of a wave execute in order).  The last n % 16 bytes are ONE 16-byte read (it may
// run past the source into bytes nobody uses; the caller's buffer has the slack) and up to four register stores: a
// short match — most of them — is a single LDS round trip.
__device__ __forceinline__ void lane_copy_l2l(lds8 *d, const lds8 *s, uint32_t n) {
    uint32_t k = 0;
    for (; k + 16 <= n; k += 16) {
        uint4 v;
        LDS_LD(&v, s + k, 16);
        LDS_CP(d + k, &v, 16);
    }
    if (k < n) {
        uint4 v;
        LDS_LD(&v, s + k, 16);
        store_prefix16(d + k, v, n - k);
    }
}
// overlapping match with a period below 16 (off < n): 8 <= off: 8-byte pieces are still a valid forward copy;
// off < 8: the period is widened to 8 bytes in a register once and written out with a rotating phase
__device__ __forceinline__ void lane_copy_period(lds8 *d, const lds8 *s, uint32_t n, uint32_t off) {
    uint32_t k = 0;
    if (off >= 8) {
        for (; k + 8 <= n; k += 8) { uint64_t v; LDS_LD(&v, s + k, 8); LDS_CP(d + k, &v, 8); }
        if (k < n) {  // < 8 bytes left, the distance is >= 8: one read, register stores
            uint64_t cur;
            LDS_LD(&cur, s + k, 8);
            const uint32_t r = n - k;
            if (r & 4) { const uint32_t x = (uint32_t)cur; LDS_CP(d + k, &x, 4); k += 4; cur >>= 32; }
            if (r & 2) { const uint16_t x = (uint16_t)cur; LDS_CP(d + k, &x, 2); k += 2; cur >>= 16; }
            if (r & 1) d[k] = (uint8_t)cur;
        }
        return;
    }
    uint64_t m;
    LDS_LD(&m, s, 8);  // the period is the first off (<= 7) bytes of this
    m &= (1ull << (8 * off)) - 1;
    uint64_t r = m;
    for (uint32_t w = off; w < 8; w *= 2) r |= r << (8 * w);  // r[i] = m[i % off], i < 8
    const uint32_t step = 8 % off;
    uint32_t ph = 0;  // phase of the piece: its first byte is m[ph]
    for (; k + 8 <= n; k += 8) {
        const uint64_t v = ph ? (r >> (8 * ph)) | (r << (8 * (off - ph))) : r;
        LDS_CP(d + k, &v, 8);
        ph += step;
        if (ph >= off) ph -= off;
    }
    const uint64_t v = ph ? (r >> (8 * ph)) | (r << (8 * (off - ph))) : r;
    const uint32_t rr = n - k;
    uint32_t sh = 0;
    if (rr & 4) { const uint32_t x = (uint32_t)(v >> sh); LDS_CP(d + k, &x, 4); k += 4; sh += 32; }
    if (rr & 2) { const uint16_t x = (uint16_t)(v >> sh); LDS_CP(d + k, &x, 2); k += 2; sh += 16; }
    if (rr & 1) d[k] = (uint8_t)(v >> sh);
}

// One sequence per lane (lanes with on == false idle): literals, then matches in dependency rounds — a
// match is copied once every byte of its source is final, i.e. lies before the destination of the first
// match still pending (the high-water mark); the first pending match is always ready (its own overlap is
// a forward copy).  Text needs 1-3 rounds per 64 sequences.
struct ExecProf { unsigned long long t_lits = 0, t_match = 0, rounds = 0; };
__device__ __forceinline__ void win_exec_group(uint8_t *Wg, const uint8_t *out, uint64_t chunk_abs, uint32_t hist_n, uint32_t lane,
                                               bool on, uint32_t dpos, uint32_t ll, uint32_t ml, uint32_t off, const uint8_t *lit,
                                               bool rle, uint8_t rle_byte, ExecProf *prof = nullptr) {
    lds8 *const W = (lds8 *)Wg;
    unsigned long long t0 = prof ? __builtin_amdgcn_s_memtime() : 0;
    // The callers fill parts of the window through generic pointers (cooperative copies: FLAT stores, which reach the
    // LDS through the texture path).  A ds_read issued after such a store can overtake it: drain them first.
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    // A wave-wide call costs what its slowest lane costs: a lane copies up to LANE_MAX bytes itself, anything longer
    // is moved by all 64 lanes together (16 bytes each per step) once the short ones are done.
    constexpr uint32_t LANE_MAX = 64;
    if (on && ll && ll <= LANE_MAX) {
        lds8 *d = W + dpos;
        if (rle) for (uint32_t k = 0; k < ll; k++) d[k] = rle_byte;
        else lane_copy_g2l(d, lit, ll);
    }
    for (uint64_t lm = __ballot(on && ll > LANE_MAX); lm; lm &= lm - 1) {
        const uint32_t j = (uint32_t)__ffsll((long long)lm) - 1;
        const uint32_t dj = rdlane_u(dpos, j), nj = rdlane_u(ll, j);
        const uint64_t lj = ((uint64_t)rdlane_u((uint32_t)((uintptr_t)lit >> 32), j) << 32) | rdlane_u((uint32_t)(uintptr_t)lit, j);
        if (rle) for (uint32_t k = lane; k < nj; k += 64) W[dj + k] = rle_byte;
        else {
            coop_copy(Wg + dj, reinterpret_cast<const uint8_t *>(lj), nj, lane, 64);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // FLAT stores into the window: done before any ds_read below
        }
    }
    const uint32_t mdst = dpos + ll;
    const int32_t msrc = (int32_t)mdst - (int32_t)off;  // window coordinate of the match source (may lie before the history)
    const int32_t lds_lo = (int32_t)WIN_HIST - (int32_t)hist_n;
    bool pend = on && ml != 0;
    if (prof) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t1 = __builtin_amdgcn_s_memtime(); prof->t_lits += t1 - t0; t0 = t1; }
    for (;;) {
        __builtin_amdgcn_wave_barrier();
        const uint64_t pm = __ballot(pend);
        if (!pm) break;
        if (prof) prof->rounds++;
        const uint32_t first = (uint32_t)__ffsll((long long)pm) - 1;
        const uint32_t hwm = rdlane_u(mdst, first);
        const bool ready = pend && (lane == first || msrc + (int32_t)ml <= (int32_t)hwm);
        // The common round (text: nearly all of them): every ready match is at most 16 bytes, sits in the window and
        // does not overlap itself -> one 16-byte read and a prefix store per lane.  A single wave has nobody to hide
        // its instruction latency behind, so a round costs what its instruction COUNT costs.
        if (__ballot(ready && !(ml <= 16 && msrc >= lds_lo && off >= ml)) == 0ull) {
            if (ready) {
                uint4 v;
                LDS_LD(&v, W + msrc, 16);
                if (ml == 16) LDS_CP(W + mdst, &v, 16);
                else store_prefix16(W + mdst, v, ml);
                pend = false;
            }
            continue;
        }
        if (ready && ml <= LANE_MAX) {
            uint32_t k = 0;
            if (msrc < lds_lo) {  // (part of) the source was flushed long ago: read it back from HBM
                const uint32_t nf = (uint32_t)(lds_lo - msrc) < ml ? (uint32_t)(lds_lo - msrc) : ml;
                const uint8_t *g = out + (chunk_abs - WIN_HIST) + (int64_t)msrc;
                lane_copy_g2l(W + mdst, g, nf);
                k = nf;
            }
            if (k < ml) {
                lds8 *d = W + mdst + k;
                const lds8 *sp = W + (msrc + (int32_t)k);
                const uint32_t n = ml - k;
                if (off >= 16 || off >= ml) lane_copy_l2l(d, sp, n);
                else if (k == 0) lane_copy_period(d, sp, n, off);  // short period
                else for (uint32_t i = 0; i < n; i++) d[i] = sp[i];  // short period whose first bytes came from HBM (cannot happen: off < 16 lies inside the history)
            }
        }
        // long matches that are ready, one after the other, 64 lanes each
        for (uint64_t lm = __ballot(ready && ml > LANE_MAX); lm; lm &= lm - 1) {
            const uint32_t j = (uint32_t)__ffsll((long long)lm) - 1;
            const uint32_t dj = rdlane_u(mdst, j), nj = rdlane_u(ml, j), oj = rdlane_u(off, j);
            const int32_t sj = (int32_t)rdlane_u((uint32_t)msrc, j);
            uint32_t k = 0;
            if (sj < lds_lo) {
                const uint32_t nf = (uint32_t)(lds_lo - sj) < nj ? (uint32_t)(lds_lo - sj) : nj;
                coop_copy(Wg + dj, out + (chunk_abs - WIN_HIST) + (int64_t)sj, nf, lane, 64);
                k = nf;
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // FLAT stores into the window, see above
            }
            if (k < nj && oj < 16) {  // short period: written from a register by one lane (no reads to wait for)
                if (lane == 0) lane_copy_period(W + dj + k, W + (sj + (int32_t)k), nj - k, oj);
                k = nj;
            }
            while (k < nj) {  // forward copy in steps no longer than the distance: a step's source is final when it starts
                const uint32_t lim = oj < 1024 ? oj : 1024u;
                const uint32_t step = nj - k < lim ? nj - k : lim;
                const uint32_t i = lane * 16;
                if (i < step) {
                    uint4 v;
                    LDS_LD(&v, W + (sj + (int32_t)(k + i)), 16);
                    if (step - i >= 16) LDS_CP(W + dj + k + i, &v, 16);
                    else store_prefix16(W + dj + k + i, v, step - i);
                }
                k += step;
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (ready) pend = false;
    }
    if (prof) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); prof->t_match += __builtin_amdgcn_s_memtime() - t0; }
}

// chunk -> HBM, then keep the newest bytes as history.  Returns the new history length.
__device__ __forceinline__ uint32_t win_flush(uint8_t *W, uint8_t *out, uint64_t chunk_abs, uint32_t win_n, uint32_t hist_n, uint32_t lane,
                                             bool keep_history) {
    if (win_n) coop_copy(out + chunk_abs, W + WIN_HIST, win_n, lane, 64);
    uint32_t h = 0;
    if (keep_history) {
        h = hist_n + win_n < WIN_HIST ? hist_n + win_n : WIN_HIST;
        const uint8_t *sp = W + WIN_HIST + win_n - h;
        uint8_t *dp = W + WIN_HIST - h;
        if (win_n)
            for (uint32_t base = 0; base < h; base += 1024) {  // moves down by win_n: ascending 1 KiB steps, read then write
                const uint32_t i = base + lane * 16;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (i < h) __builtin_memcpy(&v, sp + i, 16);
                __builtin_amdgcn_wave_barrier();
                if (i < h) __builtin_memcpy(dp + i, &v, 16);
                __builtin_amdgcn_wave_barrier();
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the chunk has landed: later far matches of this wave may read it back
    return h;
}



struct LitHdr { uint32_t type, regen, comp, hdr, streams; };

// Literals_Section_Header (RFC 8878 §3.1.1.3.1.1) of the block content [b, b + n)
__device__ int fz_lit_header(const uint8_t *b, uint32_t n, LitHdr &h) {
    if (n < 1) return E_TRUNC;
    const uint32_t b0 = b[0], sf = (b0 >> 2) & 3;
    h.type = b0 & 3; h.comp = 0; h.streams = 0;
    if (h.type <= 1) {
        if ((sf & 1) == 0) { h.regen = b0 >> 3; h.hdr = 1; }
        else if (sf == 1) { if (n < 2) return E_TRUNC; h.regen = (b0 >> 4) + ((uint32_t)b[1] << 4); h.hdr = 2; }
        else { if (n < 3) return E_TRUNC; h.regen = (b0 >> 4) + ((uint32_t)b[1] << 4) + ((uint32_t)b[2] << 12); h.hdr = 3; }
        if (h.regen > BLOCK_MAX) return E_CORRUPT;
        if (h.hdr + (h.type == 0 ? h.regen : 1u) > n) return E_TRUNC;
        return 0;
    }
    uint64_t v = 0;
    for (uint32_t i = 0; i < 5 && i < n; i++) v |= (uint64_t)b[i] << (8 * i);
    if (sf == 0) { h.streams = 1; h.regen = (v >> 4) & 0x3FF; h.comp = (v >> 14) & 0x3FF; h.hdr = 3; }
    else if (sf == 1) { h.streams = 4; h.regen = (v >> 4) & 0x3FF; h.comp = (v >> 14) & 0x3FF; h.hdr = 3; }
    else if (sf == 2) { h.streams = 4; h.regen = (v >> 4) & 0x3FFF; h.comp = (v >> 18) & 0x3FFF; h.hdr = 4; }
    else { h.streams = 4; h.regen = (v >> 4) & 0x3FFFF; h.comp = (v >> 22) & 0x3FFFF; h.hdr = 5; }
    if (h.hdr + h.comp > n) return E_TRUNC;
    if (h.regen > BLOCK_MAX) return E_CORRUPT;
    return 0;
}
__device__ __forceinline__ uint32_t fz_lit_section_bytes(const LitHdr &h) {
    return h.hdr + (h.type == 0 ? h.regen : (h.type == 1 ? 1u : h.comp));
}

}  // namespace zn
