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
constexpr uint32_t WIN_SCRATCH = 512;  // bytes behind the window (history + chunk + 64): win_exec_group's search arrays

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
    __attribute__((aligned(16))) uint8_t ebuf[(NW == 4 ? WIN_HIST + WIN_CAP + WIN_SCRATCH : EXP_OFF_MAX + 16 * 64 * NW) + 64];
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
// (the same as functions: inside a template the builtin with an address-space pointer resolves to the host's memcpy)
__device__ __forceinline__ uint4 lds_ld16(const lds8 *s) { uint4 v; LDS_LD(&v, s, 16); return v; }
__device__ __forceinline__ void lds_st16(lds8 *d, const uint4 v) { LDS_CP(d, &v, 16); }

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

// One sequence per lane (lanes with on == false idle): literals, then matches in dependency rounds — a match is copied
// once every byte of its source is final, i.e. once no PENDING match of an earlier lane writes into its source range.
// The lanes' match destinations are disjoint and ascending, so the lanes a match depends on are an index range found by
// two binary searches over the destinations' starts and ends (512 bytes of LDS scratch behind the window), once per
// group; a round is then one ballot and one mask test per lane.  (Round 2 compared every source against the first
// pending match's destination: 12.7 rounds per group of real text, ~4 with the exact test.)
struct ExecProf { unsigned long long t_lits = 0, t_match = 0, rounds = 0; };
template <uint32_t WH = WIN_HIST, uint32_t WC = WIN_CAP>
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
    const int32_t lds_lo = (int32_t)WH - (int32_t)hist_n;
    bool pend = on && ml != 0;
    // the earlier lanes whose match destination [mdst_j, mdst_j + ml_j) meets this lane's source [msrc, min(msrc + ml, mdst))
    uint64_t dep = 0;
    {
        typedef __attribute__((address_space(3))) int32_t lds32;
        lds32 *const starts = (lds32 *)(W + WH + WC + 64), *const ends = starts + 64;
        starts[lane] = on ? (int32_t)mdst : INT32_MAX;           // ascending over the lanes (idle lanes: never met)
        ends[lane] = on ? (int32_t)(mdst + ml) : INT32_MAX;
        __builtin_amdgcn_wave_barrier();
        const int32_t s_lo = msrc, s_hi = (int32_t)mdst < msrc + (int32_t)ml ? (int32_t)mdst : msrc + (int32_t)ml;
        uint32_t j0 = 0, j1 = 0;  // lanes whose destination ends at or before s_lo / starts before s_hi
#pragma unroll
        for (uint32_t st = 32; st; st >>= 1) {
            if (ends[j0 + st - 1] <= s_lo) j0 += st;
            if (starts[j1 + st - 1] < s_hi) j1 += st;
        }
        if (ends[j0] <= s_lo) j0++;   // (the searches above count up to 63; the 64th entry decides the last step)
        if (starts[j1] < s_hi) j1++;
        const uint32_t hi = j1 < lane ? j1 : lane;  // only earlier lanes (a match's own overlap is a forward copy)
        if (pend && j0 < hi) dep = (hi >= 64 ? ~0ull : (1ull << hi) - 1ull) & ~((1ull << j0) - 1ull);
        __builtin_amdgcn_wave_barrier();
    }
    if (prof) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t1 = __builtin_amdgcn_s_memtime(); prof->t_lits += t1 - t0; t0 = t1; }
    for (;;) {
        __builtin_amdgcn_wave_barrier();
        const uint64_t pm = __ballot(pend);
        if (!pm) break;
        if (prof) prof->rounds++;
        const bool ready = pend && (dep & pm) == 0ull;
        // The common round (text: nearly all of them): every ready match is at most 16 bytes, sits in the window and
        // does not overlap itself -> one 16-byte read and a prefix store per lane.  A single wave has nobody to hide
        // its instruction latency behind, so a round costs what its instruction COUNT costs.
        if (__ballot(ready && !(ml <= 16 && msrc >= lds_lo && off >= ml)) == 0ull) {
            if (ready) {
                const uint4 v = lds_ld16(W + msrc);
                if (ml == 16) lds_st16(W + mdst, v);
                else store_prefix16(W + mdst, v, ml);
                pend = false;
            }
            continue;
        }
        if (ready && ml <= LANE_MAX) {
            uint32_t k = 0;
            if (msrc < lds_lo) {  // (part of) the source was flushed long ago: read it back from HBM
                const uint32_t nf = (uint32_t)(lds_lo - msrc) < ml ? (uint32_t)(lds_lo - msrc) : ml;
                const uint8_t *g = out + (chunk_abs - WH) + (int64_t)msrc;
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
                coop_copy(Wg + dj, out + (chunk_abs - WH) + (int64_t)sj, nf, lane, 64);
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
                    const uint4 v = lds_ld16(W + (sj + (int32_t)(k + i)));
                    if (step - i >= 16) lds_st16(W + dj + k + i, v);
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
template <uint32_t WH = WIN_HIST>
__device__ __forceinline__ uint32_t win_flush(uint8_t *W, uint8_t *out, uint64_t chunk_abs, uint32_t win_n, uint32_t hist_n, uint32_t lane,
                                             bool keep_history) {
    if (win_n) coop_copy(out + chunk_abs, W + WH, win_n, lane, 64);
    uint32_t h = 0;
    if (keep_history) {
        h = hist_n + win_n < WH ? hist_n + win_n : WH;
        const uint8_t *sp = W + WH + win_n - h;
        uint8_t *dp = W + WH - h;
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

// ---------------------------------------------------------------------------------------------
// lane-0 serial helpers
// ---------------------------------------------------------------------------------------------
template <class Shared>
__device__ int fse_read_ncount(Shared &S, const uint8_t *src, uint32_t n, int max_log, int max_sym, int *nsym,
                               int *log, uint32_t *consumed, int nbase = 0) {
    FwdR b{src, n, 0};
    if (n == 0) return E_TRUNC;
    int alog = 5 + (int)b.read(4);
    if (alog > max_log) return E_CORRUPT;
    int remaining = 1 << alog, s = 0;
    while (remaining > 0 && s <= max_sym) {
        int bits = hibit((uint32_t)remaining + 1) + 1;
        uint32_t val = b.read(bits);
        uint32_t lower_mask = (1u << (bits - 1)) - 1;
        uint32_t threshold = (1u << bits) - 1 - ((uint32_t)remaining + 1);
        if ((val & lower_mask) < threshold) {
            b.bitpos -= 1;
            val &= lower_mask;
        } else if (val > lower_mask) {
            val -= threshold;
        }
        int proba = (int)val - 1;
        remaining -= proba < 0 ? -proba : proba;
        S.norm[nbase + s++] = (int16_t)proba;
        if (proba == 0) {
            uint32_t rep = b.read(2);
            for (;;) {
                for (uint32_t i = 0; i < rep && s <= max_sym; i++) S.norm[nbase + s++] = 0;
                if (rep == 3) rep = b.read(2); else break;
            }
        }
    }
    if (remaining != 0) return E_CORRUPT;
    if ((b.bitpos + 7) / 8 > n) return E_TRUNC;
    *nsym = s;
    *log = alog;
    *consumed = (b.bitpos + 7) / 8;
    return 0;
}

// Build a decoding table from S.norm[0..nsym).  kind selects how (symbol -> base, addbits) maps:
// LL / ML use the RFC's code tables, OF codes carry `code` extra bits on base 1<<code,
// kind < 0 = plain symbols (Huffman weights).
template <class Shared>
__device__ int fse_build(Shared &S, FseEntry *t, int nsym, int log, int kind) {
    const int size = 1 << log;
    int high = size;
    for (int s = 0; s < nsym; s++)
        if (S.norm[s] == -1) { S.fse_sym[--high] = (uint8_t)s; S.fse_next[s] = 1; }
    const int step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
    int pos = 0;
    for (int s = 0; s < nsym; s++) {
        int c = S.norm[s];
        if (c <= 0) continue;
        S.fse_next[s] = (uint16_t)c;
        for (int i = 0; i < c; i++) {
            S.fse_sym[pos] = (uint8_t)s;
            do { pos = (pos + step) & mask; } while (pos >= high);
        }
    }
    if (pos != 0) return E_CORRUPT;
    for (int i = 0; i < size; i++) {
        uint32_t sym = S.fse_sym[i];
        uint32_t ns = S.fse_next[sym]++;
        int nb = log - hibit(ns);
        FseEntry e;
        e.next = (uint16_t)((ns << nb) - size);
        e.nbits = (uint8_t)nb;
        if (kind == K_LL) { if (sym > 35) return E_CORRUPT; e.base = c_ll_base[sym]; e.addbits = c_ll_bits[sym]; }
        else if (kind == K_ML) { if (sym > 52) return E_CORRUPT; e.base = c_ml_base[sym]; e.addbits = c_ml_bits[sym]; }
        else if (kind == K_OF) { if (sym > 31) return E_CORRUPT; e.base = 1u << sym; e.addbits = (uint8_t)sym; }
        else { e.base = sym; e.addbits = 0; }
        t[i] = e;
    }
    return 0;
}

// fse_build by one wave, for alphabets of at most 64 symbols (the three sequence tables): lane = symbol for the
// counts, lane = table cell for the entries.  The serial version is two loops of dependent LDS accesses (spread the
// symbols, then hand every cell its symbol's next state: ~250 cycles per cell for a lone wave, 100-700 kcycles per block
// for the three tables).  Here a cell finds its symbol directly: the spread visits cells in the order 0, step, 2*step, ...
// (mod size) and skips the cells at the top that the "less than 1" symbols own, so cell u is the
// (u / step mod size) - (top cells visited earlier) -th cell handed out, and its symbol is the one whose cumulative
// count covers that index; the state a cell gets is its symbol's count plus the cell's rank among the symbol's
// cells, counted 64 cells at a time with ballots.  scr: 128 u16 of LDS scratch.  Counts are already validated
// (fse_read_ncount: they sum to the table size).
__device__ void fse_build_wave(const int16_t *norm, uint32_t nsym, uint32_t log, int kind, FseEntry *t, uint16_t *scr, uint32_t lane) {
    const uint32_t size = 1u << log, mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
    uint32_t inv = step;  // inverse of the odd step modulo 2^log (Newton: 3 -> 6 -> 12 correct bits)
    inv *= 2u - step * inv; inv *= 2u - step * inv;
    const int c = lane < nsym ? (int)norm[lane] : 0;
    const bool low = c == -1;
    const uint32_t cnt = c > 0 ? (uint32_t)c : 0u;
    const uint64_t lowm = __ballot(low), below = lane ? (~0ull >> (64 - lane)) : 0ull;
    const uint32_t nlow = (uint32_t)__popcll(lowm), high = size - nlow;
    auto entry = [&](uint32_t sym, uint32_t ns) -> FseEntry {
        FseEntry e;
        const uint32_t nb = log - (uint32_t)hibit(ns);
        e.next = (uint16_t)((ns << nb) - size);
        e.nbits = (uint8_t)nb;
        if (kind == K_LL) { e.base = c_ll_base[sym > 35 ? 35 : sym]; e.addbits = c_ll_bits[sym > 35 ? 35 : sym]; }
        else if (kind == K_ML) { e.base = c_ml_base[sym > 52 ? 52 : sym]; e.addbits = c_ml_bits[sym > 52 ? 52 : sym]; }
        else if (kind == K_OF) { e.base = 1u << (sym & 31); e.addbits = (uint8_t)sym; }
        else { e.base = sym; e.addbits = 0; }  // kind < 0: plain symbols (Huffman weights)
        return e;
    };
    if (low) t[size - 1 - (uint32_t)__popcll(lowm & below)] = entry(lane, 1);
    uint32_t incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(incl, d);
        if (lane >= (uint32_t)d) incl += y;
    }
    uint16_t *const cum = scr, *const cur = scr + 64;
    cum[lane] = (uint16_t)incl;
    cur[lane] = (uint16_t)cnt;
    const uint32_t jh = lane < nlow ? ((high + lane) * inv) & mask : 0xFFFFFFFFu;  // when the spread would have reached top cell `lane`
    __builtin_amdgcn_wave_barrier();
    for (uint32_t u0 = 0; u0 < high; u0 += 64) {
        const uint32_t u = u0 + lane;
        const bool on = u < high;
        const uint32_t j = (u * inv) & mask;
        uint32_t less = 0;
        for (uint32_t r = 0; r < nlow; r++) less += rdlane_u(jh, r) < j ? 1u : 0u;
        const uint32_t kf = on ? j - less : 0u;
        uint32_t sidx = 0;
#pragma unroll
        for (uint32_t st = 32; st; st >>= 1)
            if (cum[sidx + st - 1] <= kf) sidx += st;
        uint64_t todo = __ballot(on);
        uint32_t ns = 1;
        while (todo) {
            const uint32_t l = (uint32_t)__ffsll((long long)todo) - 1, sl = rdlane_u(sidx, l) & 63;
            const uint64_t same = __ballot(on && sidx == sl);
            const uint32_t b0 = cur[sl];
            if (on && sidx == sl) ns = b0 + (uint32_t)__popcll(same & below);
            if (lane == l) cur[sl] = (uint16_t)(b0 + (uint32_t)__popcll(same));
            todo &= ~same;
        }
        if (on) t[u] = entry(sidx, ns ? ns : 1);
    }
    __builtin_amdgcn_wave_barrier();
}

__device__ int fse_set_rle(FseEntry *e, uint32_t sym, int kind) {
    e->next = 0; e->nbits = 0;
    if (kind == K_LL) { if (sym > 35) return E_CORRUPT; e->base = c_ll_base[sym]; e->addbits = c_ll_bits[sym]; }
    else if (kind == K_ML) { if (sym > 52) return E_CORRUPT; e->base = c_ml_base[sym]; e->addbits = c_ml_bits[sym]; }
    else { if (sym > 31) return E_CORRUPT; e->base = 1u << sym; e->addbits = (uint8_t)sym; }
    return 0;
}

// Huffman tree description -> S.weights / S.sym_start / S.sym_len / S.huf_log (lane 0).
template <class Shared>
__device__ int huf_read_tree(Shared &S, const uint8_t *src, uint32_t n, const uint8_t *blob_end, uint32_t *consumed) {
    if (n < 1) return E_TRUNC;
    uint32_t hb = src[0];
    int nw = 0;
    if (hb >= 128) {
        int num = (int)hb - 127;
        uint32_t bytes = (uint32_t)(num + 1) / 2;
        if (1 + bytes > n) return E_TRUNC;
        for (int i = 0; i < num; i++) {
            uint8_t b = src[1 + i / 2];
            S.weights[i] = (i & 1) ? (b & 15) : (b >> 4);
        }
        nw = num;
        *consumed = 1 + bytes;
    } else {
        if (hb == 0 || 1 + hb > n) return E_TRUNC;
        int nsym, log;
        uint32_t hdr;
        int rc = fse_read_ncount(S, src + 1, hb, 6, 255, &nsym, &log, &hdr);
        if (rc) return rc;
        // weights table reuses the `of` slot region? no: keep sequence tables intact (repeat mode) -> use ll? also live.
        // A 64-entry table fits in the dml slot only if ML is not in default mode later, so build into a
        // private region: the seq_ll batch buffer is free while literals are being decoded.
        FseEntry *t = reinterpret_cast<FseEntry *>(S.seq_ll);
        rc = fse_build(S, t, nsym, log, -1);
        if (rc) return rc;
        if (hdr >= hb) return E_CORRUPT;
        BitR b;
        if (!b.init(src + 1 + hdr, hb - hdr, blob_end)) return E_CORRUPT;
        uint32_t s1 = b.read(log), s2 = b.read(log);
        for (;;) {
            if (nw >= 255) return E_CORRUPT;
            S.weights[nw++] = (uint8_t)t[s1].base;
            s1 = t[s1].next + b.read(t[s1].nbits);
            if (b.pos < 0) {
                if (nw >= 255) return E_CORRUPT;
                S.weights[nw++] = (uint8_t)t[s2].base;
                break;
            }
            if (nw >= 255) return E_CORRUPT;
            S.weights[nw++] = (uint8_t)t[s2].base;
            s2 = t[s2].next + b.read(t[s2].nbits);
            if (b.pos < 0) {
                if (nw >= 255) return E_CORRUPT;
                S.weights[nw++] = (uint8_t)t[s1].base;
                break;
            }
        }
        *consumed = 1 + hb;
    }
    // implied last weight, code lengths, canonical start index per symbol
    uint32_t total = 0;
    for (int i = 0; i < nw; i++) {
        uint32_t w = S.weights[i];
        if (w > 12) return E_CORRUPT;
        total += w ? 1u << (w - 1) : 0;
    }
    if (total == 0) return E_CORRUPT;
    int maxbits = hibit(total) + 1;
    if (maxbits > 11) return E_CORRUPT;
    uint32_t left = (1u << maxbits) - total;
    if (left & (left - 1)) return E_CORRUPT;
    S.weights[nw] = (uint8_t)(hibit(left) + 1);
    int nsym = nw + 1;
    uint32_t rank_count[13], rank_idx[13];
    for (int i = 0; i < 13; i++) rank_count[i] = 0;
    for (int i = 0; i < nsym; i++) {
        uint32_t w = S.weights[i];
        rank_count[w ? maxbits + 1 - w : 0]++;
    }
    rank_idx[maxbits] = 0;
    for (int i = maxbits; i >= 1; i--) rank_idx[i - 1] = rank_idx[i] + rank_count[i] * (1u << (maxbits - i));
    if (rank_idx[0] != (1u << maxbits)) return E_CORRUPT;
    for (int i = 0; i < 256; i++) S.sym_len[i] = 0;
    for (int i = 0; i < nsym; i++) {
        uint32_t w = S.weights[i];
        if (!w) continue;
        uint32_t bits = maxbits + 1 - w, len = 1u << (maxbits - bits);
        S.sym_start[i] = (uint16_t)rank_idx[bits];
        S.sym_len[i] = (uint16_t)len;
        rank_idx[bits] += len;
    }
    S.huf_log = maxbits;
    return 0;
}

// huf_read_tree by one wave (the general / block decoder).  What is serial by nature stays on lane 0 (reading the
// counts of the weight table, decoding the at most 255 weights with two interleaved states); the weight table itself is
// built by the wave (fse_build_wave), and so is everything behind the weights: validity, the implied last weight, and every
// symbol's range in the decoding table — four symbols per lane, ranks by ballots in (weight, symbol) order.  Same
// verdicts as the serial version, which the foreign-frame path still uses.
template <class Shared>
__device__ int huf_read_tree_wave(Shared &S, const uint8_t *src, uint32_t n, const uint8_t *blob_end, uint32_t lane) {
    if (n < 1) return E_TRUNC;
    const uint32_t hb = uni((uint32_t)src[0]);
    uint32_t nw = 0;
    if (hb >= 128) {
        nw = hb - 127;
        const uint32_t bytes = (nw + 1) / 2;
        if (1 + bytes > n) return E_TRUNC;
        for (uint32_t i = lane; i < nw; i += 64) {
            const uint8_t b = src[1 + i / 2];
            S.weights[i] = (i & 1) ? (b & 15) : (b >> 4);
        }
    } else {
        if (hb == 0 || 1 + hb > n) return E_TRUNC;
        int rc = 0, nsym = 0, log = 0;
        uint32_t hdr = 0;
        if (lane == 0) rc = fse_read_ncount(S, src + 1, hb, 6, 255, &nsym, &log, &hdr);
        rc = (int)uni((uint32_t)rc); nsym = (int)uni((uint32_t)nsym); log = (int)uni((uint32_t)log); hdr = uni(hdr);
        if (rc) return rc;
        FseEntry *t = reinterpret_cast<FseEntry *>(S.seq_ll);  // free while literals are being decoded
        __builtin_amdgcn_wave_barrier();
        if (nsym <= 64) fse_build_wave(S.norm, (uint32_t)nsym, (uint32_t)log, -1, t, reinterpret_cast<uint16_t *>(S.fse_next), lane);
        else {  // a table that names symbols beyond 63 (no weight is that large; the weights decide below)
            if (lane == 0) rc = fse_build(S, t, nsym, log, -1);
            rc = (int)uni((uint32_t)rc);
            if (rc) return rc;
        }
        __builtin_amdgcn_wave_barrier();
        if (hdr >= hb) return E_CORRUPT;
        if (lane == 0) {
            BitR b;
            if (!b.init(src + 1 + hdr, hb - hdr, blob_end)) rc = E_CORRUPT;
            else {
                uint32_t s1 = b.read(log), s2 = b.read(log);
                for (;;) {
                    if (nw >= 255) { rc = E_CORRUPT; break; }
                    S.weights[nw++] = (uint8_t)t[s1].base;
                    s1 = t[s1].next + b.read(t[s1].nbits);
                    if (b.pos < 0) {
                        if (nw >= 255) { rc = E_CORRUPT; break; }
                        S.weights[nw++] = (uint8_t)t[s2].base;
                        break;
                    }
                    if (nw >= 255) { rc = E_CORRUPT; break; }
                    S.weights[nw++] = (uint8_t)t[s2].base;
                    s2 = t[s2].next + b.read(t[s2].nbits);
                    if (b.pos < 0) {
                        if (nw >= 255) { rc = E_CORRUPT; break; }
                        S.weights[nw++] = (uint8_t)t[s1].base;
                        break;
                    }
                }
            }
        }
        rc = (int)uni((uint32_t)rc); nw = uni(nw);
        if (rc) return rc;
    }
    __builtin_amdgcn_wave_barrier();
    // implied last weight, code lengths, canonical start index per symbol: lane owns symbols lane, +64, +128, +192
    uint32_t w[4], sum = 0;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t idx = lane + 64 * j;
        w[j] = idx < nw ? S.weights[idx] : 0;
        if (w[j] > 12) bad = true;
        sum += (w[j] && w[j] <= 12) ? 1u << (w[j] - 1) : 0;
    }
    if (__ballot(bad)) return E_CORRUPT;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
    const uint32_t total = sum;
    if (total == 0) return E_CORRUPT;
    const uint32_t maxbits = (uint32_t)hibit(total) + 1;
    if (maxbits > 11) return E_CORRUPT;
    const uint32_t left = (1u << maxbits) - total;
    if (left & (left - 1)) return E_CORRUPT;
    const uint32_t lastw = (uint32_t)hibit(left) + 1;
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (lane + 64 * j == nw) { w[j] = lastw; S.weights[nw] = (uint8_t)lastw; }
    const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0ull;
    uint32_t start = 0, st[4] = {0, 0, 0, 0}, ln[4] = {0, 0, 0, 0};
    for (uint32_t bits = maxbits; bits >= 1; bits--) {  // longest codes first, as the table is laid out
        const uint32_t wt = maxbits + 1 - bits, len = 1u << (maxbits - bits);
        uint32_t before = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint64_t m = __ballot(w[j] == wt);
            if (w[j] == wt) { st[j] = start + ((before + (uint32_t)__popcll(m & below)) << (maxbits - bits)); ln[j] = len; }
            before += (uint32_t)__popcll(m);
        }
        start += before << (maxbits - bits);
    }
    if (start != (1u << maxbits)) return E_CORRUPT;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        S.sym_start[lane + 64 * j] = (uint16_t)st[j];
        S.sym_len[lane + 64 * j] = (uint16_t)ln[j];
    }
    if (lane == 0) S.huf_log = maxbits;
    return 0;
}

__device__ int fz_huf_stream(const uint16_t *huf, uint32_t log, const uint8_t *p, uint32_t n, const uint8_t *blob_end,
                             uint8_t *dst, uint32_t n_out);

// One lane decodes one Huffman stream (n_out symbols) from [p, p+n) into dst.
template <class Shared>
__device__ int huf_decode_stream(const Shared &S, const uint8_t *p, uint32_t n, const uint8_t *blob_end,
                                 uint8_t *dst, uint32_t n_out) {
    BitR b;
    if (!b.init(p, n, blob_end)) return E_CORRUPT;
    const uint32_t log = S.huf_log;
    for (uint32_t i = 0; i < n_out; i++) {
        uint32_t e = S.huf[b.peek(log)];
        dst[i] = (uint8_t)e;
        b.pos -= e >> 8;
    }
    return b.pos == 0 ? 0 : E_CORRUPT;
}

// XXH64 (RFC 8878 §3.1.1 content checksum), seed 0, computed by lanes 0..3 of one wave (one stripe
// accumulator each; the stripe recurrence is serial by definition), finalised by lane 0.
__device__ uint64_t wave_xxh64(const uint8_t *p, uint64_t len, uint32_t lane) {
    const uint64_t P1 = 0x9E3779B185EBCA87ull, P2 = 0xC2B2AE3D27D4EB4Full, P3 = 0x165667B19E3779F9ull,
                   P4 = 0x85EBCA77C2B2AE63ull, P5 = 0x27D4EB2F165667C5ull;
    auto rotl = [](uint64_t x, int r) { return (x << r) | (x >> (64 - r)); };
    auto rd64 = [](const uint8_t *q) { uint64_t v; __builtin_memcpy(&v, q, 8); return v; };
    auto rd32 = [](const uint8_t *q) { uint32_t v; __builtin_memcpy(&v, q, 4); return v; };
    auto round = [&](uint64_t acc, uint64_t in) { acc += in * P2; acc = rotl(acc, 31); return acc * P1; };
    uint64_t v = 0;
    const uint64_t stripes = len / 32;
    if (lane < 4) {
        v = lane == 0 ? P1 + P2 : (lane == 1 ? P2 : (lane == 2 ? 0 : 0 - P1));
        const uint8_t *q = p + 8 * lane;
        for (uint64_t i = 0; i < stripes; i++) v = round(v, rd64(q + 32 * i));
    }
    const uint64_t v1 = __shfl(v, 0), v2 = __shfl(v, 1), v3 = __shfl(v, 2), v4 = __shfl(v, 3);
    uint64_t h;
    if (len >= 32) {
        h = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18);
        auto merge = [&](uint64_t acc, uint64_t val) { val = round(0, val); acc ^= val; return acc * P1 + P4; };
        h = merge(h, v1); h = merge(h, v2); h = merge(h, v3); h = merge(h, v4);
    } else {
        h = P5;
    }
    h += len;
    const uint8_t *q = p + stripes * 32, *end = p + len;
    while (q + 8 <= end) { h ^= round(0, rd64(q)); h = rotl(h, 27) * P1 + P4; q += 8; }
    if (q + 4 <= end) { h ^= (uint64_t)rd32(q) * P1; h = rotl(h, 23) * P2 + P3; q += 4; }
    while (q < end) { h ^= (*q) * P5; h = rotl(h, 11) * P1; q++; }
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    return h;
}

// One lane, one Huffman stream.  While 64 or more bits are unread four symbols (<= 44 bits) come out of one 8-byte
// load with no end-of-stream case and leave as one 4-byte store; the last few go through the guarded reader.
__device__ int fz_huf_stream(const uint16_t *huf, uint32_t log, const uint8_t *p, uint32_t n, const uint8_t *blob_end,
                             uint8_t *dst, uint32_t n_out) {
    BitR b;
    if (!b.init(p, n, blob_end)) return E_CORRUPT;
    const uint32_t mask = (1u << log) - 1;
    uint32_t i = 0;
    int64_t pos = b.pos;
    while (i + 4 <= n_out && pos >= 64) {
        const int64_t b0 = ((pos + 7) >> 3) - 8;
        uint64_t c;
        __builtin_memcpy(&c, p + b0, 8);
        int32_t avail = (int32_t)(pos - b0 * 8);  // 57..64 bits of c lie below `pos`
        uint32_t w = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t e = huf[(uint32_t)(c >> (avail - (int32_t)log)) & mask];
            w |= (e & 0xFFu) << (8 * j);
            avail -= (int32_t)(e >> 8);
        }
        pos = b0 * 8 + avail;
        __builtin_memcpy(dst + i, &w, 4);
        i += 4;
    }
    b.pos = pos;
    b.refill();
    for (; i < n_out; i++) {
        const uint32_t e = huf[b.peek(log)];
        dst[i] = (uint8_t)e;
        b.pos -= e >> 8;
    }
    return b.pos == 0 ? 0 : E_CORRUPT;
}

}  // namespace zn
