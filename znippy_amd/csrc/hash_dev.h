// Device-side body of the BLAKE3 tile hash (one wavefront = one Tile, lane = 1 KiB leaf),
// shared by the standalone hash kernel (hash_kernels.hip) and the fused small-row
// decode+hash kernel (fused_small.hip).
#pragma once
#include "common.h"
#include "blake3_dev.h"
#include <type_traits>

namespace zn {

// 16 bytes at any byte address of the LDS (address space 3: a ds_read_b128, whatever the pointer came through)
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
typedef u4v __attribute__((aligned(1))) u4v_unaligned;
typedef __attribute__((address_space(3))) u4v_unaligned lds_u4;

// 16 bytes at any byte address of device memory.  The pointers that reach these two have often been through an integer
// (a 64-bit address handed from lane to lane as two dwords), which leaves the compiler with a generic pointer and FLAT
// instructions: those count on lgkmcnt as well as vmcnt, so every LDS read behind them waits for all of the wave's
// outstanding loads AND stores.  Address space 1 says what they are: global_load / global_store_dwordx4.
typedef __attribute__((address_space(1))) u4v_unaligned glb_u4;
typedef __attribute__((address_space(1))) uint8_t glb_u8;
__device__ __forceinline__ uint4 ld16(const uint8_t *p) {
    const u4v v = *(const glb_u4 *)(uintptr_t)p;  // unaligned-access-mode: one global_load_dwordx4
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st16(uint8_t *p, uint4 v) { *(glb_u4 *)(uintptr_t)p = u4v{v.x, v.y, v.z, v.w}; }

// Load one (possibly partial) 64-byte block into 16 little-endian words, zero padded.
__device__ __forceinline__ void load_block(const uint8_t *p, uint32_t n, uint32_t m[16]) {
    if (n == 64) {
        uint4 a = ld16(p), b = ld16(p + 16), c = ld16(p + 32), d = ld16(p + 48);
        m[0] = a.x; m[1] = a.y; m[2] = a.z; m[3] = a.w;
        m[4] = b.x; m[5] = b.y; m[6] = b.z; m[7] = b.w;
        m[8] = c.x; m[9] = c.y; m[10] = c.z; m[11] = c.w;
        m[12] = d.x; m[13] = d.y; m[14] = d.z; m[15] = d.w;
    } else {
        const glb_u8 *const g = (const glb_u8 *)(uintptr_t)p;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            uint32_t w = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                uint32_t idx = 4 * i + k;
                if (idx < n) w |= (uint32_t)g[idx] << (8 * k);
            }
            m[i] = w;
        }
    }
}

__device__ __forceinline__ void store_block(uint8_t *p, uint32_t n, const uint32_t m[16]) {
    if (n == 64) {
        st16(p, make_uint4(m[0], m[1], m[2], m[3]));
        st16(p + 16, make_uint4(m[4], m[5], m[6], m[7]));
        st16(p + 32, make_uint4(m[8], m[9], m[10], m[11]));
        st16(p + 48, make_uint4(m[12], m[13], m[14], m[15]));
    } else {
        glb_u8 *const g = (glb_u8 *)(uintptr_t)p;
        for (uint32_t i = 0; i < n; i++) g[i] = (uint8_t)(m[i >> 2] >> (8 * (i & 3)));
    }
}

// Fold the CVs held by lanes [s, s+n) (one node per lane, node j in lane s+j) into lane s.
// All 64 lanes call this together; lanes outside any segment pass n = 0.  `final_root`:
// the fold ends at the unit's root (ROOT flag on the last parent).
__device__ __forceinline__ void fold_segments(uint32_t cv[8], uint32_t s, uint32_t n, bool final_root) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t j = lane - s;
    while (__ballot(n > 1) != 0ull) {
        uint32_t li = (s + 2 * j) & 63, ri = (s + 2 * j + 1) & 63;
        uint32_t L[8], R[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            L[i] = __shfl(cv[i], li);
            R[i] = __shfl(cv[i], ri);
        }
        if (n > 1 && j < (n + 1) / 2) {
            if (2 * j + 1 < n) b3::parent(cv, L, R, final_root && n == 2);
            else {
#pragma unroll
                for (int i = 0; i < 8; i++) cv[i] = L[i];
            }
        }
        if (n > 1) n = (n + 1) / 2;
    }
}

// Which units of a tile a pass hashes (HashArgs::pass):
//   PASS_ALL    no status array, hash everything (write side, hash-only)
//   PASS_FUSED  small tiles only; a unit is hashed iff status == 0 (stored, or decoded by the fused path)
//   PASS_SECOND big-unit slices (status >= 0) and small-tile units the general decoder finished (status == 2)
enum { PASS_ALL = 0, PASS_FUSED = 1, PASS_SECOND = 2 };

// Optional on-chip source for the fused kernel: rows whose whole output is "literal prefix + one
// periodic match" are hashed from LDS (the staged literals followed by 64 bytes of the period),
// so their decoded bytes are written to HBM once and never read back.
//   out[i] = Y[i]                          for i <  B + off + 64   (Y = ybase .. in LDS)
//   out[i] = Y[B + ((i - B) mod off)]      for i >= B              (period `off` starts at B)
struct LdsSrc {
    const uint8_t *wl;        // this wave's staged windows
    const uint16_t *ybase;    // per tile-local row: offset of Y in wl, 0xFFFF = not periodic
    const uint16_t *pB, *poff;
    uint32_t rows;            // rows that have descriptors
    const int32_t *st;        // per tile-local row status kept on-chip (replaces the global status read)
    // the tile's index columns as loaded by the prologue, lane u = row u of the tile: with these the
    // hash issues no global load of its own, so it never queues behind the decode's pending stores
    uint64_t c_len, c_src, c_oo;
    uint32_t c_sel;
    // bit u: tile-local row u has not been written yet — its lanes store every 64-byte block they hash (the data
    // is in the message registers anyway), so writing such a row costs four store instructions per compression
    // and no data movement of its own
    uint32_t store_mask;
    // big-slice tiles (t.n_units == 0, fused block kernel): descriptor 0 describes the 128 KiB block the slice
    // belongs to and `origin` is that block's position inside the row (window positions are block-relative)
    uint64_t origin;
};

// What the leaf phase of one tile leaves in the wave: one chaining value per leaf lane plus the shape of
// the tile's units (segments of consecutive lanes).
struct LeafOut {
    uint32_t cv[8];
    uint32_t seg_start, n;  // this lane's segment: first lane, node count (0 on inactive lanes)
    uint32_t unit;
    bool active;
    uint32_t u_cnt, u_head;  // tile-local unit i sits on lane i: its leaf count and first lane (big tile: lane 0)
};

// Leaf phase of one tile: every active lane compresses its 1 KiB leaf (16 blocks).  Every lane of the wave
// must call it.
// `stage` (COPY only): 64 x STAGE_SLOT bytes of wave-private LDS.  A lane's copy of its 64-byte block is four
// 16-byte stores 1 KiB away from every other lane's — 64 different cache lines per store instruction, which
// measured ~60 ns of SIMD time each (tools/ubench_b3.hip).  With a stage the block goes through LDS instead and
// leaves transposed: four lanes write one leaf's 64 contiguous bytes, 16 leaves per instruction.
#ifndef ZN_STAGE_LOADS
#define ZN_STAGE_LOADS 1
#endif
#ifndef ZN_RAGGED_STAGE
#define ZN_RAGGED_STAGE 1
#endif
constexpr uint32_t STAGE_SLOT = 80;  // 64 + 16: consecutive lanes' slots start 20 banks apart (conflict-free b128)
constexpr uint32_t STAGE_BYTES = 64 * STAGE_SLOT;
typedef __attribute__((address_space(3))) u4v lds_u4a;  // 16-byte aligned LDS vector

constexpr uint32_t STAGE_FULL_SLOT = 144;  // 128 + 16: whole cache lines per leaf (big-slice tiles of the store path)
constexpr uint32_t STAGE_FULL_BYTES = 64 * STAGE_FULL_SLOT;
// The store path kernel's own stage (k_hash_tiles<COPY>): 128-byte slots with no padding — 8 KiB per wave, five blocks of
// four waves per CU instead of four.  Piece p of slot l sits at piece p ^ ((l >> 1) & 7): 16 consecutive lanes reading the
// same piece of their own slots then touch 16 different 16-byte columns x {low, high} half of the 64 banks, and the eight
// lanes that write one slot's eight pieces still write its 128 contiguous bytes.
constexpr uint32_t STAGE_SWZ_BYTES = 64 * 128;

// `raw_src` (big-slice tiles, fused block kernel): the tile's bytes are copied from raw_src + leaf offset instead of
// the unit's own source column (a raw block of a compressed frame: blob -> output while hashing).
constexpr uint32_t STAGE_SHIFT_SLOT = 160;  // 16 (tail of the previous 128 bytes) + 128 + 16
constexpr uint32_t STAGE_SHIFT_BYTES = 64 * STAGE_SHIFT_SLOT;

// Ragged leaves on the store path, out of line (its registers are its own: inlined it cost the whole-leaf forms of the same
// kernel 1-3 %).  Every lane of the wave must call it.  Returns the lane's chaining value.
struct Cv8 { uint32_t v[8]; };
__device__ __noinline__ Cv8 hash_ragged_through_stage(const uint8_t *src, uint8_t *dst, uint32_t leaf_len, uint32_t nblk, uint32_t k,
                                                     bool single, bool active, uint8_t *stage) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t cv[8];
    b3::set_iv(cv);
    // Ragged leaves on the store path (the last leaf of a row that is not a whole number of KiB — nearly every row of a
    // real archive — or an empty row): the same two-way stage as the whole-leaf form below it in the other branch (lane l
    // moves piece l % 4 of leaves 16j + l / 4, 64 bytes per leaf per step, the next step's loads in flight), with every
    // piece cut to what its leaf still has: whole pieces as 16-byte accesses, the one partial piece of a leaf as the 16
    // bytes that end with the leaf (nothing is read or written beyond a row's last byte), absent ones as zeros in the stage — the padding the
    // hash wants.  A lane hashes as many blocks as its leaf has.  (Until this form every tile with one ragged leaf took
    // the generic loop below: each lane its own 64-byte loads and stores 1 KiB apart, half the speed.)
    const uint64_t has = __ballot(active && dst != nullptr), act = __ballot(active);
    const uint8_t *sj[4];
    uint8_t *pj[4];
    bool onj[4], actj[4];
    lds_u4a *rs[4];
    int32_t lrem[4];  // bytes of my piece's leaf from my piece's first byte on (at step 0)
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t leaf = 16 * j + (lane >> 2);
        const uint64_t d64 = (uint64_t)(uintptr_t)dst, s64 = (uint64_t)(uintptr_t)src;
        const uint64_t g = ((uint64_t)__shfl((uint32_t)(d64 >> 32), leaf) << 32) | __shfl((uint32_t)d64, leaf);
        const uint64_t h = ((uint64_t)__shfl((uint32_t)(s64 >> 32), leaf) << 32) | __shfl((uint32_t)s64, leaf);
        pj[j] = reinterpret_cast<uint8_t *>((uintptr_t)g) + 16 * (lane & 3);
        sj[j] = reinterpret_cast<const uint8_t *>((uintptr_t)h) + 16 * (lane & 3);
        onj[j] = (has >> leaf) & 1;
        actj[j] = (act >> leaf) & 1;
        rs[j] = (lds_u4a *)(stage + leaf * STAGE_SLOT + 16 * (lane & 3));
        lrem[j] = (int32_t)__shfl(leaf_len, leaf) - 16 * (int32_t)(lane & 3);
    }
    const lds_u4a *ws = (const lds_u4a *)(stage + lane * STAGE_SLOT);
    uint32_t maxblk = active ? nblk : 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        uint32_t o = __shfl_xor(maxblk, d);
        maxblk = o > maxblk ? o : maxblk;
    }
    maxblk = __builtin_amdgcn_readfirstlane(maxblk);
    // my piece of block b of leaf group j, zero padded.  PARTIAL = this step has a piece of 1 .. 15 bytes somewhere in the
    // wave (wave-uniform: at most one step per ragged leaf); the other steps know only whole and absent pieces.
    auto piece = [&](int j, uint32_t b, bool partial) -> uint4 {
        uint4 v = make_uint4(0, 0, 0, 0);
        const int32_t rem = lrem[j] - 64 * (int32_t)b;
        if (actj[j] && rem >= 16) v = ld16(sj[j] + b * 64);
        else if (partial && actj[j] && rem > 0) {
            // the leaf's last 1 .. 15 bytes: the 16 bytes that END with them (inside the row: the caller sends no row shorter
            // than 16 bytes here), shifted down — nothing is read beyond the row's last byte
            const uint4 o = ld16(sj[j] + b * 64 + rem - 16);
            const uint32_t sh = 16u - (uint32_t)rem, ds = sh >> 2, bs = 8u * (sh & 3u);
            const uint32_t x[4] = {o.x, o.y, o.z, o.w};
            uint32_t t[5];
#pragma unroll
            for (int i = 0; i < 5; i++) {
                const uint32_t a0 = i < 4 ? x[i < 4 ? i : 0] : 0u, a1 = i + 1 < 4 ? x[i + 1 < 4 ? i + 1 : 0] : 0u,
                               a2 = i + 2 < 4 ? x[i + 2 < 4 ? i + 2 : 0] : 0u, a3 = i + 3 < 4 ? x[i + 3 < 4 ? i + 3 : 0] : 0u;
                t[i] = ds == 0 ? a0 : (ds == 1 ? a1 : (ds == 2 ? a2 : a3));
            }
            v = make_uint4(__builtin_amdgcn_alignbit(t[1], t[0], bs), __builtin_amdgcn_alignbit(t[2], t[1], bs),
                           __builtin_amdgcn_alignbit(t[3], t[2], bs), __builtin_amdgcn_alignbit(t[4], t[3], bs));
        }
        return v;
    };
    auto any_partial = [&](uint32_t b) -> bool {
        bool p = false;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int32_t rem = lrem[j] - 64 * (int32_t)b;
            p |= actj[j] && rem > 0 && rem < 16;
        }
        return __ballot(p) != 0ull;
    };
    uint4 v[4], vn[4];
    bool part = any_partial(0);
#pragma unroll
    for (int j = 0; j < 4; j++) { v[j] = piece(j, 0, part); vn[j] = make_uint4(0, 0, 0, 0); }
#pragma unroll 1
    for (uint32_t b = 0; b < maxblk; b++) {
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (actj[j]) *rs[j] = u4v{v[j].x, v[j].y, v[j].z, v[j].w};
        const bool part_next = b + 1 < maxblk && any_partial(b + 1);
        if (b + 1 < maxblk) {
            if (part_next) {
#pragma unroll
                for (int j = 0; j < 4; j++) vn[j] = piece(j, b + 1, true);
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) vn[j] = piece(j, b + 1, false);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int32_t rem = lrem[j] - 64 * (int32_t)b;
            if (onj[j] && rem >= 16) st16(pj[j] + b * 64, v[j]);
        }
        if (part) {  // (wave-uniform) the step's partial pieces: the 16 bytes that end with the leaf, to the 16 that end with it in
            // the output — the bytes in front of the piece are rewritten with what their own lane wrote (in this instruction
            // or an earlier step: the same row, the same wave, the same values)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int32_t rem = lrem[j] - 64 * (int32_t)b;
                if (onj[j] && rem > 0 && rem < 16) st16(pj[j] + b * 64 + rem - 16, ld16(sj[j] + b * 64 + rem - 16));
            }
        }
        if (active && b < nblk) {
            const u4v a0 = ws[0], a1 = ws[1], a2 = ws[2], a3 = ws[3];
            uint32_t m[16] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w, a3.x, a3.y, a3.z, a3.w};
            const uint32_t rem = leaf_len - b * 64;
            const uint32_t bl = leaf_len == 0 ? 0 : (rem < 64 ? rem : 64);
            const uint32_t flags = (b == 0 ? b3::CHUNK_START : 0u) |
                                   (b == nblk - 1 ? (b3::CHUNK_END | (single ? b3::ROOT : 0u)) : 0u);
            b3::compress(cv, m, k, 0, bl, flags);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = vn[j];
        part = part_next;
    }
    Cv8 r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = cv[i];
    return r;
}

template <bool COPY, bool LDSRC = false, bool STAGE_FULL = false, bool STAGE_SHIFT = false>
__device__ __forceinline__ void hash_tile_leaves(const HashArgs &a, const Tile &t, const LdsSrc *ls, LeafOut &out,
                                                 uint8_t *stage = nullptr, const uint8_t *raw_src = nullptr) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t unit, k, unit_leaves, seg_start, local = 0xFFFFFFFFu;
    bool active = lane < t.n_leaves;
    out.u_cnt = lane == 0 ? t.n_leaves : 0;
    out.u_head = 0;
    if (t.n_units) {
        uint32_t cnt = 0;
        if (lane < t.n_units) {
            uint64_t L = LDSRC ? ls->c_len : a.len[t.first_unit + lane];
            cnt = L ? (uint32_t)((L + 1023) >> 10) : 1u;
        }
        uint32_t inc = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t y = __shfl_up(inc, d);
            if (lane >= (uint32_t)d) inc += y;
        }
        // smallest i with inc[i] > lane
        uint32_t lo = 0, hi = t.n_units - 1;
#pragma unroll
        for (int it = 0; it < 6; it++) {
            uint32_t mid = (lo + hi) >> 1;
            uint32_t v = __shfl(inc, mid);
            if (lo < hi) {
                if (v > lane) hi = mid; else lo = mid + 1;
            }
        }
        uint32_t i = lo;
        out.u_cnt = cnt;
        out.u_head = inc - cnt;
        unit_leaves = __shfl(cnt, i);
        seg_start = __shfl(inc, i) - unit_leaves;
        unit = t.first_unit + i;
        local = i;
        k = lane - seg_start;
    } else {
        unit = t.first_unit;
        k = t.first_leaf + lane;
        unit_leaves = t.n_leaves;  // nodes of this slice
        seg_start = 0;
    }
    if (!active) { unit = t.first_unit; k = 0; }
    if (a.pass != PASS_ALL) {
        const int32_t st = (LDSRC && t.n_units) ? ls->st[local < 64 ? local : 0] : a.status[unit];
        if (a.pass == PASS_FUSED) active = active && st == 0;
        else active = active && (t.n_units ? st == 2 : st >= 0);
    }

    uint64_t ulen;
    bool from_b;
    const uint8_t *src;
    uint8_t *dst;
    if (LDSRC && t.n_units) {  // columns from the prologue's registers (row = lane `local`)
        const uint32_t li = local < 64 ? local : 0;
        ulen = __shfl(ls->c_len, li);
        from_b = __shfl(ls->c_sel, li) != 0;
        const uint64_t so = __shfl(ls->c_src, li), oo = __shfl(ls->c_oo, li);
        src = from_b ? a.srcB + oo : a.srcA + so;
        dst = (COPY && !from_b && a.srcB) ? a.srcB + oo : nullptr;
    } else {
        ulen = a.len[unit];
        from_b = a.sel && a.sel[unit];
        src = from_b ? a.srcB + a.offB[unit] : a.srcA + (a.offA[unit] - a.baseA);
        dst = (COPY && !from_b && a.srcB) ? a.srcB + a.offB[unit] : nullptr;
        if (COPY && dst && a.copy_mask && (!a.copy_mask[unit] || a.offB[unit] + ulen > a.copy_cap)) dst = nullptr;
    }
    if (!LDSRC && COPY && raw_src) {
        src = raw_src;
        dst = a.srcB + a.offB[unit];
    }
    const uint64_t leaf_off = (uint64_t)k << 10;
    uint32_t leaf_len = 0;
    if (active && ulen > leaf_off) leaf_len = (uint32_t)((ulen - leaf_off) < 1024 ? (ulen - leaf_off) : 1024);
    const uint32_t nblk = leaf_len ? (leaf_len + 63) >> 6 : 1u;
    const uint64_t total_leaves = ulen ? (ulen + 1023) >> 10 : 1;
    const bool single = total_leaves == 1;  // the leaf itself is the root

    uint32_t cv[8];
    b3::set_iv(cv);
    src += leaf_off;
    if (COPY && dst) dst += leaf_off;
    if (__ballot(active && leaf_len != 1024) == 0ull) {
        // fast path (wave-uniform): every active lane owns a full 1 KiB leaf -> 16 full blocks,
        // next block's 64 bytes are in flight while the current one is compressed
        // per-lane on-chip source (fused kernel, periodic rows): Y holds out[0 .. L0+64), L0 = B + off
        const uint8_t *Y = nullptr;
        uint32_t yB = 0, yoff = 1, yL0 = 0, r = 0, step64 = 0;
        const uint32_t yl = (LDSRC && !t.n_units) ? 0u : local;  // descriptor of this lane's row / block
        uint32_t p = (uint32_t)leaf_off;  // row position of the next block (LDS path)
        if (LDSRC && active && yl < ls->rows && ls->ybase[yl] != 0xFFFF) {
            const uint32_t local = yl;
            Y = ls->wl + ls->ybase[local];
            yB = ls->pB[local];
            yoff = ls->poff[local];
            yL0 = yB + yoff;
            p = (uint32_t)(leaf_off - ls->origin);
            const uint32_t p0 = p;
            const uint32_t p1 = p0 > yL0 ? p0 : ((yL0 >> 6) + 1) << 6;  // first block read through the period
            r = (p1 - yB) % yoff;
            step64 = 64 % yoff;
            if (COPY && (ls->store_mask >> local & 1)) dst = const_cast<uint8_t *>(src);  // src = the row's place in the output + leaf_off
        }
        uint4 n0, n1, n2, n3;
        // ALL_LDS: every active lane reads its leaf from the staged windows.  The loop is then compiled with LDS
        // instructions only (ds_read_b128, counted by lgkmcnt): with one loop for both sources the compiler has to
        // use flat loads, which count on vmcnt as well, complete in order behind every earlier global STORE of the
        // wave, and make each compression wait for the row bytes written one compression earlier.
        auto run = [&](auto all_lds) {
            constexpr bool ALL_LDS = decltype(all_lds)::value;
            auto fetch = [&](uint32_t b) {
                if (ALL_LDS || (LDSRC && Y)) {
                    const uint8_t *q = Y + p;
                    if (p > yL0) {
                        q = Y + yB + r;
                        r += step64;
                        if (r >= yoff) r -= yoff;
                    }
                    if constexpr (ALL_LDS) {
                        const lds_u4 *q3 = (const lds_u4 *)q;
                        const u4v a0 = q3[0], a1 = q3[1], a2 = q3[2], a3 = q3[3];
                        n0 = make_uint4(a0.x, a0.y, a0.z, a0.w); n1 = make_uint4(a1.x, a1.y, a1.z, a1.w);
                        n2 = make_uint4(a2.x, a2.y, a2.z, a2.w); n3 = make_uint4(a3.x, a3.y, a3.z, a3.w);
                    } else {
                        __builtin_memcpy(&n0, q, 16); __builtin_memcpy(&n1, q + 16, 16);
                        __builtin_memcpy(&n2, q + 32, 16); __builtin_memcpy(&n3, q + 48, 16);
                    }
                    p += 64;
                } else {
                    const uint8_t *q = src + b * 64;
                    n0 = ld16(q); n1 = ld16(q + 16); n2 = ld16(q + 32); n3 = ld16(q + 48);
                }
            };
            auto block = [&](uint32_t b) {
                uint32_t m[16] = {n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w,
                                  n2.x, n2.y, n2.z, n2.w, n3.x, n3.y, n3.z, n3.w};
                if (b < 15) fetch(b + 1);
                if (COPY && dst) {
                    uint8_t *d = dst + b * 64;
                    st16(d, make_uint4(m[0], m[1], m[2], m[3]));
                    st16(d + 16, make_uint4(m[4], m[5], m[6], m[7]));
                    st16(d + 32, make_uint4(m[8], m[9], m[10], m[11]));
                    st16(d + 48, make_uint4(m[12], m[13], m[14], m[15]));
                }
                const uint32_t flags = (b == 0 ? b3::CHUNK_START : 0u) |
                                       (b == 15 ? (b3::CHUNK_END | (single ? b3::ROOT : 0u)) : 0u);
                b3::compress(cv, m, k, 0, 64, flags);
            };
            if (COPY && STAGE_FULL && stage && ALL_LDS && t.n_units == 0 && __ballot(active && dst == nullptr) == 0ull) {
                // Big-slice tile hashed from a window (fused block kernel): the row stores leave as whole lines too —
                // every lane drops its blocks in its slot, and every second compression the pair goes out transposed,
                // 8 lanes x 16 bytes per leaf, 8 leaves per store instruction.
                const uint64_t db = ((uint64_t)__shfl((uint32_t)((uint64_t)(uintptr_t)dst >> 32), 0) << 32) | __shfl((uint32_t)(uintptr_t)dst, 0);
                uint8_t *const d0 = reinterpret_cast<uint8_t *>((uintptr_t)db) + ((uint64_t)(lane >> 3) << 10) + 16 * (lane & 7);
                const uint64_t act = __ballot(active);
                bool actj[8];
#pragma unroll
                for (int j = 0; j < 8; j++) actj[j] = (act >> (8 * j + (lane >> 3))) & 1;
                const lds_u4a *const r0 = (const lds_u4a *)(stage + (lane >> 3) * STAGE_FULL_SLOT + 16 * (lane & 7));
                lds_u4a *const ws = (lds_u4a *)(stage + lane * STAGE_FULL_SLOT);
                if (active) fetch(0);
#pragma unroll 1
                for (uint32_t b = 0; b < 16; b++) {
                    uint32_t m[16];
                    if (active) {
                        m[0] = n0.x; m[1] = n0.y; m[2] = n0.z; m[3] = n0.w; m[4] = n1.x; m[5] = n1.y; m[6] = n1.z; m[7] = n1.w;
                        m[8] = n2.x; m[9] = n2.y; m[10] = n2.z; m[11] = n2.w; m[12] = n3.x; m[13] = n3.y; m[14] = n3.z; m[15] = n3.w;
                        if (b < 15) fetch(b + 1);
                        lds_u4a *const wh = ws + 4 * (b & 1);
                        wh[0] = u4v{m[0], m[1], m[2], m[3]}; wh[1] = u4v{m[4], m[5], m[6], m[7]};
                        wh[2] = u4v{m[8], m[9], m[10], m[11]}; wh[3] = u4v{m[12], m[13], m[14], m[15]};
                    }
                    if (b & 1) {
#pragma unroll
                        for (int j = 0; j < 8; j++)
                            if (actj[j]) {
                                const u4v v = r0[j * (8 * STAGE_FULL_SLOT / 16)];
                                st16(d0 + j * 8192 + (b >> 1) * 128, make_uint4(v.x, v.y, v.z, v.w));
                            }
                    }
                    if (active) {
                        const uint32_t flags = (b == 0 ? b3::CHUNK_START : 0u) |
                                               (b == 15 ? (b3::CHUNK_END | (single ? b3::ROOT : 0u)) : 0u);
                        b3::compress(cv, m, k, 0, 64, flags);
                    }
                }
            } else if (COPY && STAGE_SHIFT && stage && !LDSRC && t.n_units == 0 && __ballot(active && dst == nullptr) == 0ull &&
                       (__shfl((uint32_t)(uintptr_t)dst, 0) & 15) != 0) {
                // The same for a destination that is not 16-byte aligned (a stored round behind a compressed one in the
                // packed blob region): 16-byte stores at odd addresses cost the kernel half its speed, so the bytes are
                // re-cut on their way out of the stage.  A slot keeps the last 16 bytes of the previous 128 in front of
                // the current 128; output piece q is read from the slot `dl` bytes early (LDS reads may be unaligned) and
                // stored at the 16-byte boundary below its place.  What that leaves over — the first 16 - dl bytes of a
                // leaf and its last dl — goes out as one odd store each at the leaf's first and last step, with bytes of
                // that leaf only, so no store ever writes bytes another wave is responsible for.
                const uint64_t sb = ((uint64_t)__shfl((uint32_t)((uint64_t)(uintptr_t)src >> 32), 0) << 32) | __shfl((uint32_t)(uintptr_t)src, 0);
                const uint64_t db = ((uint64_t)__shfl((uint32_t)((uint64_t)(uintptr_t)dst >> 32), 0) << 32) | __shfl((uint32_t)(uintptr_t)dst, 0);
                const uint32_t dl = (uint32_t)db & 15, q = lane & 7;
                const uint8_t *const s0 = reinterpret_cast<const uint8_t *>((uintptr_t)sb) + ((uint64_t)(lane >> 3) << 10) + 16 * q;
                uint8_t *const d0 = reinterpret_cast<uint8_t *>((uintptr_t)db) + ((uint64_t)(lane >> 3) << 10) + 16 * q;  // my piece's own place
                const uint64_t act = __ballot(active);
                bool actj[8];
#pragma unroll
                for (int j = 0; j < 8; j++) actj[j] = (act >> (8 * j + (lane >> 3))) & 1;
                uint8_t *const slot = stage + (lane >> 3) * STAGE_SHIFT_SLOT;  // leaf lane/8 (+ 8j: j * 8 slots further)
                constexpr uint32_t JS = 8 * STAGE_SHIFT_SLOT;
                const lds_u4a *const own = (const lds_u4a *)(stage + lane * STAGE_SHIFT_SLOT + 16);
                // FULL: all 64 leaves are there — no lane-dependent branch around the memory instructions (counted waits), and
                // the step's order of the aligned form below: stage write, this pair's stores, the next pair's loads over the
                // same registers, two compressions (C5's write side 3.55 -> 3.36 ms same box; the branch-free loop alone, with
                // the ragged form's loads before stores into a second set of registers: no gain).
                auto recut = [&](auto full_tile) {
                    constexpr bool FULL = decltype(full_tile)::value;
                    constexpr bool ONE_SET = FULL;  // stores before the next loads, over one set of registers
                    uint4 v[8], vn[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) { v[j] = make_uint4(0, 0, 0, 0); vn[j] = v[j]; if (FULL || actj[j]) v[j] = ld16(s0 + j * 8192); }
#pragma unroll 1
                    for (uint32_t bb = 0; bb < 8; bb++) {
                        if (bb && q == 7) {  // the previous 128 bytes' tail moves in front before they are overwritten
#pragma unroll
                            for (int j = 0; j < 8; j++)
                                if (FULL || actj[j]) *(lds_u4a *)(slot + j * JS) = *(const lds_u4a *)(slot + j * JS + 128);
                        }
#pragma unroll
                        for (int j = 0; j < 8; j++)
                            if (FULL || actj[j]) *(lds_u4a *)(slot + j * JS + 16 + 16 * q) = u4v{v[j].x, v[j].y, v[j].z, v[j].w};
                        if (!ONE_SET && bb < 7) {
#pragma unroll
                            for (int j = 0; j < 8; j++)
                                if (FULL || actj[j]) vn[j] = ld16(s0 + j * 8192 + (bb + 1) * 128);
                        }
                        // piece q, re-cut: bytes [16q - dl, 16q - dl + 16) of the slot's chunk to the boundary below; the
                        // leaf's very first piece has nothing in front of it: it goes out as it is, to its own odd place
                        const bool first = bb == 0 && q == 0;
                        const uint32_t sh = first ? 0u : dl;
#pragma unroll
                        for (int j = 0; j < 8; j++)
                            if (FULL || actj[j]) {
                                const u4v o = *(const lds_u4 *)(slot + j * JS + 16 + 16 * q - sh);
                                st16(d0 + j * 8192 + bb * 128 - sh, make_uint4(o.x, o.y, o.z, o.w));
                            }
                        if (bb == 7 && q == 7) {  // the leaf's last dl bytes: its last 16, to their own odd place
#pragma unroll
                            for (int j = 0; j < 8; j++)
                                if (FULL || actj[j]) st16(d0 + j * 8192 + bb * 128, v[j]);
                        }
                        if (ONE_SET && bb < 7) {
#pragma unroll
                            for (int j = 0; j < 8; j++) v[j] = ld16(s0 + j * 8192 + (bb + 1) * 128);
                        }
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            if (FULL || active) {
                                const u4v a0 = own[4 * h], a1 = own[4 * h + 1], a2 = own[4 * h + 2], a3 = own[4 * h + 3];
                                uint32_t m[16] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w, a3.x, a3.y, a3.z, a3.w};
                                const uint32_t b = 2 * bb + h;
                                const uint32_t flags = (b == 0 ? b3::CHUNK_START : 0u) |
                                                       (b == 15 ? (b3::CHUNK_END | (single ? b3::ROOT : 0u)) : 0u);
                                b3::compress(cv, m, k, 0, 64, flags);
                            }
                        }
                        if (!ONE_SET) {
#pragma unroll
                            for (int j = 0; j < 8; j++) v[j] = vn[j];
                        }
                    }
                };
                if (act == ~0ull) recut(std::true_type{});
                else recut(std::false_type{});
            } else if (COPY && STAGE_FULL && stage && !LDSRC && t.n_units == 0 && __ballot(active && dst == nullptr) == 0ull) {
                // Big-slice tile of the store path, whole cache lines: a leaf's bytes move 128 at a time (two blocks), 8
                // lanes per leaf, 8 leaves per instruction — the 64-byte form fetched every line from HBM twice (PMC:
                // 1.8x the bytes), once for each half, a compression apart.  The tile's 64 leaves are contiguous, so
                // the transposed addresses are one base + 8 KiB steps.
                // lane 0 is the tile's first leaf (always active): its pointers are the tile's base
                const uint64_t sb = ((uint64_t)__shfl((uint32_t)((uint64_t)(uintptr_t)src >> 32), 0) << 32) | __shfl((uint32_t)(uintptr_t)src, 0);
                const uint64_t db = ((uint64_t)__shfl((uint32_t)((uint64_t)(uintptr_t)dst >> 32), 0) << 32) | __shfl((uint32_t)(uintptr_t)dst, 0);
                const uint8_t *const s0 = reinterpret_cast<const uint8_t *>((uintptr_t)sb) + ((uint64_t)(lane >> 3) << 10) + 16 * (lane & 7);
                uint8_t *const d0 = reinterpret_cast<uint8_t *>((uintptr_t)db) + ((uint64_t)(lane >> 3) << 10) + 16 * (lane & 7);
                const uint64_t act = __ballot(active);
                bool actj[8];
#pragma unroll
                for (int j = 0; j < 8; j++) actj[j] = (act >> (8 * j + (lane >> 3))) & 1;
                // the stage: 128-byte slots, piece p of slot l at piece p ^ ((l >> 1) & 7) (STAGE_SWZ_BYTES).  Loading lane
                // (lane >> 3, lane & 7) fills slot lane >> 3 (+ 8j), whose key is (lane >> 4) + 4 * (j & 1).
                uint8_t *const rz0 = stage + (lane >> 3) * 128 + 16 * ((lane & 7) ^ (lane >> 4));
                uint8_t *const rz1 = stage + (lane >> 3) * 128 + 16 * ((lane & 7) ^ ((lane >> 4) + 4));
                const uint8_t *const wz = stage + lane * 128;
                const uint32_t wkey = 16 * ((lane >> 1) & 7);
                // FULL: all 64 leaves of the tile are there (every tile of a big unit but its last) — no lane-dependent
                // branch in the loop, so its waits are counted ones (with the per-group branches of the ragged form every
                // memory instruction sits behind an s_waitcnt vmcnt(0)).
                // Order of a step, measured (C4 store / C5, same box): stores of this pair, loads of the next over the same
                // registers, two compressions — 3 % / 5 % faster than loads before stores, with or without a second set of
                // registers; five waves per SIMD instead of four change nothing.
                auto pairs = [&](auto full_tile) {
                    constexpr bool FULL = decltype(full_tile)::value;
                    uint4 v[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) { v[j] = make_uint4(0, 0, 0, 0); if (FULL || actj[j]) v[j] = ld16(s0 + j * 8192); }
#pragma unroll 1
                    for (uint32_t bb = 0; bb < 8; bb++) {
#pragma unroll
                        for (int j = 0; j < 8; j++)
                            if (FULL || actj[j]) *(lds_u4a *)(((j & 1) ? rz1 : rz0) + j * 1024) = u4v{v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
                        for (int j = 0; j < 8; j++)
                            if (FULL || actj[j]) st16(d0 + j * 8192 + bb * 128, v[j]);
                        if (bb < 7) {
#pragma unroll
                            for (int j = 0; j < 8; j++)
                                if (FULL || actj[j]) v[j] = ld16(s0 + j * 8192 + (bb + 1) * 128);
                        }
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            if (FULL || active) {
                                const u4v a0 = *(const lds_u4a *)(wz + ((64 * h) ^ wkey)), a1 = *(const lds_u4a *)(wz + ((64 * h + 16) ^ wkey)),
                                          a2 = *(const lds_u4a *)(wz + ((64 * h + 32) ^ wkey)), a3 = *(const lds_u4a *)(wz + ((64 * h + 48) ^ wkey));
                                uint32_t m[16] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w, a3.x, a3.y, a3.z, a3.w};
                                const uint32_t b = 2 * bb + h;
                                const uint32_t flags = (b == 0 ? b3::CHUNK_START : 0u) |
                                                       (b == 15 ? (b3::CHUNK_END | (single ? b3::ROOT : 0u)) : 0u);
                                b3::compress(cv, m, k, 0, 64, flags);
                            }
                        }
                    }
                };
                if (act == ~0ull) pairs(std::true_type{});
                else pairs(std::false_type{});
            } else if (COPY && stage && !LDSRC && ZN_STAGE_LOADS) {
                // both directions through the stage: lane l moves piece l%4 of leaves 16j + l/4 (j = 0..3), so a load
                // or a store instruction covers 16 leaves x 64 contiguous bytes.  A block's registers are written to
                // the stage (every lane then reads its own leaf's block back) and stored to the output as they are.
                // Order inside a pass: stage write, NEXT block's loads, then this block's stores — vmcnt completes in
                // order, so the wait for the next block must not have this block's stores in front of it.
                const uint64_t has = __ballot(active && dst != nullptr), act = __ballot(active);
                const uint8_t *sj[4];
                uint8_t *pj[4];
                bool onj[4], actj[4];
                lds_u4a *rs[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t leaf = 16 * j + (lane >> 2);
                    const uint64_t d64 = (uint64_t)(uintptr_t)dst, s64 = (uint64_t)(uintptr_t)src;
                    const uint64_t g = ((uint64_t)__shfl((uint32_t)(d64 >> 32), leaf) << 32) | __shfl((uint32_t)d64, leaf);
                    const uint64_t h = ((uint64_t)__shfl((uint32_t)(s64 >> 32), leaf) << 32) | __shfl((uint32_t)s64, leaf);
                    pj[j] = reinterpret_cast<uint8_t *>((uintptr_t)g) + 16 * (lane & 3);
                    sj[j] = reinterpret_cast<const uint8_t *>((uintptr_t)h) + 16 * (lane & 3);
                    onj[j] = (has >> leaf) & 1;
                    actj[j] = (act >> leaf) & 1;
                    rs[j] = (lds_u4a *)(stage + leaf * STAGE_SLOT + 16 * (lane & 3));
                }
                const lds_u4a *ws = (const lds_u4a *)(stage + lane * STAGE_SLOT);
                uint4 v[4], vn[4];
#pragma unroll
                for (int j = 0; j < 4; j++) { v[j] = make_uint4(0, 0, 0, 0); vn[j] = v[j]; if (actj[j]) v[j] = ld16(sj[j]); }
#pragma unroll 1
                for (uint32_t b = 0; b < 16; b++) {
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (actj[j]) *rs[j] = u4v{v[j].x, v[j].y, v[j].z, v[j].w};
                    if (b < 15) {
#pragma unroll
                        for (int j = 0; j < 4; j++)
                            if (actj[j]) vn[j] = ld16(sj[j] + (b + 1) * 64);
                    }
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (onj[j]) st16(pj[j] + b * 64, v[j]);
                    if (active) {
                        const u4v a0 = ws[0], a1 = ws[1], a2 = ws[2], a3 = ws[3];
                        uint32_t m[16] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w, a3.x, a3.y, a3.z, a3.w};
                        const uint32_t flags = (b == 0 ? b3::CHUNK_START : 0u) |
                                               (b == 15 ? (b3::CHUNK_END | (single ? b3::ROOT : 0u)) : 0u);
                        b3::compress(cv, m, k, 0, 64, flags);
                    }
#pragma unroll
                    for (int j = 0; j < 4; j++) v[j] = vn[j];
                }
            } else if (COPY && stage) {
                // the copy goes out transposed: every lane leaves its block in its stage slot, then lane l stores
                // piece l%4 of leaves 16j + l/4 (j = 0..3) — 16 leaves x 64 contiguous bytes per store instruction.
                // (store-only form, kept for ZN_STAGE_LOADS=0: with aligned sources it is as fast as the two-way form
                // above; raw blocks inside a frame start at odd addresses and gain 16 % from coalesced loads)
                const uint64_t has = __ballot(active && dst != nullptr);
                uint8_t *pj[4];
                bool onj[4];
                const lds_u4a *rs[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t leaf = 16 * j + (lane >> 2);
                    const uint64_t d64 = (uint64_t)(uintptr_t)dst;
                    const uint64_t g = ((uint64_t)__shfl((uint32_t)(d64 >> 32), leaf) << 32) | __shfl((uint32_t)d64, leaf);
                    pj[j] = reinterpret_cast<uint8_t *>((uintptr_t)g) + 16 * (lane & 3);
                    onj[j] = (has >> leaf) & 1;
                    rs[j] = (const lds_u4a *)(stage + leaf * STAGE_SLOT + 16 * (lane & 3));
                }
                lds_u4a *ws = (lds_u4a *)(stage + lane * STAGE_SLOT);
                if (active) fetch(0);
#pragma unroll 1
                for (uint32_t b = 0; b < 16; b++) {
                    uint32_t m[16];
                    if (active) {
                        m[0] = n0.x; m[1] = n0.y; m[2] = n0.z; m[3] = n0.w; m[4] = n1.x; m[5] = n1.y; m[6] = n1.z; m[7] = n1.w;
                        m[8] = n2.x; m[9] = n2.y; m[10] = n2.z; m[11] = n2.w; m[12] = n3.x; m[13] = n3.y; m[14] = n3.z; m[15] = n3.w;
                        if (b < 15) fetch(b + 1);
                        if (dst) {
                            ws[0] = u4v{m[0], m[1], m[2], m[3]}; ws[1] = u4v{m[4], m[5], m[6], m[7]};
                            ws[2] = u4v{m[8], m[9], m[10], m[11]}; ws[3] = u4v{m[12], m[13], m[14], m[15]};
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (onj[j]) {
                            const u4v v = *rs[j];
                            st16(pj[j] + b * 64, make_uint4(v.x, v.y, v.z, v.w));
                        }
                    if (active) {
                        const uint32_t flags = (b == 0 ? b3::CHUNK_START : 0u) |
                                               (b == 15 ? (b3::CHUNK_END | (single ? b3::ROOT : 0u)) : 0u);
                        b3::compress(cv, m, k, 0, 64, flags);
                    }
                }
            } else if (active) {
                fetch(0);
#pragma unroll 1
                for (uint32_t b = 0; b < 16; b++) block(b);
            }
        };
        if (LDSRC && __ballot(active && !Y) == 0ull) run(std::true_type{});
        else run(std::false_type{});
    } else if (COPY && !LDSRC && stage && ZN_STAGE_LOADS && ZN_RAGGED_STAGE && __ballot(active && ulen != 0 && ulen < 16) == 0ull) {
        const Cv8 r = hash_ragged_through_stage(src, dst, leaf_len, nblk, k, single, active, stage);
#pragma unroll
        for (int i = 0; i < 8; i++) cv[i] = r.v[i];
    } else {
        // generic path: ragged / partial / empty leaves
        uint32_t maxblk = active ? nblk : 0;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            uint32_t o = __shfl_xor(maxblk, d);
            maxblk = o > maxblk ? o : maxblk;
        }
        maxblk = __builtin_amdgcn_readfirstlane(maxblk);
        for (uint32_t b = 0; b < maxblk; b++) {
            if (active && b < nblk) {
                uint32_t m[16];
                uint32_t rem = leaf_len - b * 64;
                uint32_t bl = leaf_len == 0 ? 0 : (rem < 64 ? rem : 64);
                load_block(src + b * 64, bl, m);
                if (COPY && dst) store_block(dst + b * 64, bl, m);
                uint32_t flags = (b == 0 ? b3::CHUNK_START : 0u) |
                                 (b == nblk - 1 ? (b3::CHUNK_END | (single ? b3::ROOT : 0u)) : 0u);
                b3::compress(cv, m, k, 0, bl, flags);
            }
        }
    }

#pragma unroll
    for (int i = 0; i < 8; i++) out.cv[i] = cv[i];
    out.seg_start = seg_start;
    out.n = active ? unit_leaves : 0;
    out.unit = unit;
    out.active = active;
}

// Parent tree of one tile, folded on the spot (4-6 compress passes with few lanes busy).
__device__ __forceinline__ void fold_tile_now(const HashArgs &a, const Tile &t, LeafOut &lo) {
    const uint32_t lane = threadIdx.x & 63;
    fold_segments(lo.cv, lo.seg_start, lo.n, t.n_units != 0);
    if (lo.active && lane == lo.seg_start) {
        uint32_t *o = t.n_units ? a.digests + (size_t)lo.unit * 8 : a.tile_cv + (size_t)t.cv_index * 8;
#pragma unroll
        for (int i = 0; i < 8; i++) o[i] = lo.cv[i];
    }
}

// Hash one tile with the calling wavefront.  Every lane of the wave must call it.
template <bool COPY, bool LDSRC = false>
__device__ __forceinline__ void hash_tile(const HashArgs &a, const Tile &t, const LdsSrc *ls = nullptr) {
    LeafOut lo;
    hash_tile_leaves<COPY, LDSRC>(a, t, ls, lo);
    fold_tile_now(a, t, lo);
}

// Deferred parent folding.  A parent level costs one full compress pass of the wave however few lanes take
// part (a 10-leaf unit folds in 4 passes with 5, 3, 2, 1 lanes busy; a 64-leaf slice in 6).  The queue keeps
// the leaf CVs of up to G tiles in a wave-private LDS array and folds all of their units together, level by
// level, with the nodes of a level packed over the 64 lanes: 6 rows x 10 leaves x 4 tiles fold in 6 passes
// instead of 16, four 64-leaf slices in 7 instead of 24.
//   nodes : wave-private LDS, G*64 chaining values; tile g's leaf CVs start at node g*64
//   table : one queued unit per lane (registers): node count, first node, where the result goes
template <int G>
struct FoldQueue {
    static constexpr uint32_t MAX_UNITS_PER_TILE = 64 / G;  // tiles with more units fold on the spot
    uint32_t n_tab = 0;
    uint32_t tb_n = 0, tb_off = 0, tb_out = 0, tb_root = 0;

    static __device__ __forceinline__ bool fits(const Tile &t) { return (t.n_units ? t.n_units : 1u) <= MAX_UNITS_PER_TILE; }

    __device__ __forceinline__ void add(uint32_t *nodes, uint32_t g, const Tile &t, const LeafOut &lo) {
        const uint32_t lane = threadIdx.x & 63;
        if (lo.active) {
            uint4 *d = reinterpret_cast<uint4 *>(nodes + (size_t)(g * 64 + lane) * 8);
            d[0] = make_uint4(lo.cv[0], lo.cv[1], lo.cv[2], lo.cv[3]);
            d[1] = make_uint4(lo.cv[4], lo.cv[5], lo.cv[6], lo.cv[7]);
        }
        add_entry(g, t, lo);
    }
    // the table entries alone: the caller puts the tile's leaf CVs at nodes[g * 64 + lane] itself, before the fold
    __device__ __forceinline__ void add_entry(uint32_t g, const Tile &t, const LeafOut &lo) {
        const uint32_t lane = threadIdx.x & 63;
        const uint32_t units = t.n_units ? t.n_units : 1u;
        // descriptor of tile-local unit i on lane i, then moved to table lanes [n_tab, n_tab + units)
        const uint32_t act = __shfl(lo.active ? 1u : 0u, lo.u_head & 63);
        const uint32_t d_n = (lane < units && act) ? lo.u_cnt : 0u;
        const uint32_t d_off = g * 64 + lo.u_head;
        const uint32_t d_out = t.n_units ? t.first_unit + lane : t.cv_index;
        const uint32_t src = (lane - n_tab) & 63;
        const uint32_t v_n = __shfl(d_n, src), v_off = __shfl(d_off, src), v_out = __shfl(d_out, src);
        if (lane >= n_tab && lane < n_tab + units) {
            tb_n = v_n; tb_off = v_off; tb_out = v_out; tb_root = t.n_units ? 1u : 0u;
        }
        n_tab += units;
    }

    __device__ __forceinline__ void fold_and_write(uint32_t *nodes, const HashArgs &a) {
        const uint32_t lane = threadIdx.x & 63;
        if (n_tab == 0) return;
        uint32_t n = lane < n_tab ? tb_n : 0u, off = tb_off;
        const bool valid = n > 0;
        while (__ballot(n > 1) != 0ull) {
            const uint32_t out_cnt = n > 1 ? (n + 1) >> 1 : n;
            uint32_t inc = out_cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                uint32_t y = __shfl_up(inc, d);
                if (lane >= (uint32_t)d) inc += y;
            }
            const uint32_t total = __shfl(inc, 63);
            const uint32_t my_oo = inc - out_cnt;
            for (uint32_t p = 0; p * 64 < total; p++) {
                const uint32_t q = p * 64 + lane;
                const bool on = q < total;
                uint32_t lo = 0, hi = 63;  // smallest u with inc[u] > q
#pragma unroll
                for (int it = 0; it < 6; it++) {
                    const uint32_t mid = (lo + hi) >> 1;
                    const uint32_t v = __shfl(inc, mid);
                    if (lo < hi) {
                        if (v > q) hi = mid; else lo = mid + 1;
                    }
                }
                const uint32_t u = lo;
                const uint32_t cu = __shfl(n, u), io = __shfl(off, u), oo = __shfl(my_oo, u), rt = __shfl(tb_root, u);
                const uint32_t j = q - oo;
                uint32_t cv[8];
                if (on) {
                    const uint4 *s = reinterpret_cast<const uint4 *>(nodes + (size_t)(io + 2 * j) * 8);
                    const uint4 l0 = s[0], l1 = s[1];
                    uint32_t L[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
                    if (cu > 1 && 2 * j + 1 < cu) {
                        const uint4 r0 = s[2], r1 = s[3];
                        uint32_t R[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
                        b3::parent(cv, L, R, rt && cu == 2);
                    } else {
#pragma unroll
                        for (int i = 0; i < 8; i++) cv[i] = L[i];
                    }
                }
                // every lane has read its children before any lane overwrites a node (one wave: LDS operations
                // execute in program order); node q only ever replaces a node at or before its own children
                __builtin_amdgcn_wave_barrier();
                if (on) {
                    uint4 *d = reinterpret_cast<uint4 *>(nodes + (size_t)q * 8);
                    d[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
                    d[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
                }
                __builtin_amdgcn_wave_barrier();
            }
            n = out_cnt;
            off = my_oo;
        }
        if (valid) {
            const uint4 *s = reinterpret_cast<const uint4 *>(nodes + (size_t)off * 8);
            const uint4 c0 = s[0], c1 = s[1];
            uint4 *o = reinterpret_cast<uint4 *>((tb_root ? a.digests : a.tile_cv) + (size_t)tb_out * 8);
            o[0] = c0;
            o[1] = c1;
        }
        n_tab = 0;
    }

    // The same for the common case that every queued unit has the SAME number of nodes n >= 2 (a table of equal rows:
    // BASELINE's 100k x 10 KiB; the slices of big rows): the node -> (unit, pair) mapping is then a division by a
    // wave-uniform constant instead of a prefix scan + binary search through ds_bpermute (16 dependent shuffles per
    // pass), and a level's odd node is CARRIED as a copy instead of taking a lane of a compress pass — four 6 x 10-leaf
    // tiles fold in 5 passes instead of 6, and a pass is little more than its compression.
    // Units must be queued densely on lanes [0, U) (tb_n = n there, 0 behind) — see uniform().
    // Layout: level l keeps unit u's n_l nodes at [u * n_l, ...) (level 0: at tb_off), in place: output u * c + j never
    // lies behind its own children, and a level's carries are read before its parents are written.
    // A table kept as four groups of 16 lanes (group g = tile g, its units on the group's first lanes), moved to
    // lanes [0, U): what fold_uniform_and_write wants.  (A group with a gap — a unit that is not hashed — leaves a gap:
    // uniform() then says no and the caller folds the original table the general way.)
    __device__ __forceinline__ FoldQueue<G> dense16() const {
        const uint32_t lane = threadIdx.x & 63;
        const uint64_t has = __ballot(lane < n_tab && tb_n != 0);
        const uint32_t c0 = (uint32_t)__popcll(has & 0xFFFFull), c1 = c0 + (uint32_t)__popcll(has >> 16 & 0xFFFFull),
                       c2 = c1 + (uint32_t)__popcll(has >> 32 & 0xFFFFull), c3 = c2 + (uint32_t)__popcll(has >> 48);
        const uint32_t g = lane < c0 ? 0u : (lane < c1 ? 1u : (lane < c2 ? 2u : 3u));
        const uint32_t src = 16 * g + lane - (g == 0 ? 0u : (g == 1 ? c0 : (g == 2 ? c1 : c2)));
        FoldQueue<G> d;
        d.n_tab = c3;
        d.tb_n = __shfl(tb_n, src & 63);
        d.tb_off = __shfl(tb_off, src & 63);
        d.tb_out = __shfl(tb_out, src & 63);
        d.tb_root = __shfl(tb_root, src & 63);
        if (lane >= c3) d.tb_n = 0;
        return d;
    }
    __device__ __forceinline__ uint32_t uniform(uint32_t *units) const {  // -> n if the table is U x n (n >= 2), else 0
        const uint32_t lane = threadIdx.x & 63;
        const uint32_t n0 = __shfl(tb_n, 0);
        const uint64_t has = __ballot(lane < n_tab && tb_n != 0), same = __ballot(lane < n_tab && tb_n == n0);
        const uint32_t U = (uint32_t)__popcll(has);
        *units = U;
        // U != n_tab: dense16() took the first c entries of a 16-entry group where the group's c active ones were not its
        // first c (a unit that is not hashed between two that are) — one of the table's entries is then an inactive unit
        // and an ACTIVE one was left out.  When that entry happened to be the table's last, the remaining test (the
        // non-zero entries are lanes [0, U)) passed and the left-out unit's digest was never written: a valid row among
        // invalid ones reported corrupt, in whichever runs the left-over list happened to group the tiles that way.
        if (n0 < 2 || has != same || U == 0 || U != n_tab || has != (U == 64 ? ~0ull : ((1ull << U) - 1))) return 0;
        // all units or all slices: fold_uniform_and_write takes the kind from entry 0.  (A wave of the store path kernel
        // can queue a 64-leaf row — one unit, a root — beside the first 64-leaf slice of a big row: same node count,
        // different kinds; folded as one kind the slice's CV got the ROOT flag and went to the digest column.)
        const uint32_t root0 = __shfl(tb_root, 0);
        if (__ballot(lane < n_tab && tb_root != root0) != 0ull) return 0;
        return n0;
    }
    __device__ __forceinline__ void fold_uniform_and_write(uint32_t *nodes, const HashArgs &a, uint32_t n, uint32_t U) {
        const uint32_t lane = threadIdx.x & 63;
        const uint32_t root = __shfl(tb_root, 0);  // (a table is all units or all slices)
        uint32_t pos = tb_off;                     // lane u: first node of unit u at the current level
        for (uint32_t nl = n; nl > 1;) {
            const uint32_t h = nl >> 1, c = h + (nl & 1);
            uint4 k0 = make_uint4(0, 0, 0, 0), k1 = k0;
            if ((nl & 1) && lane < U) {  // the level's odd node of my unit, before anything is overwritten
                const uint4 *s = reinterpret_cast<const uint4 *>(nodes + (size_t)(pos + nl - 1) * 8);
                k0 = s[0]; k1 = s[1];
            }
            const float inv = 1.0f / (float)h;
            const uint32_t total = U * h;
            for (uint32_t p = 0; p * 64 < total; p++) {
                const uint32_t q = p * 64 + lane;
                const bool on = q < total;
                const uint32_t u = on ? (uint32_t)(((float)q + 0.5f) * inv) : 0u, j = q - u * h;
                const uint32_t io = __shfl(pos, u);
                uint32_t cv[8];
                if (on) {
                    const uint4 *s = reinterpret_cast<const uint4 *>(nodes + (size_t)(io + 2 * j) * 8);
                    const uint4 l0 = s[0], l1 = s[1], r0 = s[2], r1 = s[3];
                    uint32_t L[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
                    uint32_t R[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
                    b3::parent(cv, L, R, root && nl == 2);
                }
                __builtin_amdgcn_wave_barrier();  // every lane has read its children before any node is overwritten
                if (on) {
                    uint4 *d = reinterpret_cast<uint4 *>(nodes + (size_t)(u * c + j) * 8);
                    d[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
                    d[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
                }
                __builtin_amdgcn_wave_barrier();
            }
            if ((nl & 1) && lane < U) {
                uint4 *d = reinterpret_cast<uint4 *>(nodes + (size_t)(lane * c + h) * 8);
                d[0] = k0; d[1] = k1;
            }
            __builtin_amdgcn_wave_barrier();
            pos = lane * c;
            nl = c;
        }
        if (lane < U) {
            const uint4 *s = reinterpret_cast<const uint4 *>(nodes + (size_t)pos * 8);
            const uint4 c0 = s[0], c1 = s[1];
            uint4 *o = reinterpret_cast<uint4 *>((root ? a.digests : a.tile_cv) + (size_t)tb_out * 8);
            o[0] = c0;
            o[1] = c1;
        }
        n_tab = 0;
    }
};

}  // namespace zn
